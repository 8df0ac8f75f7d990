"""
Candidate counting (python/TimeEstimate.py:91-142) and the pre-search feasibility guard (:40-86).
The exact counts come from the counting tables the HIP library builds for rank <-> candidate
unranking; the reference's n=3 figure is only an upper estimate (TimeEstimate.py:113-142), ours is
the number of matrices its enumerator really yields.
"""
import sys
import time

from . import _lib


def count_number_matrices(n, m, tau, upper_bounds, lower_bounds, ctx=None):
    """Exact number of candidates generate_next_C() yields for these bounds."""
    ctx = ctx or _lib.default_context()
    p = _lib.Problem(ctx, n, m, tau, [1] * m, [1] * m, [int(v) for v in lower_bounds], [int(v) for v in upper_bounds], 1.0)
    cnt = p.count
    p.close()
    return cnt


def count_number_matrices_2(m, upper_bounds, lower_bounds):
    """TimeEstimate.py:91-111."""
    return count_number_matrices(2, m, 2, upper_bounds, lower_bounds)


def time_estimate(n, m, k, tau, lower_bounds, upper_bounds, r, rN, max_normal, sorted_index, num_processes, multi_event,
                  force):
    """
    TimeEstimate.py:40-86.  The reference times 100 (n=2) / 20 (n=3) CPU solves and extrapolates; here a
    small rank range is timed on the GPU.  The n=3 'm > 30' refusal is kept only as a warning: the GPU
    search of a bounded rank space is feasible where the CPU one was not.
    """
    print("Estimating time...")
    ctx = _lib.default_context()
    p = _lib.Problem(ctx, n, m, tau, r, rN, [int(v) for v in lower_bounds], [int(v) for v in upper_bounds], max_normal)
    count = p.count
    if count == 0:
        print("ERROR: No valid Copy Number Profiles exist for these intervals within the bounds specified. Exiting...")
        sys.exit(1)
    probe = min(count, 1 << 20)
    t0 = time.time()
    p.search(0, probe, window=0.0)
    rate = probe / max(time.time() - t0, 1e-6)
    p.close()
    seconds = count / rate
    if seconds < 60:
        print("\tEstimated Total Time:", int(seconds + .5), "second(s)")
    elif seconds < 3600:
        print("\tEstimated Total Time:", int((seconds / 60) + .5), "minute(s)")
    else:
        hours = int((seconds / 3600) + .5)
        print("\tEstimated Total Time:", hours, "hour(s)")
        if hours > 200 and not force:
            print("WARNING: With the current settings, the runtime is likely excessive. To reduce runtime, try:\n"
                  "\t1) Reduce the number of intervals chosen using the --NUM_INTERVALS flag.\n"
                  "\t2) Disable automatic interval selection using --NO_INTERVAL_SELECTION, and hand-select a smaller "
                  "number of intervals, or set tighter bounds on the current intervals.\n"
                  "\t Run with --FORCE to continue with current settings.")
            sys.exit(1)
    return count
