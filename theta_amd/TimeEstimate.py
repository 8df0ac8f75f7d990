"""
Candidate counting (python/TimeEstimate.py:91-142) and the pre-search feasibility guard (:40-86).
The exact counts come from the counting tables the HIP library builds for rank <-> candidate
unranking; the reference's n=3 figure is only an upper estimate (TimeEstimate.py:113-142, mirrored by
count_number_matrices_3), count_number_matrices(3, ...) is the number of matrices its enumerator really yields.
"""
import sys
import time

from . import _lib
from . import search as S


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


def count_number_matrices_3(m, upper_bounds, lower_bounds, enum=None, tau=2):
    """
    TimeEstimate.py:113-142: the reference's UPPER ESTIMATE of the n=3 candidate count -- paths through the row graph that
    respect the per-interval bounds, halved for the column symmetry, with the ratio-window pruning (Enumerator.py:204-212)
    ignored.  Kept for callers of the reference's function; the search itself works with the exact count
    (count_number_matrices(3, ...), the counting table of the HIP library).  `enum` (the reference passes its Enumerator
    for the row graph) is not needed: the graph is rebuilt from max(upper_bounds) and tau (Enumerator.py:272-298).
    """
    upper_bounds = [int(v) for v in upper_bounds]
    lower_bounds = [int(v) for v in lower_bounds]
    K = max(upper_bounds)
    rows = [(a, b) for b in range(K + 1) for a in range(K + 1) if (tau - a) * (tau - b) >= 0]
    edges = [[j for j, w in enumerate(rows) if (v == w or w[0] > v[0] or w[1] > v[1])] for v in rows]
    poss = [0] * len(rows)
    for i, row in enumerate(rows):
        if min(row) >= lower_bounds[0] and max(row) <= upper_bounds[0]:
            poss[i] += 1
    for i in range(m - 1):
        nxt = [0] * len(rows)
        for j, v in enumerate(poss):
            if v > 0:
                for k in edges[j]:
                    if all(lower_bounds[i + 1] <= a <= upper_bounds[i + 1] for a in rows[k]):
                        nxt[k] += v
        poss = nxt
    return sum(poss) // 2          # (Python 2 integer division in the reference)


def time_estimate(n, m, k, tau, lower_bounds, upper_bounds, r, rN, max_normal, sorted_index, num_processes, multi_event,
                  force):
    """
    TimeEstimate.py:40-86 -- same messages, same exits.  The reference times 100 (n=2) / 20 (n=3) CPU solves and
    extrapolates over its candidate count; here a small rank range is timed on the GPU and extrapolated over the exact count.
    Like the reference, n=3 with more than 30 intervals is refused unless --FORCE is given (:48-50).
    """
    print("Estimating time...")
    if n == 3 and m > 30 and not force:
        print("\tWARNING: With n=3 and", m, "intervals, the runtime would likely be excessive. Try reducing the number of "
              "intervals below 25. Run with --FORCE to continue.")
        sys.exit(1)
    try:
        ctx = _lib.default_context()
        p = _lib.Problem(ctx, n, m, tau, r, rN, [int(v) for v in lower_bounds], [int(v) for v in upper_bounds], max_normal)
    except _lib.ThetaError as e:
        if e.code in (_lib.ERR_OVERFLOW, _lib.ERR_ARG):
            # a search the library cannot hold (n=3: more than 256 intervals, copy numbers above 15)
            print("ERROR: %s. Use fewer intervals (--NUM_INTERVALS) or tighter bounds. Exiting..." % e)
            sys.exit(1)
        raise
    count = p.count
    if count == 0:
        print("ERROR: No valid Copy Number Profiles exist for these intervals within the bounds specified. Exiting...")
        sys.exit(1)
    seconds = None
    if n == 3 and count >= S.BNB_MIN_CANDIDATES and S.USE_MIX:
        # a space the search does not walk rank by rank (search.py: mix_records, branch and bound over the mixture space): the
        # estimate is the search itself, timed -- about a second where it succeeds; where it gives up (a flat likelihood) the
        # extrapolation below stands
        t0 = time.time()
        try:
            S.mix_records(p, ctx, r, rN, max_normal, ([int(v) for v in lower_bounds], [int(v) for v in upper_bounds]))
            seconds = 2.0 * (time.time() - t0)
        except _lib.ThetaError:
            pass
    if seconds is None:
        probe = min(count, 1 << 20)
        t0 = time.time()
        try:
            p.search(0, probe, window=0.0)
        except _lib.ThetaError:
            pass                      # (an estimate only: the search proper reports errors)
        rate = probe / max(time.time() - t0, 1e-6)
        seconds = count / rate
    p.close()
    if seconds < 60:
        print("\tEstimated Total Time:", int(seconds + .5), "second(s)")
    elif seconds < 3600:
        print("\tEstimated Total Time:", int((seconds / 60) + .5), "minute(s)")
    else:
        hours = int((seconds / 3600) + .5)
        print("\tEstimated Total Time:", hours, "hour(s)")
        if hours > 200 and not force:
            print("WARNING: With the current settings, the runtime is likely excessive. To reduce runtime, try:\n"
                  "\t1) Increase the number of processes used with the --NUM_PROCESSES flag.\n"
                  "\t2) Reduce the number of intervals chosen using the --NUM_INTERVALS flag.\n"
                  "\t3) Disable automatic interval selection using --NO_INTERVAL_SELECTION, and hand-select a smaller "
                  "number of intervals, or set tighter bounds on the current intervals.\n"
                  "\t Run with --FORCE to continue with current settings.")
            sys.exit(1)
    return count
