"""The n=2 fused search alone, for profilers (tools/pmc_generic.sh OUT n2_search_kernel -- python tools/n2_search_run.py):
m=100, k=5 exhaustive (9.7e7 candidates) and a 2^33-candidate rank range of m=200, k=7.  Prints per-launch statistics."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import theta_amd

ctx = theta_amd.Context(0)
which = sys.argv[1] if len(sys.argv) > 1 else "both"
for tag, m, k, span in (("m100_k5", 100, 5, None), ("m200_k7", 200, 7, 1 << 33)):
    if which not in ("both", tag):
        continue
    r, rN, order = bench.synth(seed=11, m=m, n=2, k=k)
    p = theta_amd.Problem(ctx, 2, m, 2, r, rN, [0] * m, [k] * m, 1.0)
    b = 0 if span is None else p.count // 3
    e = p.count if span is None else b + span
    p.search(b, e)
    st = p.search(b, e)["stats"]
    print(tag, "cand %.4g  kernel_ms %.3f  C/s %.4g  iters/cand %.3f  terms/iter %.2f  accepted %.3f  flop/cand %.0f  TFLOP/s %.2f" % (
        e - b, st["kernel_ms"], (e - b) / st["kernel_ms"] * 1e3, st["iterations"] / st["evaluated"],
        st["terms"] / max(st["iterations"], 1), st["accepted"] / st["evaluated"], st["flops"] / st["evaluated"],
        st["flops"] / st["kernel_ms"] / 1e9))
    p.close()
