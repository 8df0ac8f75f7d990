import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, theta_amd
ctx = theta_amd.Context(0)
for m, k in ((50, 6), (100, 5)):
    r, rN, order = bench.synth(seed=11, m=m, n=2, k=k)
    p = theta_amd.Problem(ctx, 2, m, 2, r, rN, [0] * m, [k] * m, 1.0)
    p.search(0, p.count)
    st = p.search(0, p.count)["stats"]
    print(m, k, "cand %.3g  kernel_ms %.2f  C/s %.3g  iters/cand %.2f  terms/iter %.2f  accepted %.3f  flops/cand %.0f" % (
        p.count, st["kernel_ms"], p.count / st["kernel_ms"] * 1e3, st["iterations"] / st["evaluated"],
        st["terms"] / max(st["iterations"], 1), st["accepted"] / st["evaluated"], st["flops"] / st["evaluated"]))
