import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np, theta_amd, theta_oracle as orc
ctx = theta_amd.Context(0)
for (m, k, seed) in ((10, 3, 1), (12, 3, 2), (14, 2, 3), (16, 2, 4), (10, 4, 5), (20, 2, 6)):
    r, rN, L, Ct, mu = orc.synth_counts(m, 3, k, seed)
    rs, rNs, order = orc.sort_r(rN, r)
    p = theta_amd.Problem(ctx, 3, m, 2, rs, rNs, [0] * m, [k] * m)
    res = p.search(0, p.count, window=0.5)
    st = res["stats"]
    rk, lb, Cs = p.last_suspects
    bm = ctx.boundary_min(2, rs, rNs, Cs) if len(rk) else np.array([np.inf])
    print("   suspects=%d  min unconstrained=%.4f  min over simplex boundary=%.4f  (winner %.4f)" % (len(rk), lb.min() if len(rk) else np.inf, bm.min(), st["best_nll"]))
    print("m=%d k=%d count=%.3g acc=%.3f best=%.4f rejected_bound=%.4f gap=%.4g ties=%d truth_mu=%s" % (
        m, k, p.count, st["accepted"] / st["evaluated"], st["best_nll"], st["rejected_bound"],
        st["rejected_bound"] - st["best_nll"], len(res["rank"]), np.round(mu, 3)))
