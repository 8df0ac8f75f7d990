import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import bench, theta_amd
from theta_amd import search as S
ctx = theta_amd.default_context()
m, K, seed = 50, 6, 4242
r, rN, order = bench.synth(seed=seed, m=m, n=3, k=K)
best = S.do_optimization_single(3, m, K, 2, [0]*m, [K]*m, r, rN, 1.0, order, False, False)
Cw = np.asarray(best[0][0])[np.asarray(order)][:, 1:].astype(np.uint8)
print("optimum mu", best[0][1], best[0][2])
print("optimum C", Cw.T.tolist())
print("passes", S.last_report.mix["passes"])
p = theta_amd.Problem(ctx, 3, m, 2, r, rN, [0]*m, [K]*m, 1.0)
p.set_option("mix_dive_blend", 0.2)
props, st = p.mix_search(float("inf"), leaf_rel=1e-3, cap=256, dive=True)
pr = S.canonical_columns(props)
ok, mu, nll, _v = ctx.solve_batch(3, 2, r, rN, np.ascontiguousarray(pr), 1.0, want_vals=False)
j = int(np.nanargmin(np.where(ok > 0, nll, np.inf)))
print("dive mu", mu[j], nll[j])
print("dive C", pr[j].T.tolist())
print("diff rows", int((pr[j] != Cw).any(axis=1).sum()))
tr = np.array(r, float) / np.array(rN, float)
print("ratios", np.round(tr / tr.mean(), 3).tolist())
