import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, theta_amd
ctx = theta_amd.Context(0)
g = np.load("/root/repo/tests/golden/solve_n3_m6k3.npz")
C = g["C"]; r = g["r"]; rN = g["rN"]
ok, mu_b, nll_b, _ = ctx.solve_batch(3, 2, r, rN, C, 1.0)
fb = ctx.last_solve_fallback
acc = g["accepted"].astype(bool); rn = g["nll"]
B, m, _ = C.shape
full = np.concatenate([np.full((B, m, 1), 2.0), C.astype(float)], axis=2)
Cw = full * rN.astype(float)[None, :, None]; S = Cw.sum(1)
with np.errstate(all="ignore"):
    Ch = Cw / S[:, None, :]; F = -(r.astype(float)[None, :] * np.log(Ch.sum(2) / 3.0)).sum(1)
    ref_fb = acc & (np.abs(F - rn) <= 1e-9 * np.abs(rn))
    same = ok & acc & (np.abs(nll_b - rn) <= 1e-9 * np.abs(rn))
print("ref: acc %d, fallback-valued %d, None %d, NaN-acc %d" % (acc.sum(), ref_fb.sum(), (~acc).sum(), (acc & np.isnan(rn)).sum()))
print("gpu: ok=1 %d, fallback %d, none %d" % ((ok & ~fb).sum(), fb.sum(), (~ok).sum()))
print("ref fallback & gpu fallback:", (ref_fb & fb).sum(), " ref fallback & gpu inside:", (ref_fb & ok & ~fb).sum(), " ref fallback & gpu none:", (ref_fb & ~ok).sum())
ref_in = acc & ~ref_fb & np.isfinite(rn)
print("ref inside-valued & gpu inside same:", (ref_in & ok & ~fb & same).sum(), " & gpu inside differ:", (ref_in & ok & ~fb & ~same).sum(), " & gpu fallback:", (ref_in & fb).sum(), " & gpu none:", (ref_in & ~ok).sum())
print("ref None & gpu inside:", (~acc & ok & ~fb).sum(), " ref None & gpu fallback:", (~acc & fb).sum(), " ref None & gpu none:", (~acc & ~ok).sum())
x = ref_fb & ok & ~fb
with np.errstate(all="ignore"):
    print("ref-fallback/gpu-inside: gpu nll lower by (rel) min/median:", np.min((rn[x]-nll_b[x])/rn[x]), np.median((rn[x]-nll_b[x])/rn[x]))
    nu = mu_b * S; nu = nu / nu.sum(1, keepdims=True)
    print("  gpu nu min over those (how close to the boundary): quantiles", np.quantile(nu[x].min(1), [0, .1, .5, .9, 1]))
    rk = np.array([np.linalg.matrix_rank(np.column_stack([np.ones(m), c[:, 0], c[:, 1]])) for c in C[x].astype(float)])
    print("  rank histogram of those:", np.bincount(rk))
    y = ref_in & ok & ~fb & same
    print("  gpu nu min over agreeing inside entries: quantiles", np.quantile(nu[y].min(1), [0, .1, .5, .9, 1]))
