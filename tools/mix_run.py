"""Round 5: branch and bound over the mixture space (theta_mix_search).   python tools/mix_run.py m K seed [leaf_rel] [window]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, bench, theta_amd
from theta_amd import _lib, search as S
ctx = theta_amd.default_context()
m, K, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
leaf = float(sys.argv[4]) if len(sys.argv) > 4 else 2e-4
window = float(sys.argv[5]) if len(sys.argv) > 5 else 0.5
r, rN, order = bench.synth(seed=seed, m=m, n=3, k=K)
lb, ub = [0] * m, [K] * m
p = theta_amd.Problem(ctx, 3, m, 2, r, rN, lb, ub, 1.0)
print("space %.3g" % p.count)
t = time.time()
hv, hC = S.heuristic_incumbent(ctx, m, 2, lb, ub, r, rN, 1.0)
print("heuristic incumbent %.6f in %.2f s" % (hv, time.time() - t))
t = time.time()
try:
    mats, st = p.mix_search(hv + window, leaf_rel=leaf, cap=1 << 18)
except _lib.ThetaError as e:
    print("FAILED", e, p.last_mix)
    sys.exit(0)
print("mix_search: %d matrices, %.2f s" % (len(mats), time.time() - t), st)
ins = np.array([S.in_space_n3(M, lb, ub, 2) for M in mats], bool)
print("in the reference's space:", int(ins.sum()))
if ins.any():
    ok, mu, nll, _ = ctx.solve_batch(3, 2, r, rN, np.ascontiguousarray(mats[ins]), 1.0, want_vals=False)
    o = np.argsort(np.where(ok > 0, nll, np.inf))
    for j in o[:8]:
        print("  ok %d nll %.6f mu %s" % (ok[j], nll[j], np.round(mu[j], 6)), mats[ins][j].T.tolist() if m <= 20 else "")
    print("within window of the best:", int(((ok > 0) & (nll <= np.nanmin(np.where(ok > 0, nll, np.inf)) + window)).sum()))
if p.count < 3e8:
    S.BNB_MIN_CANDIDATES = 2 ** 200
    S.NAN_SWEEP_MAX = 0
    t = time.time()
    best = S.do_optimization_single(3, m, K, 2, list(lb), list(ub), r, rN, 1.0, order, False, False)
    fin = [b for b in best if b[2] == b[2]]
    print("exhaustive: %d entries (%d finite) in %.2f s; best NLL %.6f" % (len(best), len(fin), time.time() - t, min(b[2] for b in fin)))
