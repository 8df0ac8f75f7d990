"""Round 6 probe: what a dive (THETA_MIX_DIVE) proposes, by beam width and ranking blend, against the known minima of configs 3 / 4 / 5-shape."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench, theta_amd
from theta_amd import search as S

KNOWN = {"c3": 22588904.807977, "c4": 23131607.563541, "c5": 137040271.350991}
ctx = theta_amd.default_context()
for name, m, K, seed in (("c3", 50, 4, 7), ("c4", 50, 6, 4242), ("c5", 200, 7, 55)):
    r, rN, order = bench.synth(seed=seed, m=m, n=3, k=K)
    p = theta_amd.Problem(ctx, 3, m, 2, r, rN, [0] * m, [K] * m, 1.0)
    lb, ub = [0] * m, [K] * m
    for leaf in (1e-3,):
        for beam in (512, 1024):
            for w in (0.1, 0.2, 0.35, 0.5):
                p.set_option("mix_beam", beam)
                p.set_option("mix_dive_blend", w)
                t0 = time.time()
                props, st = p.mix_search(float("inf"), leaf_rel=leaf, cap=256, dive=True)
                dt = time.time() - t0
                pr = S.canonical_columns(props)
                keep = pr[S.in_space_n3_batch(pr, lb, ub, 2)] if len(pr) else []
                best = None
                if len(keep):
                    ok, _mu, nll, _v = ctx.solve_batch(3, 2, r, rN, np.ascontiguousarray(np.asarray(keep, np.uint8)), 1.0, want_vals=False)
                    fin = [float(v) for v, o in zip(nll, ok) if o and v == v]
                    best = min(fin) if fin else None
                print("%s leaf %g beam %d blend %.2f: %d proposals, %d in space, best - known = %s (%.1f ms, %d boxes)" %
                      (name, leaf, beam, w, len(props), len(keep), "%.3f" % (best - KNOWN[name]) if best is not None else None, dt * 1e3, st["boxes_tested"]), flush=True)
    p.close()
