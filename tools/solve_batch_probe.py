"""theta_solve_batch_device on batches whose candidates are NOT neighbours (random rows): the shape where hybrj's evaluation counts have a
long tail.  python tools/solve_batch_probe.py [m] [B]   (THETA_SOLVE_ONE_PASS=1: the one-pass kernel)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import theta_amd

m = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 18
ctx = theta_amd.Context()
r, rN, _ = bench.synth(seed=12, m=m, n=3, k=5)
rng = np.random.RandomState(1)
for what in ("random sorted columns", "random rows"):
    C = rng.randint(0, 6, (B, m, 2)).astype(np.uint8)
    if what.startswith("random sorted"):
        C = np.sort(C, axis=1)
    C = np.ascontiguousarray(C)
    ctx.solve_batch(3, 2, r, rN, C[:4096], 1.0, want_vals=False)
    best = None
    for _ in range(3):
        t = time.time()
        ok, mu, nll, _v = ctx.solve_batch(3, 2, r, rN, C, 1.0, want_vals=False)
        dt = time.time() - t
        best = dt if best is None else min(best, dt)
    print("%s: m=%d B=%d  %.1f ms end to end (uploads included)  %.3g candidates/s  outcomes %s  checksum %.17g" % (
        what, m, B, best * 1e3, B / best, np.bincount(ok, minlength=3).tolist(), float(np.nansum(nll))))
