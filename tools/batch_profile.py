"""Throughput of the materialised operators (Optimizer.solve / CalcAllC.L3 restated in the reference's arithmetic) on
B candidates from the generator.  Wall time includes the PCIe copies of the host-buffer ABI; run under
`rocprofv3 --kernel-trace --stats` for the kernels alone."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench, theta_amd
ctx = theta_amd.Context(0)
out = {}
for n, m, k, B in ((3, 50, 6, 1 << 18), (2, 50, 6, 1 << 18)):
    r, rN, order = bench.synth(seed=11, m=m, n=n, k=k)
    p = theta_amd.Problem(ctx, n, m, 2, r, rN, [0] * m, [k] * m, 1.0)
    Cs = p.enumerate(p.count // 3, B)
    ctx.solve_batch(n, 2, r, rN, Cs[:1024], 1.0)
    t0 = time.time(); ok, mu, nll, vals = ctx.solve_batch(n, 2, r, rN, Cs, 1.0); dt = time.time() - t0
    out["solve_batch_n%d_m%d" % (n, m)] = {"B": B, "wall_ms": dt * 1e3, "candidates_per_s_wall": B / dt, "accepted": float(np.mean(ok))}
    Cw = np.concatenate([np.full((B, m, 1), 2.0), Cs.reshape(B, m, n - 1).astype(float)], axis=2) * np.asarray(rN, float)[None, :, None]
    mus = np.where(np.isfinite(mu), mu, 1.0 / n)
    ctx.score_batch(n, Cw[:1024], mus[:1024], r)
    t0 = time.time(); ctx.score_batch(n, Cw, mus, r); dt = time.time() - t0
    out["score_batch_n%d_m%d" % (n, m)] = {"B": B, "wall_ms": dt * 1e3, "candidates_per_s_wall": B / dt}
print(json.dumps(out))
