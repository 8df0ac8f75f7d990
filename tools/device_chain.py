"""
Device-resident chain of the materialised operators, run ON THE GPU BOX: theta_enumerate_device -> theta_score_masked_device
(CalcAllC.L3's contract on byte candidates, no mask) -> download of the NLLs only; and -> theta_solve_batch_device (the
reference's own per-candidate procedure).  Reports kernel times and HBM GB/s against SURVEY 8(d)'s algorithmic bytes:
generator m(n-1) B written per candidate; scorer m(n-1) + 8n B read + 8 B written per candidate.

    python tools/device_chain.py [log2 candidates, default 26] > gpurun_out/device_chain.json
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import bench
import theta_amd

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 26
B = 1 << lg
ctx = theta_amd.Context(0)
out = {"candidates": B}
for name, n, m, k in (("n3_m50_k6", 3, 50, 6), ("n2_m100_k5", 2, 100, 5)):
    r, rN, order = bench.synth(seed=77, m=m, n=n, k=k)
    p = theta_amd.Problem(ctx, n, m, 2, r, rN, [0] * m, [k] * m, 1.0)
    cnt = min(B, p.count)
    nc = n - 1
    d_C = ctx.device_array((cnt, m * nc), np.uint8)
    start = p.count // 3 if p.count > 3 * cnt else 0
    t_en = min(p.enumerate_device(start, cnt, d_C) for _ in range(3))
    mu = np.random.RandomState(1).dirichlet(np.ones(n) * 4, cnt)
    d_mu = ctx.device_array((cnt, n), np.float64).upload(mu)
    w, rr = np.asarray(rN, float), np.asarray(r, float)
    best = None
    for _ in range(3):
        d_nll, ms = ctx.score_masked_device(n, 2, d_C, cnt, m, w, rr, d_mu)
        best = ms if best is None else min(best, ms)
    nll = d_nll.download()[:, 0]
    # spot check against the host-buffer entry point
    sub = slice(0, 4096)
    C_host = p.enumerate(start, 4096)
    ref, _ms = ctx.score_masked(n, 2, C_host, w, rr, mu[sub])
    ok = bool(np.allclose(ref[:, 0], nll[sub], rtol=1e-13, equal_nan=True))
    e = {"candidates": cnt, "enumerate_ms": t_en, "enumerate_GBps": cnt * m * nc / (t_en * 1e-3) / 1e9,
         "score_ms": best, "score_bytes_per_candidate": m * nc + 8 * n + 8,
         "score_GBps": cnt * (m * nc + 8 * n + 8) / (best * 1e-3) / 1e9, "score_frac_of_8TBps": cnt * (m * nc + 8 * n + 8) / (best * 1e-3) / 8e12,
         "score_candidates_per_s": cnt / (best * 1e-3), "matches_host_entry_point": ok}
    # the reference's own procedure per candidate (hybrj / brenth restatement) on a slice of the same device buffer
    sb = min(cnt, 1 << 18)
    _ok, _mu, _nll, _v, ms_s = ctx.solve_batch_device(n, 2, r, rN, d_C, sb, m, 1.0)
    e["solve_batch_device_ms"] = ms_s
    e["solve_batch_candidates_per_s"] = sb / (ms_s * 1e-3)
    out[name] = e
    for a in (d_C, d_mu, d_nll, _ok, _mu, _nll):
        a.free()
    p.close()
print(json.dumps(out))
