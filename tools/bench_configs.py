#!/usr/bin/env python3
"""
Secondary measurements on one MI355X for the other BASELINE.json configs (bench.py reports the headline):
  config 2  m=25, n=2, k=5 exhaustive (142 506 candidates)   -- launch-bound by size
  n=2 large m=50, n=2, k=6 exhaustive (32 468 436 candidates)
  config 3  m=50, n=3, k=4 rank-range search
  config 5  m=200, n=3, k=7 masked scorer (B candidates x S masks): HBM GB/s against 8 TB/s
Prints one JSON object.
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import theta_amd


def timed_search(p, b, e, reps=3):
    p.search(b, e, window=0.5)
    best = None
    for _ in range(reps):
        t0 = time.time()
        res = p.search(b, e, window=0.5)
        dt = time.time() - t0
        if best is None or dt < best[0]:
            best = (dt, res["stats"])
    return best


def main():
    ctx = theta_amd.Context(0)
    out = {"device": ctx.name}
    for tag, m, k in (("config2_n2_m25_k5", 25, 5), ("n2_m50_k6", 50, 6), ("n2_m100_k5", 100, 5)):
        r, rN, order = bench.synth(seed=11, m=m, n=2, k=k)
        p = theta_amd.Problem(ctx, 2, m, 2, r, rN, [0] * m, [k] * m, 1.0)
        dt, st = timed_search(p, 0, p.count)
        out[tag] = {"candidates": p.count, "wall_ms": 1e3 * dt, "kernel_ms": st["kernel_ms"],
                    "candidates_per_s_wall": p.count / dt, "candidates_per_s_kernel": p.count / (st["kernel_ms"] * 1e-3),
                    "tflops_kernel": (st["flops"] + st["flops_f32"]) / (st["kernel_ms"] * 1e-3) / 1e12}
        p.close()
    r, rN, order = bench.synth(seed=12, m=50, n=3, k=4)
    p = theta_amd.Problem(ctx, 3, 50, 2, r, rN, [0] * 50, [4] * 50, 1.0)
    n = 1 << 27
    dt, st = timed_search(p, p.count // 3, p.count // 3 + n)
    out["config3_n3_m50_k4"] = {"candidates": n, "wall_ms": 1e3 * dt, "kernel_ms": st["kernel_ms"],
                                "candidates_per_s_kernel": n / (st["kernel_ms"] * 1e-3),
                                "tflops_kernel": (st["flops"] + st["flops_f32"]) / (st["kernel_ms"] * 1e-3) / 1e12}
    p.close()
    # config 5: masked scorer (FP64 MFMA GEMM: masks x per-candidate row terms)
    rng = np.random.RandomState(5)
    m, nn = 200, 3
    w = rng.randint(1000, 90000, m).astype(float)
    rr = rng.randint(1000, 90000, m).astype(float)
    words = (m + 63) // 64
    for B, S in ((1 << 16, 64), (1 << 14, 512), (1 << 17, 512)):
        C = rng.randint(0, 8, (B, m, 2)).astype(np.uint8)
        mu = rng.dirichlet(np.ones(3) * 3, B)
        masks = rng.randint(0, 2 ** 63, (S, words), dtype=np.int64).astype(np.uint64)
        ctx.score_masked(nn, 2, C[:1024], w, rr, mu[:1024], masks)
        ms = min(ctx.score_masked(nn, 2, C, w, rr, mu, masks)[1] for _ in range(5))
        algo_bytes = B * (m * 2 + 8 * nn) + B * S * 8 + S * words * 8      # candidates + mu read, NLL written, masks
        flops = 2.0 * S * m * 2 * B                                        # the two masked sums as a GEMM
        out["config5_scorer_m200_k7_S%d_B%d" % (S, B)] = {
            "pairs": B * S, "kernel_ms": ms, "pairs_per_s": B * S / (ms * 1e-3),
            "algorithmic_GBps": algo_bytes / (ms * 1e-3) / 1e9, "hbm_peak_GBps": 8000.0,
            "mfma_fp64_tflops": flops / (ms * 1e-3) / 1e12, "mfma_fp64_peak_tflops": 78.6}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
