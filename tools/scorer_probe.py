"""Masked scorer probe (config 5 shape): a few launches of theta_score_masked at m=200, n=3, S masks, B candidates."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import theta_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 16
S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
m = int(sys.argv[3]) if len(sys.argv) > 3 else 200
ctx = theta_amd.Context(0)
rng = np.random.RandomState(5)
w = rng.randint(1000, 90000, m).astype(float)
rr = rng.randint(1000, 90000, m).astype(float)
words = (m + 63) // 64
C = rng.randint(0, 8, (B, m, 2)).astype(np.uint8)
mu = rng.dirichlet(np.ones(3) * 3, B)
masks = rng.randint(0, 2 ** 63, (S, words), dtype=np.int64).astype(np.uint64)
ms = [ctx.score_masked(3, 2, C, w, rr, mu, masks)[1] for _ in range(4)]
best = min(ms)
print("B=%d S=%d m=%d: %.3f ms  %.2fe10 pairs/s  %.1f TFLOP/s f64 MFMA" % (B, S, m, best, B * S / best / 1e7, 2.0 * S * m * 2 * B / best / 1e9))
