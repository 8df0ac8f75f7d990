import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, theta_amd
ctx = theta_amd.Context(0)
rng = np.random.RandomState(3)
for n, m, B in ((3, 64, 1 << 22), (3, 200, 1 << 20), (2, 256, 1 << 21), (3, 50, 1 << 22), (3, 256, 1 << 20), (3, 202, 1 << 20), (3, 150, 1 << 20)):
    C = rng.randint(0, 7, (B, m, n - 1)).astype(np.uint8)
    if n == 2: C = C[:, :, 0]
    w = rng.randint(1000, 90000, m).astype(float); r = rng.randint(1000, 90000, m).astype(float)
    mu = rng.dirichlet(np.ones(n) * 3, B)
    ctx.score_masked(n, 2, C[:1024], w, r, mu[:1024], None)
    ms = min(ctx.score_masked(n, 2, C, w, r, mu, None)[1] for _ in range(4))
    print(n, m, B, "%.3f ms  %.2f TB/s" % (ms, B * (m * (n - 1) + 8 * n + 8) / ms / 1e9))
