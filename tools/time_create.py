"""Round 5: theta_problem_create alone on config 5's shape (m = 200, k = 7, full bounds), for rocprofv3 --kernel-trace --stats."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, theta_amd
ctx = theta_amd.default_context()
r, rN, order = bench.synth(seed=55, m=200, n=3, k=7)
for _ in range(2):
    t = time.time(); p = theta_amd.Problem(ctx, 3, 200, 2, r, rN, [0] * 200, [7] * 200, 1.0); print("create %.3f s" % (time.time() - t)); p.close()
