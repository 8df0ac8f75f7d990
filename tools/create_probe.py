import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, theta_amd
ctx = theta_amd.Context()
for m, K, seed in ((50, 6, 4242), (200, 7, 55)):
    r, rN, _ = bench.synth(seed=seed, m=m, n=3, k=K)
    for rep in range(3):
        t = time.time()
        p = theta_amd.Problem(ctx, 3, m, 2, r, rN, [0] * m, [K] * m, 1.0)
        t1 = time.time()
        p.close()
        print("m=%d create %.1f ms close %.1f ms" % (m, (t1 - t) * 1e3, (time.time() - t1) * 1e3), file=sys.stderr)
