import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, theta_amd
ctx = theta_amd.Context(0)
for m, k in ((50, 6), (20, 4), (12, 4)):
    r, rN, order = bench.synth(seed=11, m=m, n=3, k=k)
    p = theta_amd.Problem(ctx, 3, m, 2, r, rN, [0] * m, [k] * m, 1.0)
    cnt = min(p.count, 1 << 24)
    b = (p.count - cnt) // 3
    p.search(b, b + (1 << 16))
    t0 = time.time(); p.search(b, b + cnt); t1 = time.time() - t0
    p.set_option("n3_nan_sweep", 1)
    t0 = time.time(); p.search(b, b + cnt); t2 = time.time() - t0
    print("m=%d k=%d: %d candidates, search %.3f s, with the NaN sweep %.3f s -> sweep %.2e candidates/s, listed %d" % (m, k, cnt, t1, t2, cnt / (t2 - t1), len(p.last_degenerate[0])))
    p.close()
