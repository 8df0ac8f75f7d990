import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, theta_amd, numpy as np
ctx = theta_amd.Context()
r, rN, _ = bench.synth()
p = theta_amd.Problem(ctx, 3, bench.M, bench.TAU, r, rN, [0] * bench.M, [bench.K_MAX] * bench.M, 1.0)
opts = {"n3_no_dismiss": 1, "n3_force_f64": 1, "n3_conv_l2": bench.certified_conv_l2(r), "n3_mu_tol": 1e-6}
for k, v in opts.items():
    p.set_option(k, v)
total = p.count
b0 = int(total * 0.96)          # (about the bench's last stretch)
probe = p.search(b0, b0 + (1 << 16), window=0.0)
known = 23131607.56354141      # (the space's minimum: profiles/r6/bench_n1.json, wall_clock_to_best.config4)
for span_log, off in ((24, 0), (24, 1 << 30), (26, 0), (28, 0), (31, 0)):
    for pt in (0,):
        p.set_option("n3_per_task", pt)
        if known is not None:
            p.hint(known)
        try:
            f = p.search(b0 + off, b0 + off + (1 << span_log), window=0.0)
        except Exception as ex:
            print('span 2^%d: %s' % (span_log, str(ex)[:80]))
            continue
        if len(f['nll']):
            known = min(known, float(f['nll'].min())) if known is not None else float(f['nll'].min())
        st = f["stats"]
        print("span 2^%d offset %d per_task %d: evaluations per candidate %.3f  kernel ms %.2f" % (span_log, off, pt, st["iterations"] / st["evaluated"], st.get("kernel_ms", 0.0)))
