import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, theta_amd
ctx = theta_amd.Context(0); r, rN, order = bench.synth()
p = theta_amd.Problem(ctx, 3, 50, 2, r, rN, [0]*50, [6]*50, 1.0)
tot = p.count; batch = 1 << 27; nsteps = 10
stride = (tot - batch) // nsteps
for i in range(nsteps):
    b = i * stride
    res = p.search(b, b + batch, window=0.5); st = res["stats"]
    print(i, "kernel_ms %.1f setup %.2f acc %.3f iters %.2f ties %d suspects %d overflow %d best %.1f rej %.1f" % (
        st["kernel_ms"], st["setup_ms"], st["accepted"]/st["evaluated"], st["iterations"]/st["evaluated"], len(res["rank"]),
        len(p.last_suspects[0]), st["list_overflow"], st["best_nll"], st["rejected_bound"]))
