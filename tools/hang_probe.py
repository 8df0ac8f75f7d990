"""tools/hang_probe.py N: one search of 2^N candidates of the bench problem, counters printed (to be run under `timeout`)."""
import os, sys, faulthandler
faulthandler.dump_traceback_later(12, exit=False)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, theta_amd
n = int(sys.argv[1])
ctx = theta_amd.Context(0); r, rN, order = bench.synth()
p = theta_amd.Problem(ctx, 3, 50, 2, r, rN, [0]*50, [6]*50, 1.0)
for k, v in (("n3_force_f64", int(os.environ.get("F64", "0"))), ("n3_no_dismiss", int(os.environ.get("NODIS", "0")))):
    p.set_option(k, v)
print("created", flush=True)
res = p.search(0, 1 << n, window=0.5); st = res["stats"]
print(n, "kernel_ms %.2f evaluated %d iters %d best %.3f launches %s" % (st["kernel_ms"], st["evaluated"], st["iterations"], st["best_nll"], st.get("launches")), flush=True)
