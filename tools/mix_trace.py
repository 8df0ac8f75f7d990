"""One whole-space search of BASELINE config 4 (and config 5's shape) for a kernel trace:
   rocprofv3 --kernel-trace --stats -d gpurun_out/prof_mix -- python tools/mix_trace.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, theta_amd
from theta_amd import search as S
for name, m, K, seed in (("config 4", 50, 6, 4242), ("config 5 shape", 200, 7, 55)):
    r, rN, order = bench.synth(seed=seed, m=m, n=3, k=K)
    for rep in range(2):
        t0 = time.time()
        S.do_optimization_single(3, m, K, 2, [0] * m, [K] * m, r, rN, 1.0, order, False, False)
        print(name, "%.1f ms" % (1e3 * (time.time() - t0)), S.last_report.mix["kernel_ms"], S.last_report.mix["boxes_tested"], S.last_report.mix["levels"])
