"""Burst generator probe: theta_enumerate_device at several request sizes (run under rocprofv3 --kernel-trace --stats for the split)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, theta_amd
ctx = theta_amd.Context(0)
shapes = [(50, 6), (50, 4)] if len(sys.argv) < 3 else [(int(sys.argv[1]), int(sys.argv[2]))]
sizes = (24, 26, 28) if len(sys.argv) < 4 else (int(sys.argv[3]),)
for m, k in shapes:
    r, rN, order = bench.synth(seed=11, m=m, n=3, k=k)
    p = theta_amd.Problem(ctx, 3, m, 2, r, rN, [0] * m, [k] * m, 1.0)
    for lg in sizes:
        cnt = 1 << lg
        b = (p.count - cnt) // 3
        nbytes = cnt * m * 2
        buf = torch.empty(nbytes, dtype=torch.uint8, device="cuda:0")
        if len(sizes) > 1:
            p.enumerate_device(b, 1 << 16, buf.data_ptr())
        ms = min(p.enumerate_device(b, cnt, buf.data_ptr()) for _ in range(3))
        print("m=%d k=%d 2^%d: %.3f ms  %.2fe10 cand/s  %.2f TB/s" % (m, k, lg, ms, cnt / ms * 1e3 / 1e10, nbytes / ms / 1e9), flush=True)
        del buf
    p.close()
