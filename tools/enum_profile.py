"""theta_enumerate_device throughput (materialised generator into HBM): 2 m (n=3) / m (n=2) bytes written per candidate."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, theta_amd, numpy as np
ctx = theta_amd.Context(0)
out = {}
for name, n, m, k, cnt in (("n3_m50_k6", 3, 50, 6, 1 << 28), ("n3_m50_k4", 3, 50, 4, 1 << 28), ("n3_m64_k3", 3, 64, 3, 1 << 28), ("n3_m25_k4", 3, 25, 4, 1 << 28), ("n3_m49_k5", 3, 49, 5, 1 << 28),
                           ("n2_m50_k6", 2, 50, 6, 1 << 24), ("n2_m100_k5", 2, 100, 5, 1 << 26)):
    r, rN, order = bench.synth(seed=11, m=m, n=n, k=k)
    p = theta_amd.Problem(ctx, n, m, 2, r, rN, [0] * m, [k] * m, 1.0)
    cnt = int(min(cnt, p.count))
    b = (p.count - cnt) // 3
    nbytes = cnt * m * (n - 1)
    buf = ctx.device_array((nbytes,), np.uint8)          # (HBM owned by the caller: theta_device_alloc, no torch)
    p.enumerate_device(b, min(cnt, 1 << 16), buf)
    ms = min(p.enumerate_device(b, cnt, buf) for _ in range(3))
    chk = p.enumerate(b + cnt - 5, 5).reshape(-1)
    tail = np.zeros(chk.size, np.uint8)
    theta_amd._lib._check(theta_amd._lib.load().theta_device_copy(ctx._h, tail.ctypes.data, buf.ptr.value + nbytes - chk.size, chk.size, 0))
    assert np.array_equal(tail, chk)
    buf.free()
    out[name] = {"candidates": cnt, "bytes": nbytes, "kernel_ms": ms, "candidates_per_s": cnt / ms * 1e3, "GBps": nbytes / ms / 1e6,
                 "hbm_frac": nbytes / ms / 1e6 / 8000.0}
    print(name, out[name], file=sys.stderr)
print(json.dumps(out))
