"""GPU-side dump for the n=2 render generator: first differing records against the lane-stream kernel (run on the GPU box)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import theta_amd

ctx = theta_amd.default_context()
for m, k in ((25, 5), (7, 3), (130, 2), (50, 6)):
    p = theta_amd.Problem(ctx, 2, m, 2, [1] * m, [1] * m, [0] * m, [k] * m)
    cnt = int(min(p.count, 300000))
    os.environ["THETA_N2_ENUM_LEGACY"] = "1"
    old = p.enumerate(0, cnt)
    del os.environ["THETA_N2_ENUM_LEGACY"]
    os.environ["THETA_N2_ENUM_RENDER"] = "1"
    new = p.enumerate(0, cnt)
    del os.environ["THETA_N2_ENUM_RENDER"]
    o, n = old.reshape(cnt, m), new.reshape(cnt, m)
    badrec = np.nonzero((o != n).any(axis=1))[0]
    print("m=%d k=%d count=%d: %d differing records" % (m, k, cnt, len(badrec)))
    if len(badrec):
        fo, fn = o.reshape(-1), n.reshape(-1)
        badbytes = np.nonzero(fo != fn)[0]
        print("  differing bytes: %d, first at flat offsets %s" % (len(badbytes), badbytes[:24].tolist()))
        print("  offsets mod 128:", (badbytes[:24] % 128).tolist(), " mod 16:", (badbytes[:24] % 16).tolist())
        print("  records:", badrec[:16].tolist(), " record mod T candidates:", (badrec[:16] % 128).tolist())
        for rec in badrec[:4]:
            print("  rec %d old %s" % (rec, "".join("%x" % v for v in o[rec])))
            print("  rec %d new %s" % (rec, "".join("%x" % v for v in n[rec])))
        # distribution of (new - old) values
        d = (fn[badbytes].astype(int) - fo[badbytes].astype(int))
        vals, cts = np.unique(d, return_counts=True)
        print("  new-old histogram:", dict(zip(vals.tolist(), cts.tolist())))
    p.close()
