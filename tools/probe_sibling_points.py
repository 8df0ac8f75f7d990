"""
Probe (round 5): how far apart are the optima of SIBLING candidates?  Dense witness of a mid-space range of the bench instance (tight
leg: the recorded mu is the optimum to 1e-12), the same ranks enumerated; for every candidate lambda^2 / sum r evaluated exactly
(tests/witness_check.py) at the MEAN mixture of (a) its siblings (same first m - 1 rows), (b) its cousins (same first m - 2 rows),
(c) its chunk of 768 consecutive ranks.  A shared SECOND evaluation at such a mean certifies a candidate only where that number is
below n3_conv_l2 (4.4e-8 on this instance).  Run: python tools/probe_sibling_points.py [log2 span]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import bench                      # noqa: E402
import theta_amd                  # noqa: E402
from witness_check import value_and_decrement   # noqa: E402


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 15
    ctx = theta_amd.default_context()
    rr, rn, _ = bench.synth()
    p = theta_amd.Problem(ctx, 3, bench.M, bench.TAU, rr, rn, [0] * bench.M, [bench.K_MAX] * bench.M, 1.0)
    for where, b in (("middle", p.count // 3), ("two thirds", 2 * (p.count // 3)), ("tenth", p.count // 10)):
        p.set_option("n3_no_dismiss", 1)
        p.set_option("n3_force_f64", 1)
        p.set_option("n3_conv_l2", 1e-12)
        rec, st = p.witness(b, b + (1 << lg), every_log2=0)
        C = p.enumerate(b, 1 << lg)
        ok = np.isin(rec["status"], (1, 2))
        mu = rec["mu"].astype(np.float64)
        key1 = np.concatenate([[True], np.any(C[1:, :-1] != C[:-1, :-1], axis=(1, 2))])
        key2 = np.concatenate([[True], np.any(C[1:, :-2] != C[:-1, :-2], axis=(1, 2))])
        g1 = np.cumsum(key1) - 1
        g2 = np.cumsum(key2) - 1
        g3 = np.arange(1 << lg) // 768
        print("== %s: %d candidates, %d solved, %d sibling groups (%.1f per group), %d cousin groups (%.1f)" %
              (where, 1 << lg, ok.sum(), g1.max() + 1, (1 << lg) / (g1.max() + 1), g2.max() + 1, (1 << lg) / (g2.max() + 1)))
        for name, g in (("siblings", g1), ("cousins", g2), ("chunk of 768", g3)):
            n = g.max() + 1
            w = ok.astype(np.float64)
            cnt = np.bincount(g, w, n)
            mean = np.stack([np.bincount(g, w * mu[:, j], n) for j in range(3)], axis=1) / np.maximum(cnt, 1)[:, None]
            at = mean[g]
            sel = ok & (cnt[g] > 1)
            _v, l2 = value_and_decrement(C[sel], rr, rn, at[sel], bench.TAU)
            q = np.nanquantile(l2, [0.1, 0.25, 0.5, 0.75, 0.9, 0.99])
            print("   l2 at the mean of the %-13s quantiles 10/25/50/75/90/99 %%: %s   below 4.4e-8: %.1f %%" %
                  (name, " ".join("%.1e" % x for x in q), 100.0 * np.mean(l2 < 4.4e-8)))
        print("   l2 at the kernel's shared first evaluation:                        %s" %
              " ".join("%.1e" % x for x in np.nanquantile(rec["l2_first"][ok].astype(np.float64), [0.1, 0.25, 0.5, 0.75, 0.9, 0.99])))
    p.close()


if __name__ == "__main__":
    main()
