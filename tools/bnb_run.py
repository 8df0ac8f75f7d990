"""Round 5: the branch and bound (theta_bnb + search of the surviving ranges) against the exhaustive search on whole small spaces,
and on BASELINE configs 3 / 4.   python tools/bnb_run.py [small|c3|c4|c5|c100] ..."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
import theta_amd
from theta_amd import search as S


def best_plain(best):
    return [(b[0].astype(int).tolist(), [float(x) for x in np.atleast_1d(b[1])], float(b[2])) for b in best]


def same(a, b):
    if len(a) != len(b):
        return "len %d vs %d" % (len(a), len(b))
    for i, (x, y) in enumerate(zip(a, b)):
        if x[0] != y[0]:
            return "C differs at %d" % i
        if not ((x[2] != x[2] and y[2] != y[2]) or abs(x[2] - y[2]) <= 1e-9 * abs(y[2])):
            return "nll differs at %d: %r %r" % (i, x[2], y[2])
    return ""


def run(m, K, seed, lb=None, ub=None, tau=2, bnb=True, sweep=False):
    r, rN, order = bench.synth(seed=seed, m=m, n=3, k=K)
    lb = [0] * m if lb is None else lb
    ub = [K] * m if ub is None else ub
    S.BNB_MIN_CANDIDATES = 0 if bnb else 2 ** 200
    S.NAN_SWEEP_MAX = 2 ** 33 if sweep else 0
    t = time.time()
    best = S.do_optimization_single(3, m, K, tau, list(lb), list(ub), r, rN, 1.0, order, False, False)
    return best_plain(best), time.time() - t, S.last_report


if __name__ == "__main__":
    what = sys.argv[1:] or ["small"]
    if "prof" in what:
        import cProfile, pstats
        run(12, 3, 31, bnb=True)
        pr = cProfile.Profile()
        pr.enable()
        b, tb, rb = run(14, 3, 9, bnb=True)
        pr.disable()
        print("bnb %.2fs" % tb, {k: v for k, v in rb.bnb.items() if k != "frontier"})
        pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
    if "small" in what:
        for m, K, seed in ((12, 3, 31), (14, 3, 9), (13, 4, 5), (16, 2, 3), (12, 5, 8), (18, 2, 4)):
            a, ta, ra = run(m, K, seed, bnb=False)
            b, tb, rb = run(m, K, seed, bnb=True)
            info = rb.bnb or {}
            print("m=%d K=%d seed=%d: space %.3g, exhaustive %.2fs, bnb %.2fs (plan %.2fs, %s ranges, %.3g leaves, %s nodes, lines complete %s) entries %d / %d  %s" %
                  (m, K, seed, ra.candidates, ta, tb, info.get("plan_seconds", -1), info.get("ranges"), info.get("leaves", 0), info.get("nodes"),
                   info.get("rank_deficient_complete"), len(a), len(b), same(a, b) or "IDENTICAL"), flush=True)
    for cfg, mm, K, seed in (("c3", 50, 4, 7), ("c4", 50, 6, 4242), ("c5", 200, 7, 55), ("c100", 100, 6, 11)):
        if cfg in what:
            t = time.time()
            b, tb, rb = run(mm, K, seed, bnb=True)
            print(cfg, "m=%d K=%d: %.2f s end to end, best NLL %.6f, %d entries" % (mm, K, tb, b[0][2], len(b)))
            print(json.dumps(rb.mix or rb.bnb, default=float))
            print("report: finalists %s fallback %s seconds %.2f" % (rb.finalists, rb.fallback_finalists, rb.seconds))
