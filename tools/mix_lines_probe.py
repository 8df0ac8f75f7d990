"""Round 6 probe: the mixture-space search with the lines' trees (rank-deficient matrices) on small whole spaces -- against the
linear walk's complete records -- and on BASELINE configs 3 / 4 / 5's shape.   python tools/mix_lines_probe.py [small|configs|all]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import bench
import campaign
import theta_amd
from theta_amd import search as S
from conftest import rank_deficient


def plain(best):
    return [(np.asarray(t["c"]).tolist(), [float(x) for x in t["mu"]], float(t["nll"])) for t in best]


def small(ctx, m, K, seed):
    r, rN, order = bench.synth(seed=seed, m=m, n=3, k=K)
    lb, ub = [0] * m, [K] * m
    p = theta_amd.Problem(ctx, 3, m, 2, r, rN, lb, ub, 1.0)
    p.set_option("mix_max_boxes", 5e8)
    rep = S.SearchReport()
    t0 = time.time()
    recs, _ = S.mix_records(p, ctx, r, rN, 1.0, (lb, ub), report=rep)
    t1 = time.time()
    p.set_option("n3_nan_sweep", 0)
    walk, _st = S.collect_finalists(p, ctx, r, rN, 1.0, 0, p.count)
    walk = walk + S.fallback_records(p, ctx, r, rN, 1.0, walk)
    listed = set(p.last_degenerate[0])
    walk = [t for t in walk if t["rank"] not in listed] + S.degenerate_records(p, ctx, r, rN, 1.0, recs=walk)
    t2 = time.time()
    fin = lambda rc: [t for t in rc if t["nll"] == t["nll"]]
    want, got = S.replay_records(fin(walk), False), S.replay_records(fin(recs), False)
    why = campaign.compare_best(plain(got), plain(want), tol=1e-9)
    ndef = len(p.last_degenerate[0])
    k = rep.mix
    print("m=%d K=%d seed=%d: %.3g matrices (%d rank deficient), mix %.3f s (walk %.2f s): %s | boxes %d, lines %d, leaves %d (%d of lines), listed %d, "
          "records %d (%d rank deficient), syncs %s, bound %.3f thr %.3f min %.3f" %
          (m, K, seed, p.count, ndef, t1 - t0, t2 - t1, "IDENTICAL" if why == "" else "DIFFER: " + why, k["boxes_tested"], k["lines"], k["leaves"], k["line_leaves"],
           k["listed"], k["records"], k["rank_deficient_records"], k["syncs"], k["rank_deficient_bound"], k["threshold"], k["minimum"]), flush=True)
    p.close()
    return why == ""


def config(ctx, name, m, K, seed):
    r, rN, order = bench.synth(seed=seed, m=m, n=3, k=K)
    for rep_i in range(2):
        t0 = time.time()
        best = S.do_optimization_single(3, m, K, 2, [0] * m, [K] * m, r, rN, 1.0, order, False, False)
        dt = time.time() - t0
        k = S.last_report.mix
        print("%s: %.3f s end to end; dive %s; heuristic %s s; passes %s; final: boxes %d leaves %d (%d of %d lines) listed %d records %d kernel %.1f ms search %.1f ms syncs %s; "
              "best %.6f; rank-deficient bound %.3f (threshold %.3f)" %
              (name, dt, k.get("dive"), k.get("heuristic_seconds"), [(q.get("leaf"), round(q.get("ms", 0), 1)) for q in k["passes"]], k["boxes_tested"], k["leaves"], k["line_leaves"], k["lines"],
               k["listed"], k["records"], k["kernel_ms"], k["search_ms"], k["syncs"], best[0][2], k["rank_deficient_bound"], k["threshold"]), flush=True)


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    ctx = theta_amd.default_context()
    if what in ("small", "all"):
        ok = True
        for m, K, seed in [(12, 3, 31), (10, 6, 3), (12, 4, 5), (11, 5, 8)]:
            ok = small(ctx, m, K, seed) and ok
        print("small spaces:", "all identical" if ok else "DIFFERENCES")
    if what in ("configs", "all"):
        config(ctx, "config 3", 50, 4, 7)
        config(ctx, "config 4", 50, 6, 4242)
        config(ctx, "config 5 shape", 200, 7, 55)


if __name__ == "__main__":
    main()
