"""Round 5: where the wall clock of a whole-space search goes (BASELINE config 3 and config 5's shape): theta_problem_create, the
mixture-space search, do_optimization_single end to end with a cProfile of the m = 200 case.  THETA_CREATE_DEBUG=1 adds the
phases of the creation (host tables, uploads, counting DP)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, theta_amd
from theta_amd import search as S
ctx = theta_amd.default_context()
for m, K, seed in ((50, 4, 7), (200, 7, 55), (200, 7, 55)):
    r, rN, order = bench.synth(seed=seed, m=m, n=3, k=K)
    t = time.time(); p = theta_amd.Problem(ctx, 3, m, 2, r, rN, [0]*m, [K]*m, 1.0); t1 = time.time() - t
    t = time.time(); c = p.count; t2 = time.time() - t
    t = time.time(); recs, st = S.mix_records(p, ctx, r, rN, 1.0, ([0]*m, [K]*m)); t3 = time.time() - t
    t = time.time(); p.close(); t4 = time.time() - t
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    t = time.time(); b = S.do_optimization_single(3, m, K, 2, [0]*m, [K]*m, r, rN, 1.0, order, False, False); t5 = time.time() - t
    pr.disable()
    print("m=%d: create %.3f count %.3f mix %.3f close %.3f | whole %.3f" % (m, t1, t2, t3, t4, t5))
    if m == 200:
        pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
