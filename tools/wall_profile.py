"""Where the host time of do_optimization_single goes on configs 4 / 5-shape (cProfile, cumulative)."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, theta_amd
from theta_amd import search as S
for name, m, K, seed in (("config 4", 50, 6, 4242), ("config 5 shape", 200, 7, 55)):
    r, rN, order = bench.synth(seed=seed, m=m, n=3, k=K)
    S.do_optimization_single(3, m, K, 2, [0] * m, [K] * m, r, rN, 1.0, order, False, False)
    pr = cProfile.Profile()
    pr.enable()
    S.do_optimization_single(3, m, K, 2, [0] * m, [K] * m, r, rN, 1.0, order, False, False)
    pr.disable()
    out = io.StringIO()
    pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(22)
    print("=====", name)
    print("\n".join(l[:150] for l in out.getvalue().splitlines()[4:40]))
