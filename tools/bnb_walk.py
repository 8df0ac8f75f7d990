"""exact walk only, against a given incumbent:  python tools/bnb_walk.py m K seed incumbent window budget"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, bench, theta_amd
from theta_amd import _lib
ctx = theta_amd.default_context()
m, K, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
inc = float(sys.argv[4])
r, rN, order = bench.synth(seed=seed, m=m, n=3, k=K)
p = theta_amd.Problem(ctx, 3, m, 2, r, rN, [0] * m, [K] * m, 1.0)
for w in sys.argv[5].split(","):
    try:
        t = time.time()
        rg2, st2 = p.bnb(inc + float(w), max_nodes=int(float(sys.argv[6])), cap=1 << 22)
        print("window %s: %d ranges, %.3g leaves, wall %.1f ms kernel %.1f ms, nodes %d, bounded %d, iters %d, max frontier %d" % (w, len(rg2), st2["leaves"], st2["wall_ms"], st2["kernel_ms"], st2["nodes_expanded"], st2["children_bounded"], st2["newton_iterations"], st2["max_frontier"]))
        print("frontier", st2["frontier"])
    except _lib.ThetaError as e:
        st2 = p.last_bnb
        print("window", w, "FAILED", str(e)[:120])
