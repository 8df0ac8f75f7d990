"""Where does the wall time of a search on BASELINE config 5's shape (m=200, n=3, k=7, full bounds) go?  The bench's rider
reports 0.30 s of wall for 0.09 s of kernel per 2^30 candidates: this prints, per theta_search call, wall / kernel / set-up time
and the counters that send a call off the fast path (contenders, redone slices, tasks with collinear prefixes, rank-deficient
candidates).  Ranges at several ranks of the space: low ranks have long constant prefixes."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
import theta_amd

ctx = theta_amd.default_context()
T = {}


def timed(cls, name):
    f = getattr(cls, name)

    def g(*a, **k):
        t0 = time.time()
        try:
            return f(*a, **k)
        finally:
            T[name] = T.get(name, 0.0) + time.time() - t0
    setattr(cls, name, g)


for nm in ("_probe", "_piece", "_search_once", "suspects", "degenerate"):
    if hasattr(theta_amd.Problem, nm):
        timed(theta_amd.Problem, nm)
r, rN, order = bench.synth(seed=55, m=200, n=3, k=7)
p = theta_amd.Problem(ctx, 3, 200, bench.TAU, r, rN, [0] * 200, [7] * 200, 1.0)
span = 1 << int(os.environ.get("SPAN_LOG2", "30"))
for lg in (100, 120, 127):
    b = 1 << lg
    res = p.search(b, b + (1 << 22), window=0.5)
    for rep in range(3):
        if len(res["nll"]):
            p.hint(float(res["nll"].min()))
        T.clear()
        t0 = time.time()
        res = p.search(b, b + span, window=0.5)
        dt = time.time() - t0
        st = res["stats"]
        print("rank 2^%d rep %d: wall %.1f ms, kernel %.1f, setup %.2f, redo %.1f, launches %d, survivors %d, fallback %d, degenerate %d, "
              "dismissed %.4f, finalists %d, deg list %d; host ms %s" % (lg, rep, 1e3 * dt, st["kernel_ms"], st["setup_ms"], st["redo_kernel_ms"], st["kernel_launches"],
                                                st["survivors"], st["fallback_candidates"], st["degenerate"], st["dismissed"] / span, len(res["rank"]),
                                                len(p.last_degenerate[0]), {k: round(1e3 * v, 1) for k, v in T.items()}), flush=True)
p.close()
