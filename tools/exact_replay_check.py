"""The search against the per-candidate procedure at a scale the CPU oracle cannot reach, run ON THE GPU BOX.
For whole seeded spaces of 1e5 .. 4e6 matrices: EVERY candidate through theta_solve_batch (the reference's own per-candidate
procedure, restated and checked entry by entry against the reference's tables) and the reference's sequential rule replayed over
all of those outcomes -- against `best` of the shipped driver (sieve + finish kernels, suspects / fallback records, rank-deficient
list, NaN sweep, replay over the finalists only).  Complete lists, entry by entry.

    python tools/exact_replay_check.py [instances per shape] [max candidates]
    python tools/exact_replay_check.py bench [ranges] [candidates per range]     # rank ranges of the bench's own m=50, K=6 space
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import campaign
import theta_amd
import theta_amd.search as S


def exact_best(ctx, inst, window, n=3):
    """`best` from the outcomes of ALL candidates (no search kernels involved)."""
    m, tau = inst["m"], inst["tau"]
    p = theta_amd.Problem(ctx, n, m, tau, inst["r"], inst["rN"], inst["lb"], inst["ub"], inst["mx"])
    keep_rank, keep_C = [], []
    lowest = np.inf
    chunks = []
    for b in range(0, p.count, 1 << 19):
        C = p.enumerate(b, min(1 << 19, p.count - b))
        ok, _mu, nll, _ = ctx.solve_batch(n, tau, inst["r"], inst["rN"], C, inst["mx"], want_vals=False)
        rep = ok != 0
        fin = rep & ~np.isnan(nll)
        if fin.any():
            lowest = min(lowest, float(nll[fin].min()))
        chunks.append((b, C, rep, nll))
    for b, C, rep, nll in chunks:
        with np.errstate(invalid="ignore"):
            sel = rep & (np.isnan(nll) | (nll <= lowest + window))
        idx = np.nonzero(sel)[0]
        keep_rank += (b + idx).tolist()
        keep_C.append(C[idx])
    count = p.count
    p.close()
    recs = []
    if keep_rank:
        Cs = np.concatenate(keep_C)
        ok, mu, nll, vals = ctx.solve_batch(n, tau, inst["r"], inst["rN"], Cs, inst["mx"], want_vals=True)
        recs = [{"rank": keep_rank[i], "c": Cs[i], "mu": mu[i].copy(), "nll": float(nll[i]), "vals": vals[i].copy()}
                for i in range(len(keep_rank)) if ok[i]]
    q1 = S._q1_record(ctx, n, m, tau, inst["r"], inst["rN"], inst["mx"]) if n == 3 else None
    return S.replay_ties(recs, n, tau, inst["order"], first_duplicate=(n == 2), q1_first=q1), count


def exact_range(ctx, p, inst, begin, end, window):
    """replay over the outcomes of the candidates of the rank range [begin, end) (no quirk-Q1 matrix: a sub-range)"""
    n, tau = 3, inst["tau"]
    lowest, chunks = np.inf, []
    for b in range(begin, end, 1 << 20):
        C = p.enumerate(b, min(1 << 20, end - b))
        ok, _mu, nll, _ = ctx.solve_batch(n, tau, inst["r"], inst["rN"], C, inst["mx"], want_vals=False)
        rep = ok != 0
        fin = rep & ~np.isnan(nll)
        if fin.any():
            lowest = min(lowest, float(nll[fin].min()))
        with np.errstate(invalid="ignore"):
            near = rep & (np.isnan(nll) | (nll <= lowest + 2 * window))      # (lowest only falls: a superset of the final selection)
        idx = np.nonzero(near)[0]
        chunks.append(([b + int(i) for i in idx], C[idx], nll[idx]))          # (ranks beyond 64 bits: python ints)
    recs = []
    for ranks, Cs, nl in chunks:
        with np.errstate(invalid="ignore"):
            sel = np.isnan(nl) | (nl <= lowest + window)
        if sel.any():
            Cs, ranks = Cs[sel], [rk for rk, t in zip(ranks, sel) if t]
            ok, mu, nll, vals = ctx.solve_batch(n, tau, inst["r"], inst["rN"], Cs, inst["mx"], want_vals=True)
            recs += [{"rank": int(ranks[i]), "c": Cs[i], "mu": mu[i].copy(), "nll": float(nll[i]), "vals": vals[i].copy()}
                     for i in range(len(ranks)) if ok[i]]
    return S.replay_ties(recs, n, tau, inst["order"], first_duplicate=False, q1_first=None)


def bench_ranges(ctx, nranges, size):
    """the bench's own instance (m = 50, K = 6, full bounds: the sieve at its full depth): rank ranges of `size` candidates spread
    over the space, the driver's per-shard pipeline (theta_amd.search._search_local) against the replay over every outcome"""
    import bench
    r, rN, order = bench.synth()
    m, k = bench.M, bench.K_MAX
    inst = dict(m=m, k=k, tau=bench.TAU, mx=1.0, r=r, rN=rN, order=order, lb=[0] * m, ub=[k] * m)
    S.NAN_SWEEP_MAX = 1 << 200                        # (the ranges are swept although the space is not)
    bad = 0
    for j in range(nranges):
        probe = theta_amd.Problem(ctx, 3, m, bench.TAU, r, rN, inst["lb"], inst["ub"], 1.0)
        count = probe.count
        G = count // size
        g = (G * (2 * j + 1)) // (2 * nranges)
        begin, end = count * g // G, count * (g + 1) // G
        rep = S.SearchReport()
        problem, _ctx, recs, _stats = S._search_local(3, m, bench.TAU, inst["lb"], inst["ub"], r, rN, 1.0, shard=(g, G), ctx=ctx, report=rep)
        problem.close()
        got = S.replay_ties(recs, 3, bench.TAU, order, first_duplicate=False, q1_first=None)
        ref = exact_range(ctx, probe, inst, begin, end, rep.window)
        probe.close()
        why = campaign.compare_best(campaign.best_to_plain(got), campaign.best_to_plain(ref))
        print("bench range %d: ranks [%d, +%d): %d entries%s" % (j, begin, end - begin, len(ref), (" DIFFERS: " + why) if why else ""))
        bad += bool(why)
    return bad


def wide_spaces(ctx, want, cap):
    """n=3 with more than 64 intervals (two rows per lane in the sieve, the wide generator): tests/test_gpu_wide.py's instances with
    more free rows, whole spaces of 1e4 .. cap matrices"""
    import test_gpu_wide as tw
    bad = done = tot = 0
    seed = 600
    while done < want and seed < 600 + 60 * want:
        seed += 1
        m = (72, 100, 128, 90)[seed % 4]
        r, rN, order, _truth, lb, ub = tw._wide_instance(m, seed, 2 + seed % 3, kmax=3)
        inst = dict(n=3, m=m, k=int(max(ub)), tau=2, mx=1.0, r=r, rN=rN, order=order, lb=[int(v) for v in lb], ub=[int(v) for v in ub])
        cnt = campaign.count_candidates(inst)
        if not (10_000 <= cnt <= cap):
            continue
        done += 1
        gpu = S.do_optimization_single(3, m, inst["k"], 2, inst["lb"], inst["ub"], r, rN, 1.0, order)
        ref, count = exact_best(ctx, inst, S.last_report.window)
        why = campaign.compare_best(campaign.best_to_plain(gpu), campaign.best_to_plain(ref))
        tot += count
        if why:
            bad += 1
            print("DIFFERS: wide m=%d seed %d (%d matrices): %s" % (m, seed, count, why))
    print("wide instances %d, candidates %d, lists that differ %d" % (done, tot, bad))
    return bad


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "wide":
        return 1 if wide_spaces(theta_amd.Context(0), int(sys.argv[2]) if len(sys.argv) > 2 else 8, int(float(sys.argv[3])) if len(sys.argv) > 3 else 4_000_000) else 0
    if len(sys.argv) > 1 and sys.argv[1] == "bench":
        return 1 if bench_ranges(theta_amd.Context(0), int(sys.argv[2]) if len(sys.argv) > 2 else 6, int(float(sys.argv[3])) if len(sys.argv) > 3 else 1 << 24) else 0
    want = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    cap = int(float(sys.argv[2])) if len(sys.argv) > 2 else 4_000_000
    ctx = theta_amd.Context(0)
    tot = inst_n = bad = nan_entries = narrowed = lost = refused = 0
    shapes = ((3, "mid"), (3, "low"), (2, "synth"))
    if len(sys.argv) > 3 and sys.argv[3] == "toy":         # (small spaces, m = 4..7: the fused kernel's path)
        shapes = ((3, "toy"), (2, "toy"), (2, "mid"))
    if len(sys.argv) > 3 and sys.argv[3] in ("amp", "ampw"):   # (copy numbers above 7: bounds of the reference's heuristic, the compact alphabet;
        shapes = ((3, sys.argv[3]),)                            #  ampw: 8-10 intervals, the sieve path)
    for n, shape in shapes:
        seed, got = int(os.environ.get("EXACT_SEED0", 30000)), 0
        while got < want:
            seed += 1
            if seed > int(os.environ.get("EXACT_SEED0", 30000)) + 60 * want:
                break                                   # (few instances of this shape are this large)
            if shape == "synth":                        # n=2: the campaign shapes are tiny; the bench's generator with full bounds
                import bench
                rng = np.random.RandomState(seed)
                m, k = int(rng.randint(18, 45)), int(rng.randint(3, 7))
                r, rN, order = bench.synth(seed=seed, m=m, n=2, k=k)
                inst = dict(seed=seed, n=2, m=m, k=k, tau=2, mx=float(rng.choice([1.0, 1.0, 0.6])), r=r, rN=rN, order=order, lb=[0] * m, ub=[k] * m, shape=shape)
            else:
                inst = campaign.instance(seed, n, shape)
            cnt = campaign.count_candidates(inst)
            if shape in ("amp", "ampw") and max(inst["ub"]) < 8:
                continue
            if not ((200 if shape in ("toy", "amp", "ampw") or (n == 2 and shape == "mid") else 100_000 if n == 3 else 20_000) <= cnt <= cap):
                continue
            got += 1
            if os.environ.get("EXACT_TAIL") and n == 3:     # (the whole-space NaN sweep off: only the tail behind the first entry of best is swept)
                S.NAN_SWEEP_MAX = int(cnt) - 1
            try:
                gpu = S.do_optimization_single(n, inst["m"], inst["k"], inst["tau"], list(inst["lb"]), list(inst["ub"]), inst["r"], inst["rN"],
                                               inst["mx"], inst["order"])
            except SystemExit:
                gpu = []
            except theta_amd.ThetaError as e:
                # (a flat likelihood -- a few reads per interval -- can put a million rejected candidates within the narrowest window of
                # the minimum: without the whole-space sweep, whose listed outcomes stand in for them, the driver refuses rather than
                # return an incomplete list)
                refused += 1
                print("refused: n=%d shape %s seed %d: %s" % (n, shape, seed, str(e)[:120]))
                continue
            ref, count = exact_best(ctx, inst, S.last_report.window, n)      # (the window the driver ended up with: narrowed on flat likelihoods)
            g_plain, r_plain = campaign.best_to_plain(gpu), campaign.best_to_plain(ref)
            if S.NAN_SWEEP_MAX == 0:                    # (THETA_NAN_SWEEP_MAX=0: what is lost without the sweep?  NaN tuples only)
                lost += len([b for b in r_plain if b[2] != b[2]]) - len([b for b in g_plain if b[2] != b[2]])
                g_plain, r_plain = [b for b in g_plain if b[2] == b[2]], [b for b in r_plain if b[2] == b[2]]
            why = campaign.compare_best(g_plain, r_plain)
            tot += count
            inst_n += 1
            nan_entries += sum(1 for b in ref if b[2] != b[2])
            narrowed += S.last_report.window < S.COLLECT_WINDOW
            if why:
                bad += 1
                print("DIFFERS: n=%d shape %s seed %d (%d matrices): %s" % (n, shape, seed, count, why))
    print("instances %d, candidates %d, NaN entries in the exact lists %d, searches that narrowed their window %d, refused %d, lists that differ %d"
          % (inst_n, tot, nan_entries, narrowed, refused, bad))
    if S.NAN_SWEEP_MAX == 0:
        print("without the sweep: NaN tuples missing from the driver's lists %d (finite entries compared above)" % lost)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
