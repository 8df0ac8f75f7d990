"""
GPU tests of the two branch-and-bound searches of round 5 (`-m gpu`, through the C ABI; round 4's verdict, Next 3):

  * theta_mix_search (branch and bound over the MIXTURE space, csrc/bnb.hip) behind do_optimization_single for spaces no linear
    walk finishes: against the exhaustive search on whole seeded spaces, against `best` lists written by the reference itself
    (tests/golden/best_campaign*.json), and on BASELINE configs 3 and 4 themselves (m = 50: 4e27 / 2.6e38 matrices) with a
    tight-bounds instance around the optimum searched exhaustively as the cross-check;
  * theta_bnb (branch and bound over the row tree, rank ranges handed to theta_search_ranges): complete `best` lists, rank-deficient
    entries included, identical to the exhaustive search's on whole small spaces.
"""
import time

import numpy as np
import pytest

import campaign
from conftest import load_json, rank_deficient

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import theta_amd
    return theta_amd.default_context()


def _records_exhaustive(S, problem, ctx, r, rN):
    """finalists + nu = 1/3 fallbacks of the WHOLE space, by the linear walk (no NaN sweep, no rank-deficient list)"""
    problem.set_option("n3_nan_sweep", 0)
    recs, _st = S.collect_finalists(problem, ctx, r, rN, 1.0, 0, problem.count)
    return recs + S.fallback_records(problem, ctx, r, rN, 1.0, recs)


def _records_exhaustive_complete(S, problem, ctx, r, rN):
    """... and the COMPLETE records of the whole space as the driver gathers them (search._search_local: gather): finalists, nu = 1/3
    fallbacks, and every rank-deficient matrix of the space valued by the reference's own procedure, whatever its value"""
    recs = _records_exhaustive(S, problem, ctx, r, rN)
    listed = set(problem.last_degenerate[0])
    recs = [t for t in recs if t["rank"] not in listed]
    # (degenerate_records returns the NaN ones and the finite ones within the window of the smallest value known -- a space of 1e10
    # matrices holds millions of rank-deficient ones, and a Python record each took the first GPU box of round 6 down)
    return recs + S.degenerate_records(problem, ctx, r, rN, 1.0, recs=recs)


def _finite(recs):
    return [t for t in recs if t["nll"] == t["nll"]]


def _full_rank(recs):
    if not recs:
        return []
    keep = ~rank_deficient(np.array([t["c"] for t in recs]))
    return [t for t, k in zip(recs, keep) if k]


def _plain(best):
    return [(np.asarray(t["c"]).tolist(), [float(x) for x in t["mu"]], float(t["nll"])) for t in best]


SPACES = [(12, 3, 31), (13, 3, 2), (14, 3, 9), (12, 4, 5), (13, 4, 6), (11, 5, 8), (12, 5, 1), (10, 6, 3), (12, 3, 55),
          (12, 3, 41), (13, 3, 42), (12, 4, 43), (11, 5, 44), (10, 6, 45), (14, 3, 46), (13, 4, 47), (11, 5, 56), (11, 4, 49), (10, 6, 51),
          (11, 5, 52), (13, 3, 53), (12, 4, 54)]


@pytest.mark.parametrize("m,K,seed", SPACES)
def test_mixture_space_search_equals_the_exhaustive_search_on_whole_spaces(ctx, m, K, seed):
    """Whole spaces of 1e6 .. 1e9 matrices: what do_optimization_single keeps of the space -- every full-rank matrix the reference
    reports within its tie margin of the minimum, in enumeration order, C bit-exact, NLL and mu to 1e-9 -- found by branch and
    bound over the mixture space, against the replay over the linear walk's records."""
    import bench
    import theta_amd
    from theta_amd import search as S
    r, rN, order = bench.synth(seed=seed, m=m, n=3, k=K)
    lb, ub = [0] * m, [K] * m
    p = theta_amd.Problem(ctx, 3, m, 2, r, rN, lb, ub, 1.0)
    assert 1e5 < p.count < 2e10
    rep = S.SearchReport()
    recs_mix, _ = S.mix_records(p, ctx, r, rN, 1.0, (lb, ub), report=rep)
    walk = _records_exhaustive_complete(S, p, ctx, r, rN)
    want = S.replay_records(_full_rank(walk), False)
    got = S.replay_records(_full_rank(recs_mix), False)
    # round 6: the COMPLETE list -- the rank-deficient matrices too, which the reference reports wherever its hybrj stops on their
    # singular Jacobian (a finite value BELOW the matrix's minimum over mu >= 0 included): the lines' trees of theta_mix_search
    # bound them all; only NaN outcomes are beyond any bound
    want_all = S.replay_records(_finite(walk), False)
    got_all = S.replay_records(_finite(recs_mix), False)
    n_def = len(p.last_degenerate[0])
    p.close()
    assert len(want) >= 1
    assert campaign.compare_best(_plain(got), _plain(want), tol=1e-9) == "", (m, K, seed, len(got), len(want), rep.mix)
    assert campaign.compare_best(_plain(got_all), _plain(want_all), tol=1e-9) == "", (m, K, seed, len(got_all), len(want_all), rep.mix)
    assert rep.mix["minimum"] <= rep.mix["incumbent"] + 1e-9 and rep.mix["boxes_tested"] > 0
    assert rep.mix["rank_deficient_complete"] and rep.mix["lines"] > 0 and n_def > 0
    # nothing finite below the bound the report states for the rank-deficient matrices that are NOT among its records
    listed = set(np.asarray(t["c"]).tobytes() for t in recs_mix)
    rest = [t["nll"] for t in _finite(walk) if rank_deficient(np.asarray(t["c"])[None])[0] and np.asarray(t["c"]).tobytes() not in listed]
    assert not rest or min(rest) >= rep.mix["rank_deficient_bound"] - 1e-6, (min(rest), rep.mix["rank_deficient_bound"])


@pytest.mark.parametrize("m,K,seed", [(12, 3, 31), (12, 4, 5), (11, 5, 8), (13, 3, 2)])
def test_row_tree_walk_returns_the_complete_best_list_of_the_exhaustive_search(ctx, m, K, seed, monkeypatch):
    """theta_bnb + theta_search_ranges behind do_optimization_single (THETA_USE_MIX off): the COMPLETE list -- rank-deficient and NaN
    entries included, every collinear prefix followed -- identical to the linear walk's, entry by entry."""
    import bench
    from theta_amd import search as S
    r, rN, order = bench.synth(seed=seed, m=m, n=3, k=K)
    monkeypatch.setattr(S, "NAN_SWEEP_MAX", 0)
    monkeypatch.setattr(S, "USE_MIX", False)
    monkeypatch.setattr(S, "BNB_MIN_CANDIDATES", 2 ** 200)
    a = S.do_optimization_single(3, m, K, 2, [0] * m, [K] * m, r, rN, 1.0, order, False, False)
    monkeypatch.setattr(S, "BNB_MIN_CANDIDATES", 0)
    b = S.do_optimization_single(3, m, K, 2, [0] * m, [K] * m, r, rN, 1.0, order, False, False)
    rb = S.last_report
    assert rb.bnb is not None and rb.bnb["rank_deficient_complete"] and rb.bnb["ranges"] >= 1
    assert campaign.compare_best(campaign.best_to_plain(b), campaign.best_to_plain(a), tol=1e-9) == "", (m, K, seed, rb.bnb)


def test_mixture_space_search_against_lists_written_by_the_reference(ctx, monkeypatch):
    """The n=3 instances of the reference-written campaign fixtures (complete `best` lists of python/RunTHetA.py itself) through the
    mixture-space search: every finite entry of the reference's list -- rank-deficient ones included (round 6) --, in order, C bit-exact,
    NLL / mu to 1e-6 -- and nothing else.  (NaN entries are the linear walk's: DESIGN.md section 8.)"""
    from theta_amd import search as S
    import theta_amd
    monkeypatch.setattr(S, "BNB_MIN_CANDIDATES", 0)
    checked = agree = skipped = 0
    bad = []
    for name in ("best_campaign.json", "best_campaign2.json", "best_campaign3.json"):
        for c in load_json(name)["cases"][::3]:           # (every third instance: 70 of them, 15 s)
            if c["n"] != 3 or c["m"] < 5:
                continue
            ref = [(np.asarray(b["C"])[:, 1:].astype(int), [float(x) for x in b["mu"]], float(b["nll"]) if b["nll"] is not None else float("nan"))
                   for b in c["best"]]
            p = theta_amd.Problem(ctx, 3, c["m"], c["tau"], c["r"], c["rN"], c["lb"], c["ub"], 1.0)
            try:
                recs, _ = S.mix_records(p, ctx, c["r"], c["rN"], 1.0, (list(c["lb"]), list(c["ub"])))
            except theta_amd.ThetaError:
                skipped += 1                       # (a flat likelihood: the driver falls back to the walks; not this test's subject)
                p.close()
                continue
            p.close()
            got = S.replay_records(_finite(recs), False)
            # the reference's entries in SORTED interval order (C_sorted[i] = C_original[order[i]], DataTools.py:132-146), NaN ones set
            # aside (round 6: the rank-deficient entries with a finite value are the search's too)
            want = []
            for Cr, mu, nll in ref:
                Cs = Cr[np.asarray(c["order"])]
                if nll != nll:
                    continue
                want.append((Cs.tolist(), mu, nll))
            checked += 1
            why = campaign.compare_best(_plain(got), want, tol=1e-6)
            if why == "":
                agree += 1
            else:
                bad.append((name, c["seed"], why, len(ref), len(want), len(got)))
    assert checked >= 40, (checked, skipped)
    assert not bad, bad[:5]


def _config(K, seed, m=50):
    import bench
    r, rN, order = bench.synth(seed=seed, m=m, n=3, k=K)
    return r, rN, order


@pytest.mark.parametrize("name,m,K,seed", [("config 3", 50, 4, 7), ("config 4", 50, 6, 4242), ("config 5's shape", 200, 7, 55)])
def test_baseline_configs_3_and_4_are_searched_whole(ctx, name, m, K, seed, capsys):
    """BASELINE config 3 / 4 (synthetic m = 50 intervals, n = 3, k = 4 / 6, full bounds: 4e27 / 2.6e38 matrices) through
    do_optimization_single: the arg-min of the WHOLE space in about a second.  No exhaustive search can confirm it (the equality
    with the linear walk is the whole-space tests' above); what can be checked here: the winner is a matrix of the reference's
    space and theta_solve_batch reports the same NLL for it; no matrix of the space that differs from it in ONE row is better
    (every single-row change valued by the reference's procedure); and the octree's own certificate -- the smallest bound among
    the boxes of the last proposal pass, a lower bound of the minimum over everything the search covers up to the boxes' size,
    lies within the window below the reported minimum."""
    from theta_amd import search as S
    r, rN, order = _config(K, seed, m)
    t0 = time.time()
    best = S.do_optimization_single(3, m, K, 2, [0] * m, [K] * m, r, rN, 1.0, order, False, False)
    dt = time.time() - t0
    rep = S.last_report
    assert rep.mix is not None and "gave_up" not in rep.mix, rep.mix
    assert rep.candidates > 1e27            # (the time is printed below; tests/test_gpu_perf.py asserts on times, behind the `perf` marker)
    fin = [b for b in best if b[2] == b[2]]
    assert fin and abs(min(b[2] for b in fin) - rep.mix["minimum"]) < 1e-6
    with capsys.disabled():
        print("\n%s: %.3g matrices, best NLL %.6f, %d entries, %.2f s (%d boxes, %d leaves, %d matrices listed)" %
              (name, rep.candidates, fin[0][2], len(best), dt, rep.mix["boxes_tested"], rep.mix["leaves"], rep.mix["listed"]))
    low = rep.mix["minimum"]
    assert low - S.COLLECT_WINDOW - 0.5 <= rep.mix["min_bound"] <= low + 1e-6, (rep.mix["min_bound"], low)
    # round 6: the rank-deficient matrices (1e21 of them in config 4) are bounded too -- nothing finite the reference could report
    # for one of them lies below the stated bound, which is at least the threshold unless a line's box came within it
    assert rep.mix["rank_deficient_complete"] and rep.mix["lines"] >= 100
    assert rep.mix["rank_deficient_bound"] >= low - S.COLLECT_WINDOW - 0.5, rep.mix
    with capsys.disabled():
        print("    rank-deficient matrices: %d lines, %d leaves of lines, %d records, nothing finite below %.3f (threshold %.3f)" %
              (rep.mix["lines"], rep.mix["line_leaves"], rep.mix["rank_deficient_records"], rep.mix["rank_deficient_bound"], rep.mix["threshold"]))
    # the winner in sorted interval order: in the space, valued alike by the reference's procedure, no better neighbour
    Cw = np.asarray(best[0][0])[np.asarray(order)][:, 1:].astype(np.uint8)
    assert S.in_space_n3(Cw, [0] * m, [K] * m, 2)
    rows = [(a, b) for b in range(K + 1) for a in range(K + 1) if (2 - a) * (2 - b) >= 0]
    trials = [Cw]
    for i in range(m):
        for a, b in rows:
            if (a, b) != (int(Cw[i][0]), int(Cw[i][1])):
                T = Cw.copy()
                T[i] = (a, b)
                if S.in_space_n3(T, [0] * m, [K] * m, 2):
                    trials.append(T)
    ok, _mu, nll, _v = ctx.solve_batch(3, 2, r, rN, np.ascontiguousarray(np.array(trials, np.uint8)), 1.0, want_vals=False)
    assert ok[0] == 1 and abs(nll[0] - low) <= 1e-9 * abs(low)
    others = nll[1:][(ok[1:] > 0) & (nll[1:] == nll[1:])]
    assert len(trials) > 20 and (len(others) == 0 or others.min() >= low - 1e-9 * abs(low)), (len(trials), float(others.min()), low)


def test_more_than_64_rows_within_the_bounds_are_searched_whole(ctx, monkeypatch):
    """Full bounds [0, 9] on every interval: 72 valid rows (a, b), more than a child mask of the rank-walking kernels holds -- round 4
    refused such a problem (Enumerator._create_graph, Enumerator.py:272-298, has no such limit).  It has no ranks here either
    (theta_search says so), but do_optimization_single searches it WHOLE over the mixture space: the winner is a matrix of the
    reference's space valued alike by theta_solve_batch, not worse than the winner of the same instance under bounds tightened around
    the planted truth (a space the linear walk exhausts), with no better matrix one row away."""
    import bench
    import theta_amd
    from theta_amd import search as S
    m, K = 12, 9
    r, rN, order = bench.synth(seed=77, m=m, n=3, k=K)
    p = theta_amd.Problem(ctx, 3, m, 2, r, rN, [0] * m, [K] * m, 1.0)
    assert p.count == 2 ** 128 - 1
    with pytest.raises(theta_amd.ThetaError):
        p.search(0, 1000)
    p.close()
    best = S.do_optimization_single(3, m, K, 2, [0] * m, [K] * m, r, rN, 1.0, order, False, False)
    rep = S.last_report
    assert rep.mix is not None and "gave_up" not in rep.mix and rep.mix["records"] >= 1
    low = min(b[2] for b in best if b[2] == b[2])
    Cw = np.asarray(best[0][0])[np.asarray(order)][:, 1:].astype(np.uint8)
    assert S.in_space_n3(Cw, [0] * m, [K] * m, 2)
    rows = [(a, b) for b in range(K + 1) for a in range(K + 1) if (2 - a) * (2 - b) >= 0]
    assert len(rows) > 64
    trials = [Cw]
    for i in range(m):
        for a, b in rows:
            T = Cw.copy()
            T[i] = (a, b)
            if (a, b) != (int(Cw[i][0]), int(Cw[i][1])) and S.in_space_n3(T, [0] * m, [K] * m, 2):
                trials.append(T)
    ok, _mu, nll, _v = ctx.solve_batch(3, 2, r, rN, np.ascontiguousarray(np.array(trials, np.uint8)), 1.0, want_vals=False)
    assert ok[0] == 1 and abs(nll[0] - low) <= 1e-9 * abs(low)
    others = nll[1:][(ok[1:] > 0) & (nll[1:] == nll[1:])]
    assert len(others) == 0 or others.min() >= low - 1e-9 * abs(low)
    # the same instance with bounds around the winner's rows that survive Enumerator._check_bound_order (lb the suffix minimum, ub the
    # prefix maximum of the winner's entries: already monotone) -- at most 64 rows, a space the linear walk exhausts
    lo_i = [int(min(Cw[i])) for i in range(m)]
    hi_i = [int(max(Cw[i])) for i in range(m)]
    lb = [min(lo_i[i:]) for i in range(m)]
    ub = [max(hi_i[:i + 1]) for i in range(m)]
    assert S.adjusted_bounds(lb, ub) == (lb, ub) and S.in_space_n3(Cw, lb, ub, 2)
    q = theta_amd.Problem(ctx, 3, m, 2, r, rN, lb, ub, 1.0)
    count = q.count
    q.close()
    assert 1 < count < 2e11, count
    monkeypatch.setattr(S, "BNB_MIN_CANDIDATES", 2 ** 200)
    monkeypatch.setattr(S, "NAN_SWEEP_MAX", 0)
    tight = S.do_optimization_single(3, m, K, 2, list(lb), list(ub), r, rN, 1.0, order, False, False)
    fin = [b for b in tight if b[2] == b[2]]
    assert fin and abs(min(b[2] for b in fin) - low) <= 1e-9 * abs(low), (min(b[2] for b in fin), low, count)
    assert np.array_equal(np.asarray(fin[0][0])[np.asarray(order)][:, 1:].astype(np.uint8), Cw) or len(fin) > 1


def test_command_line_on_a_space_no_walk_finishes(ctx, tmp_path, capsys):
    """`RunTHetA file -n 3 -k 4 --NO_INTERVAL_SELECTION --FORCE` on BASELINE config 3 written as an .intervals file with full bounds
    [0, 4] (4e27 matrices): the time estimate (the mixture-space search itself, timed: seconds, not the 1e19 years of a walk) lets
    the run through, and the results file holds the winner do_optimization_single reports for the same instance."""
    import bench
    from theta_amd import RunTHetA
    from theta_amd import search as S
    r, rN, order = bench.synth(seed=7, m=50, n=3, k=4)
    path = tmp_path / "c3.intervals"
    with open(path, "w") as f:
        f.write("#ID\tchrm\tstart\tend\ttumorCount\tnormalCount\tupperBound\tlowerBound\n")
        for i in range(50):
            f.write("i%d\t1\t%d\t%d\t%d\t%d\t4\t0\n" % (i, 1000 * i, 1000 * i + 900, r[i], rN[i]))
    t0 = time.time()
    RunTHetA.main([str(path), "-n", "3", "-k", "4", "--NO_INTERVAL_SELECTION", "--FORCE", "-p", "c3", "-d", str(tmp_path)])
    wall = time.time() - t0
    out = capsys.readouterr().out
    assert "Estimated Total Time:" in out and "second(s)" in out.split("Estimated Total Time:")[1].splitlines()[0]
    lines = [l for l in open(tmp_path / "c3.n3.results") if not l.startswith("#")]
    nll_cli = float(lines[0].split("\t")[0])
    best = S.do_optimization_single(3, 50, 4, 2, [0] * 50, [4] * 50, r, rN, 1.0, list(range(50)), False, False)
    assert abs(nll_cli - best[0][2]) <= 1e-9 * abs(best[0][2]) and len(lines) == len(best)
    rows = [[int(v) for v in row.split(",")] for row in lines[0].split("\t")[2].split(":")]
    assert np.array_equal(np.array(rows), np.asarray(best[0][0])[:, 1:].astype(int))
    # (round 6) the run says what a whole-space search covers: every finite outcome within the window, a bound for the rank-deficient rest
    assert "Whole space by branch and bound over the mixture space" in out and "NaN outcomes are not listed" in out
    print("command line on config 3: %.2f s" % wall)          # (timing: tests/test_gpu_perf.py)


def test_get_values_on_a_space_no_walk_finishes_is_skipped_with_a_warning(ctx, tmp_path, capsys):
    """--GET_VALUES (RunTHetA.py:210-215) writes a line per candidate: on BASELINE config 3 (4e27 matrices) the search completes, the
    dump is skipped with a warning (THETA_GET_VALUES_MAX) -- the reference would evaluate and write for 1e19 years."""
    import bench
    from theta_amd import search as S
    r, rN, order = bench.synth(seed=7, m=50, n=3, k=4)
    S.pre = str(tmp_path / "big")
    try:
        best = S.do_optimization_single(3, 50, 4, 2, [0] * 50, [4] * 50, r, rN, 1.0, order, False, True)
    finally:
        pre, S.pre = S.pre, "theta"
    out = capsys.readouterr().out
    assert "WARNING: --GET_VALUES" in out and not (tmp_path / "big.likelihoods").exists()
    assert len(best) >= 1 and abs(best[0][2] - 22588904.807977) < 1e-3


def test_counting_table_of_a_space_beyond_2_to_the_128_waits_for_the_first_rank(ctx, monkeypatch):
    """BASELINE config 5's shape (m = 200, k = 7, full bounds: ~1e150 matrices): theta_problem_create knows from a host-side lower
    bound (theta_count_lower_bound, 2^530) that the count saturates and leaves the 2 GB counting table (200 launches, 0.4 s) to the
    first call that takes ranks -- the mixture-space search never asks.  A rank range searched afterwards returns what the same
    problem returns with the table built at creation (THETA_N3_LAZY_TABLE=0)."""
    import bench
    import theta_amd
    r, rN, order = bench.synth(seed=55, m=200, n=3, k=7)
    times = {}
    out = {}
    for lazy in ("1", "0"):
        monkeypatch.setenv("THETA_N3_LAZY_TABLE", lazy)
        t0 = time.time()
        p = theta_amd.Problem(ctx, 3, 200, 2, r, rN, [0] * 200, [7] * 200, 1.0)
        times[lazy] = time.time() - t0
        assert p.count == 2 ** 128 - 1
        b = 2 ** 100
        res = p.search(b, b + (1 << 22), window=0.5)
        out[lazy] = (res["rank"], res["nll"], p.enumerate(b + 12345, 3))
        p.close()
    print("problem creation, lazy / eager counting table: %.3f / %.3f s" % (times["1"], times["0"]))     # (timing: tests/test_gpu_perf.py)
    assert list(out["1"][0]) == list(out["0"][0]) and np.array_equal(out["1"][1], out["0"][1]) and np.array_equal(out["1"][2], out["0"][2])
    assert len(out["1"][0]) >= 1


@pytest.mark.parametrize("G", [2, 8])
def test_sharded_boxes_partition_the_search(ctx, G):
    """Options mix_shard_world / mix_shard_rank (round 6): the boxes a few cuts below the roots are dealt out over G ranks by their path.
    On one GPU, rank after rank: the ranks' lists together are the unsharded list -- nothing lost at the cuts --, most ranks find leaves
    of their own, the leaves add up to the unsharded search's (a leaf belongs to exactly one rank), and the boxes tested add up to the
    unsharded count plus the few hundred every rank walks alike above the cut."""
    import bench
    import theta_amd
    from theta_amd import search as S
    m, K, seed = 50, 4, 7
    r, rN, _order = bench.synth(seed=seed, m=m, n=3, k=K)
    p = theta_amd.Problem(ctx, 3, m, 2, r, rN, [0] * m, [K] * m, 1.0)
    thr = 22588904.807977 + S.COLLECT_WINDOW + 4 * S.TIE_MARGIN          # (config 3's minimum: test_baseline_configs_3_and_4_are_searched_whole)
    whole, st0 = p.mix_search(thr, leaf_rel=1.3e-4, cap=1 << 16, lines=True)
    assert len(whole) >= 2 and st0["leaves"] > 100
    union, leaves, boxes, with_leaves = set(), 0, 0, 0
    p.set_option("mix_shard_world", G)
    for g in range(G):
        p.set_option("mix_shard_rank", g)
        part, st = p.mix_search(thr, leaf_rel=1.3e-4, cap=1 << 16, lines=True)
        with_leaves += 1 if st["leaves"] > 0 else 0
        leaves += st["leaves"]
        boxes += st["boxes_tested"]
        union |= set(np.asarray(c).tobytes() for c in part)
    p.set_option("mix_shard_world", 1)
    p.close()
    assert union == set(np.asarray(c).tobytes() for c in whole)
    assert leaves == st0["leaves"] and with_leaves >= (G + 1) // 2
    assert st0["boxes_tested"] <= boxes <= st0["boxes_tested"] + G * 200000          # (what every rank walks alike above the cut)
