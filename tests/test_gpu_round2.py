"""
GPU parity tests added in round 2 (run with `-m gpu` on the MI355X box), all through the C ABI:

  * `best` COMPLETE and entry by entry -- NaN entries included -- against lists written by the reference itself on seeded
    campaign instances (tests/golden/best_campaign.json, tests/golden/make_golden_campaign.py), and against the oracle's port
    of the reference driver run live on this box's cores on another slice of the same generator;
  * matrices with an all-zero tumour column: what the reference reports for them (M3's hybrd residue) is reproduced;
  * n=2 intervals with zero tumour reads (the reference's 0/0 at nu = 0);
  * config 4 (m=50, n=3, k=6): tight-bounds instance against the oracle, and an 8-way rank partition on one GPU;
  * suspect-list overflow in a chunked search is repaired or raised, never silent;
  * (the library's communicator on the GPU box: tests/test_gpu_zz_comm.py -- last in the run, so that no test forks a
    worker pool from a process that has initialised RCCL.)
"""
import multiprocessing as mp
import os
import socket
import sys
import warnings

import numpy as np
import pytest

import campaign
import theta_oracle as orc
from conftest import ROOT, load_json, unfl

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import theta_amd
    return theta_amd.default_context()


def _gpu_best(inst):
    from theta_amd.search import do_optimization_single
    try:
        best = do_optimization_single(inst["n"], inst["m"], inst["k"], inst["tau"], list(inst["lb"]), list(inst["ub"]), inst["r"],
                                      inst["rN"], inst["mx"], inst["order"])
    except SystemExit:
        best = []
    return campaign.best_to_plain(best)


# ---------------------------------------------------------------------------------------------------
# `best` against the reference's own lists (fixtures) -- complete lists, NaN entries included
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fixture", [f for f in ("best_campaign.json", "best_campaign2.json", "best_campaign3.json", "best_campaign4.json", "best_campaign5.json", "best_nan_cases.json", "best_tau.json")
                                     if os.path.exists(os.path.join(ROOT, "tests", "golden", f))])
def test_best_identical_to_reference_lists_on_campaign_fixtures(ctx, fixture):
    """(best_campaign2.json: a second set of seeds on larger spaces -- up to 60 000 candidates for n=3, 200 000 for n=2 --
    written by the reference after the hybrj restatement was settled: tests/golden/make_golden_campaign.py second;
    best_campaign3.json: n=3 only, 80 instances of the low-coverage shape -- a few to a few hundred reads per interval, one tumour
    population in half of them -- and 30 mid ones: ... third; best_campaign4.json: two whole spaces of 4e5 / 4.7e5 matrices in which
    tools/nan_hunt.py found FULL-RANK matrices the reference reports with a NaN likelihood -- the NaN sweep's case: ... fourth;
    best_nan_cases.json: 18 small spaces built around six such matrices, tests/golden/make_golden_nan_cases.py -- 11 NaN tuples
    of full-rank matrices in the reference's lists, which only the sweep finds; best_tau.json: n=3 under --TAU 1 and 3,
    tests/golden/make_golden_tau.py; best_campaign5.json: n=3 with the bounds of the reference's own heuristic on counts with a strongly
    amplified interval -- copy numbers 8 to 10, the compact row alphabet: ... fifth)"""
    cases = load_json(fixture)["cases"]
    assert len(cases) >= {"best_campaign4.json": 2, "best_campaign5.json": 12, "best_nan_cases.json": 15, "best_tau.json": 25}.get(fixture, 40)
    bad, n_nan, n_cand = [], 0, 0
    for c in cases:
        ref = [(b["C"], [unfl(x) for x in b["mu"]], unfl(b["nll"])) for b in c["best"]]
        got = _gpu_best(c)
        why = campaign.compare_best(got, ref)
        if why:
            bad.append((c["n"], c["shape"], c["seed"], why))
        n_nan += sum(1 for b in ref if b[2] != b[2])
        n_cand += c["count"]
    assert not bad, bad
    assert n_nan >= {"best_campaign.json": 5, "best_campaign2.json": 1, "best_nan_cases.json": 8}.get(fixture, 0)      # the fixtures do exercise the isClose(NaN) entries


def _oracle_side(inst):
    warnings.simplefilter("ignore")
    best, cnt = orc.search_single(inst["n"], inst["m"], inst["tau"], list(inst["lb"]), list(inst["ub"]), inst["r"], inst["rN"],
                                  inst["mx"], inst["order"])
    return campaign.best_to_plain(best), cnt


def test_campaign_slice_against_the_oracle_driver(ctx):
    """60 seeded instances (toy + mid, n=2 and n=3; other seeds than the fixtures): the GPU driver against the oracle's port of
    the reference driver -- every candidate through scipy -- spread over this box's cores.  Complete `best` lists."""
    want = {(2, "toy"): 15, (2, "mid"): 15, (3, "toy"): 15, (3, "mid"): 15}
    limit = {2: (50, 30000), 3: (50, 6000)}
    insts = []
    for (n, shape), k in want.items():
        seed, got = 9000, 0
        while got < k:
            seed += 1
            inst = campaign.instance(seed, n, shape)
            cnt = campaign.count_candidates(inst)
            if limit[n][0] <= cnt <= limit[n][1]:
                inst["count"] = cnt
                insts.append(inst)
                got += 1
    gpu = [_gpu_best(i) for i in insts]
    procs = max(1, min(len(insts), (os.cpu_count() or 2) - 2))
    with mp.get_context("fork").Pool(procs) as pool:
        ref = pool.map(_oracle_side, insts, chunksize=1)
    bad = []
    for inst, g, (rb, cnt) in zip(insts, gpu, ref):
        assert cnt in (inst["count"], inst["count"] + 1)           # (+1: quirk Q1, the first matrix is evaluated twice / extra)
        why = campaign.compare_best(g, rb)
        if why:
            bad.append((inst["n"], inst["shape"], inst["seed"], why))
    assert not bad, bad


# ---------------------------------------------------------------------------------------------------
# all-zero tumour columns
# ---------------------------------------------------------------------------------------------------
def test_all_zero_tumour_column_outcomes_match_the_reference_table(ctx):
    import theta_amd
    g = np.load(os.path.join(ROOT, "tests", "golden", "solve_n3_m6k3.npz"))
    C = g["C"]
    z = np.nonzero((C[:, :, 0].sum(axis=1) == 0) | (C[:, :, 1].sum(axis=1) == 0))[0]
    assert len(z) == 28
    ok, mu, nll, vals = ctx.solve_batch(3, 2, g["r"], g["rN"], C[z], 1.0)
    assert ok.all()                                                 # the reference returns a tuple for every one of them
    ref_nll, ref_mu = g["nll"][z], g["mu"][z]
    assert np.array_equal(np.isnan(nll), np.isnan(ref_nll)) and np.isnan(ref_nll).sum() == 13
    fin = ~np.isnan(ref_nll)
    assert (np.abs(nll[fin] - ref_nll[fin]) <= 1e-9 * np.abs(ref_nll[fin])).all()
    # mu is a unit vector plus MINPACK's rounding residue (1e-24 .. 1e-39): the residue itself is reproduced
    assert np.allclose(mu, ref_mu, rtol=1e-6, atol=0.0)
    # ... and the search hands exactly the RANK-DEFICIENT matrices -- rows (x_i, y_i) on one line, the 28 among them -- to the
    # host (theta_search_degenerate): what the reference reports for those is not their optimum (n3_core.hpp: N3Line)
    from conftest import rank_deficient
    d = np.nonzero(rank_deficient(C))[0]
    assert set(z.tolist()) <= set(d.tolist()) and 500 < len(d) < 2000
    p = theta_amd.Problem(ctx, 3, 6, 2, g["r"].tolist(), g["rN"].tolist(), g["lb"].tolist(), g["ub"].tolist())
    for sieve in (1, 0):                                            # (m = 6 runs on the fused kernel either way; both settings)
        p.set_option("n3_sieve", sieve)
        p.search(0, p.count, window=0.5)
        ranks, Cd = p.last_degenerate
        assert ranks == [int(k) for k in d] and np.array_equal(Cd, C[d])
        # a sub-range holds its own share, in rank order
        p.search(100, 5000, window=0.5)
        assert p.last_degenerate[0] == [int(k) for k in d if 100 <= k < 5000]
    p.close()


def test_best_with_nan_entries_matches_reference_fixture(ctx):
    """tests/golden/best_synth.json, the (3, 5, 3) case: the reference's list holds 13 entries, several with a NaN likelihood."""
    from theta_amd.search import do_optimization_single
    case = [c for c in load_json("best_synth.json")["cases"] if c["n"] == 3 and c["m"] == 5 and c["k"] == 3][0]
    ref = [(b["C"], [unfl(x) for x in b["mu"]], unfl(b["nll"])) for b in case["best"]]
    assert any(b[2] != b[2] for b in ref)
    best = do_optimization_single(3, 5, 3, 2, list(case["lb"]), list(case["ub"]), case["r"], case["rN"], case["max_normal"],
                                  case["order"], False, False)
    assert campaign.compare_best(campaign.best_to_plain(best), ref) == ""


# ---------------------------------------------------------------------------------------------------
# n = 2: an interval without tumour reads
# ---------------------------------------------------------------------------------------------------
def test_n2_interval_with_zero_tumour_reads(ctx):
    """r_i = 0 with c_i = 0 makes the reference's dL_dMu 0/0 at nu = 0: brenth raises, the candidate is None
    (Optimizer.py:208-221, 117-121) -- in the fused search as well as in theta_solve_batch."""
    import theta_amd
    from theta_amd.search import do_optimization_single
    rN = [150, 200, 250, 300, 280, 260]
    for r in ([0, 200, 300, 400, 420, 500], [0, 0, 300, 400, 420, 500], [3, 200, 300, 400, 420, 500]):
        rs, rNs, order = orc.sort_r(rN, r)
        m = len(r)
        p = theta_amd.Problem(ctx, 2, m, 2, rs, rNs, [0] * m, [3] * m, 1.0)
        nll, mu, st = p.values(0, p.count)
        cols = p.enumerate(0, p.count)
        ok_b, mu_b, nll_b, _ = ctx.solve_batch(2, 2, rs, rNs, cols, 1.0)
        for k, col in enumerate(cols):
            s = orc.solve_n2(orc.col_to_matrix_n2(col.tolist(), 2), rs, rNs, 1.0)
            assert (s is None) == bool(np.isnan(nll[k])) == (not ok_b[k]), (r, col.tolist())
            if s is not None:
                assert abs(s[1] - nll[k]) <= 1e-9 * abs(s[1]) and abs(s[0][0] - mu[k, 0]) < 1e-9
        p.close()
        best = do_optimization_single(2, m, 3, 2, [0] * m, [3] * m, rs, rNs, 1.0, order)
        ref, cnt = orc.search_single(2, m, 2, [0] * m, [3] * m, rs, rNs, 1.0, order)
        assert campaign.compare_best(campaign.best_to_plain(best), campaign.best_to_plain(ref)) == ""


# ---------------------------------------------------------------------------------------------------
# config 4 (m=50, n=3, k=6)
# ---------------------------------------------------------------------------------------------------
def _config4_instance(seed=406, free=6):
    """m=50, k=6: truth a valid DFS path with copy numbers up to 6; bounds = truth +-1 on `free` intervals."""
    rng = np.random.RandomState(seed)
    m = 50
    a = np.sort(rng.randint(0, 6, m))
    b = a.copy()
    b[m - free:] += 1
    L = rng.randint(2_000_000, 20_000_000, m)
    rN = np.maximum(rng.poisson(L * 0.01), 1)
    mu = np.array([0.3, 0.45, 0.25])
    p = rN * (2 * mu[0] + a * mu[1] + b * mu[2])
    p = p / p.sum()
    r = rng.multinomial(int(rN.sum() * 1.2), p)
    rs, rNs, order = orc.sort_r([int(x) for x in rN], [int(x) for x in r])
    truth = np.stack([a, b], 1)[order]
    lb = truth.min(axis=1)
    ub = truth.max(axis=1)
    lb[m - free:] = np.maximum(lb[m - free:] - 1, 0)
    ub[m - free:] = np.minimum(ub[m - free:] + 0, 6)
    return rs, rNs, order, truth, lb.tolist(), ub.tolist()


def test_config4_tight_bounds_against_the_oracle(ctx):
    import theta_amd
    from theta_amd.search import do_optimization_single
    rs, rNs, order, truth, lb, ub = _config4_instance()
    assert max(ub) == 6
    p = theta_amd.Problem(ctx, 3, 50, 2, rs, rNs, lb, ub, 1.0)
    cnt = p.count
    assert 1000 < cnt <= 8000, cnt          # 5 857
    seq = np.array(list(orc.enumerate_n3(50, 2, lb, ub)), dtype=np.uint8)
    assert len(seq) == cnt and np.array_equal(p.enumerate(0, cnt), seq)
    p.close()
    best = do_optimization_single(3, 50, 6, 2, list(lb), list(ub), rs, rNs, 1.0, order, False, False)
    procs = max(1, (os.cpu_count() or 2) - 2)
    # the oracle's exhaustive search of the same space (every candidate through scipy), split over the cores by rank range
    chunks = np.array_split(np.arange(cnt), min(procs, 64))
    with mp.get_context("fork").Pool(min(procs, 64)) as pool:
        parts = pool.map(_oracle_solve_chunk, [(seq[c], rs, rNs) for c in chunks], chunksize=1)
    table = [t for part in parts for t in part]
    # replay of the reference's driver on the oracle's per-candidate table (RunTHetA.py:188-208), Q1 matrix first
    first = orc.solve_n3(orc.first_matrix_n3(50, 2), rs, rNs)
    seq_solns = ([(orc.first_matrix_n3(50, 2), first)] if first is not None else []) + \
                [(orc.rows_to_matrix_n3([tuple(x) for x in seq[k]], 2), table[k]) for k in range(cnt) if table[k] is not None]
    ref, lowest = [], float("inf")
    for Cm, s in seq_solns:
        L = s[1]
        if orc.is_close(L, lowest):
            ref.append((orc.reverse_sort_C(Cm, order), s[0], L))
        elif L < lowest:
            ref, lowest = [(orc.reverse_sort_C(Cm, order), s[0], L)], L
    assert campaign.compare_best(campaign.best_to_plain(best), campaign.best_to_plain([(a, b, c, None) for a, b, c in ref])) == ""
    assert np.array_equal(best[0][0][order][:, 1:], truth)


def _oracle_solve_chunk(args):
    rows, rs, rNs = args
    warnings.simplefilter("ignore")
    out = []
    for c in rows:
        s = orc.solve_n3(orc.rows_to_matrix_n3([tuple(x) for x in c], 2), rs, rNs)
        out.append(None if s is None else ([float(x) for x in s[0]], float(s[1])))
    return out


def test_eight_way_rank_partition_on_one_gpu_equals_single_shard(ctx):
    """The sharded search of SURVEY 8(e) on ONE GPU: 8 x _search_local(shard=(g, 8)) -> merge within the window -> replay_ties
    equals do_optimization_single, on n=3 instances whose finalists include nu = 1/3 fallback entries and NaN entries."""
    from theta_amd import search as S
    picked = 0
    for seed in range(9100, 9400):
        inst = campaign.instance(seed, 3, "toy" if seed % 2 else "mid")
        cnt = campaign.count_candidates(inst)
        if not (400 <= cnt <= 200000):
            continue
        single = _gpu_best(inst)
        fb_single = S.last_report.fallback_finalists + S.last_report.degenerate
        recs = []
        shared = float("inf")
        for g in range(8):
            problem, c2, rr, st = S._search_local(3, inst["m"], inst["tau"], inst["lb"], inst["ub"], inst["r"], inst["rN"], inst["mx"],
                                                  shard=(g, 8), ctx=ctx, hint_exchange=(lambda local: min(local, shared)))
            recs += rr
            problem.close()
        gmin = min([t["nll"] for t in recs if t["nll"] == t["nll"]], default=float("inf"))
        merged = [t for t in recs if not (t["nll"] > gmin + S.COLLECT_WINDOW)]
        q1 = S._q1_record(ctx, 3, inst["m"], inst["tau"], inst["r"], inst["rN"])
        best8 = S.replay_ties(merged, 3, inst["tau"], inst["order"], first_duplicate=False, q1_first=q1)
        assert campaign.compare_best(campaign.best_to_plain(best8), single) == "", (seed, cnt)
        picked += 1 if fb_single else 0
        if picked >= 6:
            break
    assert picked >= 3              # instances with fallback / degenerate finalists were among them


# ---------------------------------------------------------------------------------------------------
# suspect-list overflow
# ---------------------------------------------------------------------------------------------------
def test_chunked_search_repairs_suspect_overflow(ctx, monkeypatch):
    """Problem.search over several pieces: a piece whose device suspect list overflowed is searched again with the minimum of
    the WHOLE range as its hint; the merged lists are those of the one-call search.  (Small pieces are forced here; the first
    ranks of this space are poor, so the first pieces run with a loose threshold.)"""
    import theta_amd
    r, rN, L, Ct, mu = orc.synth_counts(10, 3, 3, 1)
    rs, rNs, order = orc.sort_r(rN, r)
    p = theta_amd.Problem(ctx, 3, 10, 2, rs, rNs, [0] * 10, [3] * 10)
    whole = p.search(0, p.count, window=0.5)
    sus_whole = set(p.last_suspects[0])
    deg_whole = list(p.last_degenerate[0])
    gmin = float(whole["nll"].min())
    monkeypatch.setattr(theta_amd.Problem, "MAX_PER_CALL", {2: 1 << 40, 3: 1 << 18})
    # make the list of every piece that ran with a looser threshold than the final minimum look overflowed: the repair path
    # (second pass with the minimum of the whole range as hint) must run for exactly those pieces, and agree
    real = theta_amd.Problem._piece
    seen = {"loose": 0, "second_pass": False}

    def flaky(self, b, e, window, cap, hint):
        res, sus, dropped, deg = real(self, b, e, window, cap, hint)
        if not seen["second_pass"] and hint > gmin + 1e-9:
            seen["loose"] += 1
            dropped = 7                                   # pretend the device list lost entries
        return res, sus, dropped, deg

    def search_twice(self, *a, **k):
        return orig_search(self, *a, **k)
    orig_search = theta_amd.Problem.search
    monkeypatch.setattr(theta_amd.Problem, "_piece", flaky)
    # (the second pass calls _piece again with hint == gmin: not loose, so it is not marked)
    parts = p.search(0, p.count, window=0.5)
    assert seen["loose"] >= 1 and p.suspect_reruns == seen["loose"]
    assert parts["rank"] == whole["rank"] and np.allclose(parts["nll"], whole["nll"], rtol=1e-12)
    assert set(p.last_suspects[0]) == sus_whole
    assert list(p.last_degenerate[0]) == deg_whole

    # a piece that still overflows with the global minimum as its hint raises -- never a silent, incomplete list
    def always(self, b, e, window, cap, hint):
        res, sus, dropped, deg = real(self, b, e, window, cap, hint)
        return res, sus, 5, deg
    monkeypatch.setattr(theta_amd.Problem, "_piece", always)
    with pytest.raises(theta_amd.ThetaError):
        p.search(0, p.count, window=0.5)
    p.close()


# ---------------------------------------------------------------------------------------------------
# the two implementations of the n=3 search: sieve + finish kernels (n3_sieve.hip) against the fused kernel (n3.hip)
# ---------------------------------------------------------------------------------------------------
def _both_paths(ctx, p, begin, end, r, rN, window=0.5, hint=None):
    out = []
    for sieve in (1, 0):
        p.set_option("n3_sieve", sieve)
        if hint is not None:
            p.hint(hint)
        res = p.search(begin, end, window=window)
        sus_rk, sus_lb, sus_C = p.last_suspects
        fb = set()
        if len(sus_rk):          # the suspects that matter: those whose nu = 1/3 fallback value comes within the window
            ok, mu, nll, _ = ctx.solve_batch(3, p.tau, r, rN, sus_C, 1.0, want_vals=False)
            gmin = res["nll"].min() if len(res["nll"]) else np.inf
            fb = set(rk for rk, o, v in zip(sus_rk, ok, nll) if o and v <= min(gmin, np.nanmin(np.where(ok, nll, np.inf))) + window)
        out.append((res, fb, list(p.last_degenerate[0])))
    p.set_option("n3_sieve", 1)
    return out


def test_sieve_and_fused_search_kernels_return_identical_lists(ctx):
    import bench
    import theta_amd
    cases = []
    r, rN, order = bench.synth()                                    # the bench's instance: m=50, k=6 (burst depth 4)
    cases.append(("bench m50 k6", 50, r, rN, [0] * 50, [6] * 50, [("mid", 1 << 24), (0, 1 << 22), ("end", 1 << 22)]))
    r4, rN4, _ = bench.synth(seed=7, m=50, n=3, k=4)                # k=4 branches less: burst depth 6
    cases.append(("m50 k4", 50, r4, rN4, [0] * 50, [4] * 50, [(0, 1 << 21), ("mid", 1 << 23)]))
    r5, rN5, _ = bench.synth(seed=8, m=49, n=3, k=5)                # odd m
    cases.append(("m49 k5", 49, r5, rN5, [0] * 49, [5] * 49, [("mid", 1 << 22)]))
    r6, rN6, _ = bench.synth(seed=9, m=14, n=3, k=3)                # exhaustible, whole space, ragged bounds
    cases.append(("m14 k3", 14, r6, rN6, [0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2], [2, 2, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3], [("all", None)]))
    r7, rN7, _ = bench.synth(seed=10, m=9, n=3, k=3)
    cases.append(("m9 k3", 9, r7, rN7, [0] * 9, [3] * 9, [("all", None)]))
    r8, rN8, _ = bench.synth(seed=12, m=20, n=3, k=7)               # the largest alphabet (K = 7: 64 slots)
    cases.append(("m20 k7", 20, r8, rN8, [0] * 20, [7] * 20, [("mid", 1 << 23), (0, 1 << 21)]))
    r9, rN9, _ = bench.synth(seed=13, m=64, n=3, k=2)               # the longest matrix the n=3 kernels hold
    cases.append(("m64 k2", 64, r9, rN9, [0] * 64, [2] * 64, [("mid", 1 << 23)]))
    ra, rNa, _ = bench.synth(seed=14, m=16, n=3, k=3)               # an interval with a handful of tumour reads: Rmin << sum r,
    ra = list(ra)                                                   # the lower bound rarely applies (lambda^2 < Rmin / 4 is needed)
    ra[0], ra[7] = 3, 11
    cases.append(("m16 k3 tiny Rmin", 16, ra, rNa, [0] * 16, [3] * 16, [("mid", 1 << 21)]))
    rb, rNb, _ = bench.synth(seed=15, m=12, n=3, k=4)               # tau = 3
    cases.append(("m12 k4 tau3", 12, rb, rNb, [0] * 12, [4] * 12, [("all", None)], 3))
    total_surv = 0
    for case in cases:
        name, m, rr, rn, lb, ub, ranges = case[:7]
        tau = case[7] if len(case) > 7 else 2
        p = theta_amd.Problem(ctx, 3, m, tau, rr, rn, lb, ub, 1.0)
        known = None
        for where, span in ranges:
            if where == "all":
                b, e = 0, p.count
            elif where == "mid":
                b, e = p.count // 3, p.count // 3 + span
            elif where == "end":
                b, e = p.count - span, p.count
            else:
                b, e = where, where + span
            # (ranges after the first one of an instance start from the minimum found so far, like the pieces of one job: the
            # end of a space on its own has a poor minimum and millions of rejected matrices below it)
            (a, fa, da), (f, ff, df) = _both_paths(ctx, p, b, e, rr, rn, hint=known)
            if len(a["nll"]):
                known = float(a["nll"].min()) if known is None else min(known, float(a["nll"].min()))
            assert a["stats"]["evaluated"] == f["stats"]["evaluated"] == e - b, (name, where)
            assert a["rank"] == f["rank"] and (len(a["rank"]) >= 1 or known is not None), (name, where, len(a["rank"]), len(f["rank"]))
            assert np.array_equal(a["C"], f["C"])
            assert np.allclose(a["nll"], f["nll"], rtol=1e-11, atol=0)
            assert fa == ff, (name, where, len(fa), len(ff))
            assert da == df
            total_surv += a["stats"]["survivors"]
            print(name, where, e - b, "survivors", a["stats"]["survivors"], "redone fused", a["stats"]["fallback_candidates"], "finalists", len(a["rank"]))
        p.close()
    assert total_surv > 0


def test_sieve_end_to_end_against_the_fused_driver(ctx):
    """do_optimization_single with the fast path on and off on campaign instances (mid shape, m >= 10: the sieve applies)."""
    import theta_amd
    checked = 0
    for seed in range(9800, 9900):
        inst = campaign.instance(seed, 3, "mid")
        cnt = campaign.count_candidates(inst)
        if not (2000 <= cnt <= 3000000):
            continue
        fast = _gpu_best(inst)
        os.environ["THETA_N3_SIEVE"] = "0"
        try:
            fused = _gpu_best(inst)
        finally:
            del os.environ["THETA_N3_SIEVE"]
        assert campaign.compare_best(fast, fused, tol=1e-9) == "", seed
        checked += 1
        if checked >= 12:
            break
    assert checked >= 8


# ---------------------------------------------------------------------------------------------------
# device-resident chain of the materialised operators; per-matrix r in the literal scorer
# ---------------------------------------------------------------------------------------------------
def test_device_resident_chain_matches_host_entry_points(ctx):
    import bench
    import theta_amd
    for n, m, k in ((3, 20, 4), (2, 30, 5)):
        r, rN, order = bench.synth(seed=21, m=m, n=n, k=k)
        p = theta_amd.Problem(ctx, n, m, 2, r, rN, [0] * m, [k] * m, 1.0)
        B = 20000
        start = p.count // 2
        d_C = ctx.device_array((B, m * (n - 1)), np.uint8)
        p.enumerate_device(start, B, d_C)
        C_host = p.enumerate(start, B)
        assert np.array_equal(d_C.download().reshape(C_host.shape), C_host)
        ok_d, mu_d, nll_d, vals_d, ms = ctx.solve_batch_device(n, 2, r, rN, d_C, B, m, 1.0, want_vals=True)
        ok, mu, nll, vals = ctx.solve_batch(n, 2, r, rN, C_host, 1.0)
        assert np.array_equal(ok_d.download().astype(bool), ok) and ms > 0
        assert np.array_equal(nll_d.download(), nll, equal_nan=True) and np.array_equal(mu_d.download(), mu, equal_nan=True)
        assert np.array_equal(vals_d.download(), vals, equal_nan=True)
        mu_in = np.where(np.isnan(mu), 1.0 / n, mu)
        d_mu = ctx.device_array((B, n), np.float64).upload(mu_in)
        nll_s, ms2 = ctx.score_masked_device(n, 2, d_C, B, m, np.asarray(rN, float), np.asarray(r, float), d_mu)
        ref, _ = ctx.score_masked(n, 2, C_host, np.asarray(rN, float), np.asarray(r, float), mu_in)
        assert np.array_equal(nll_s.download(), ref, equal_nan=True)
        p.close()


def test_literal_scorer_with_one_r_per_matrix(ctx):
    """theta_score_batch_rows: the (m+1)-row matrices of calc_all_c_* differ in their last row AND its read count."""
    from theta_amd import CalcAllC
    rng = np.random.RandomState(5)
    m, B = 9, 40
    Cs = rng.randint(0, 5, (B, m, 3)).astype(float) * rng.randint(100, 900, (1, m, 1))
    Cs[:, :, 0] = 2.0 * rng.randint(100, 900, m)
    mus = rng.dirichlet(np.ones(3), B)
    rs = rng.randint(10, 1000, (B, m)).astype(float)
    many = CalcAllC.L3_many(mus, Cs, m, rs, 3)
    for b in range(B):
        one = CalcAllC.L3(mus[b], Cs[b].copy(), m, rs[b], 3)
        assert (one[0] == many[b][0]) or (one[0] != one[0] and many[b][0] != many[b][0])
        nll_ref, _ = orc.calc_L3(mus[b], Cs[b].copy(), m, rs[b], 3)
        assert (nll_ref != nll_ref and one[0] != one[0]) or abs(nll_ref - one[0]) <= 1e-12 * abs(nll_ref)


def test_n2_whole_line_generator_equals_the_lane_stream_generator(ctx, monkeypatch):
    """theta_enumerate for n=2 at sizes that take the whole-line writer (LDS transposition, n2_enumerate_lines_kernel) against the
    one-stream-per-lane kernel (THETA_N2_ENUM_LEGACY=1) and, on a prefix, against the oracle's successor -- m multiple of 4,
    even, odd, ragged bounds, ranges that start and end in the middle of a run."""
    import bench
    import theta_amd
    for m, k, lb, ub in ((100, 5, None, None), (50, 6, None, None), (25, 5, None, None), (61, 3, None, None),
                         (30, 4, [0] * 10 + [1] * 10 + [2] * 10, [2] * 10 + [3] * 10 + [4] * 10)):
        r, rN, order = bench.synth(seed=3, m=m, n=2, k=k)
        lb = [0] * m if lb is None else lb
        ub = [k] * m if ub is None else ub
        p = theta_amd.Problem(ctx, 2, m, 2, r, rN, lb, ub, 1.0)
        for begin, cnt in ((0, min(p.count, 1 << 21)), (p.count // 3 + 5, min(p.count // 2, (1 << 20) + 77))):
            if cnt < 1 << 18:
                continue
            monkeypatch.delenv("THETA_N2_ENUM_LEGACY", raising=False)
            a = p.enumerate(begin, cnt)
            monkeypatch.setenv("THETA_N2_ENUM_LEGACY", "1")
            b = p.enumerate(begin, cnt)
            monkeypatch.delenv("THETA_N2_ENUM_LEGACY", raising=False)
            assert np.array_equal(a, b), (m, k, begin, cnt)
            if begin == 0:
                it = orc.enumerate_n2(m, 2, lb, ub)
                ref = np.array([next(it) for _ in range(3000)], dtype=np.uint8)
                assert np.array_equal(a[:3000], ref)
        p.close()


def test_sieve_path_in_a_chunked_search(ctx, monkeypatch):
    """Problem.search walks a range larger than one call in pieces (hint chained); with the sieve path each piece is sieve + finish.
    Small pieces are forced here; the result must be the one-call result."""
    import bench
    import theta_amd
    r, rN, order = bench.synth()
    p = theta_amd.Problem(ctx, 3, 50, 2, r, rN, [0] * 50, [6] * 50, 1.0)
    b = p.count // 7
    span = (1 << 24) + 12345
    whole = p.search(b, b + span, window=0.5)
    monkeypatch.setattr(theta_amd.Problem, "MAX_PER_CALL", {2: 1 << 40, 3: 1 << 22})
    parts = p.search(b, b + span, window=0.5)
    assert parts["stats"]["evaluated"] == whole["stats"]["evaluated"] == span
    assert parts["rank"] == whole["rank"] and np.allclose(parts["nll"], whole["nll"], rtol=1e-12)
    assert list(p.last_degenerate[0]) == []
    p.close()
