"""
Randomised differential test: the GPU driver (`do_optimization_single`, through the C ABI) against the CPU
oracle's port of the reference driver on many small seeded instances with ragged bounds, tau != 2 and
max_normal < 1.  n=2: the complete `best` list must agree (chosen C bit-exact, mu, NLL, p*).  n=3: the
reference's tie list must be a sub-sequence of the GPU's and the first common entry must agree.
"""
import numpy as np
import pytest

import theta_oracle as orc

pytestmark = pytest.mark.gpu


def _instance(rng, n, m, k, tau):
    L = rng.randint(2_000_000, 20_000_000, m)
    rN = np.maximum(rng.poisson(L * 0.004), 5)
    C = np.full((m, n), float(tau))
    for j in range(1, n):
        C[:, j] = rng.randint(0, k + 1, m)
    mu = rng.dirichlet(np.ones(n) * 3)
    p = (C * rN[:, None]) @ mu
    p = p / p.sum()
    r = rng.multinomial(int(rN.sum() * rng.uniform(0.8, 1.5)), p)
    r = np.maximum(r, 1)
    rs, rNs, order = orc.sort_r([int(x) for x in rN], [int(x) for x in r])
    lb = [int(x) for x in rng.randint(0, 2, m)]
    ub = [int(x) for x in rng.randint(max(1, k - 1), k + 1, m)]
    return rs, rNs, order, lb, ub


def test_n2_best_lists_agree_on_random_instances():
    from theta_amd.search import do_optimization_single
    rng = np.random.RandomState(20240928)
    checked = 0
    for case in range(24):
        m = int(rng.randint(3, 10))
        k = int(rng.randint(2, 6))
        tau = int(rng.choice([1, 2, 2, 2, 3]))
        mx = float(rng.choice([1.0, 1.0, 0.5, 0.7]))
        rs, rNs, order, lb, ub = _instance(rng, 2, m, k, tau)
        ref, cnt = orc.search_single(2, m, tau, list(lb), list(ub), rs, rNs, mx, order)
        ref = [b for b in ref if b[2] == b[2]]
        if not ref:
            with pytest.raises(SystemExit):
                best = do_optimization_single(2, m, k, tau, list(lb), list(ub), rs, rNs, mx, order)
                if best == []:
                    raise SystemExit(1)           # (the caller of the reference exits on an empty list, RunTHetA.py:448-450)
            continue
        best = do_optimization_single(2, m, k, tau, list(lb), list(ub), rs, rNs, mx, order)
        assert len(best) == len(ref), (case, m, k, tau, mx)
        for b, rb in zip(best, ref):
            assert np.array_equal(b[0], rb[0])
            assert abs(b[1][0] - rb[1][0]) < 1e-9 and abs(b[2] - rb[2]) <= 1e-11 * abs(rb[2])
            assert np.allclose(b[3], rb[3], rtol=1e-9, atol=0)
        checked += 1
    assert checked >= 18


def test_n3_winners_agree_on_random_instances():
    from theta_amd.search import do_optimization_single
    rng = np.random.RandomState(777)
    for case in range(8):
        m = int(rng.randint(4, 6))
        k = int(rng.randint(2, 4))
        rs, rNs, order, lb, ub = _instance(rng, 3, m, k, 2)
        ref, cnt = orc.search_single(3, m, 2, list(lb), list(ub), rs, rNs, 1.0, order)
        ref = [b for b in ref if b[2] == b[2]]
        best = do_optimization_single(3, m, k, 2, list(lb), list(ub), rs, rNs, 1.0, order)
        assert best and ref
        # same optimum value; every reference entry appears in the GPU list, in order
        assert abs(best[0][2] - ref[0][2]) <= 1e-6 * abs(ref[0][2]) + 1e-3
        it = iter(best)
        for rb in ref:
            assert any(np.array_equal(b[0], rb[0]) for b in it), (case, "reference tie entry missing from the GPU list")


def test_n3_packed_f32_pass_and_fp64_iterations_return_the_same_finalists(monkeypatch):
    """
    The n=3 search evaluates in packed single precision; with THETA_N3_FORCE_F64=1 every evaluation
    is FP64 instead (round 3: the sieve kernel's double instantiation; m < 8: the fused kernel) -- here together with
    THETA_N3_NO_DISMISS=1, the full-solve mode.  Finalists (ranks, C, exact NLL) and the accept statistics must not depend on
    that choice.
    """
    import theta_amd
    ctx = theta_amd.Context(0)
    rng = np.random.RandomState(4242)
    for case, (m, k) in enumerate([(12, 3), (16, 2), (9, 4)]):
        rs, rNs, order, lb, ub = _instance(rng, 3, m, k, 2)
        if case:
            lb, ub = [0] * m, [k] * m        # full bounds
        out = []
        for force in ("0", "1"):
            monkeypatch.setenv("THETA_N3_FORCE_F64", force)
            monkeypatch.setenv("THETA_N3_NO_DISMISS", force)
            p = theta_amd.Problem(ctx, 3, m, 2, rs, rNs, lb, ub, 1.0)     # the switches are read when the problem is created
            total = p.count
            res = p.search(0, min(total, 1 << 22), window=0.5)
            out.append((res, p.last_suspects))
        (a, sa), (b, sb) = out
        assert a["stats"]["evaluated"] == b["stats"]["evaluated"] > 1000
        assert a["stats"]["flops_f32"] > 0 and a["stats"]["flops"] < b["stats"]["flops"]
        assert a["rank"] == b["rank"] and len(a["rank"]) >= 1, case
        assert np.array_equal(a["C"], b["C"])
        assert np.allclose(a["nll"], b["nll"], rtol=1e-12, atol=0)
        assert np.allclose(a["mu"], b["mu"], rtol=0, atol=1e-9)
        # statistics: with FP64 iterations every candidate is solved; the packed pass finishes most of them by the lower
        # bound of their optimum ("dismissed") and counts admissible optima among the others only
        assert b["stats"]["dismissed"] == 0 and a["stats"]["dismissed"] > 0
        # (`accepted` counts admissible optima among the candidates the finish kernel / cold path saw -- a subset in both modes, but
        # never fewer than the finalists, all of which went through it; and every candidate had at least one evaluation -- except the
        # candidates of prefixes the search finished by the prefix bound, phase_cycles[1]: none of them in the full-solve mode)
        for st, res in ((a["stats"], a), (b["stats"], b)):
            assert len(res["rank"]) <= st["accepted"] <= st["evaluated"] <= st["iterations"] + st["pruned"], (case, st["accepted"], st["iterations"])
        assert b["stats"]["pruned"] == 0
        assert b["stats"]["iterations"] >= a["stats"]["iterations"]            # nothing dismissed: nobody stops after the shared evaluation
        # suspects (rejected matrices whose LOWER BOUND is within the window): the two arithmetics bound borderline cases
        # differently, but what decides `best` -- the ones whose nu = 1/3 fallback value is within the window -- must agree
        def relevant(sus, res):
            if not len(sus[0]):
                return set()
            ok, mu_s, nll_s, _ = ctx.solve_batch(3, 2, rs, rNs, sus[2], 1.0, want_vals=False)
            gmin = min(float(res["nll"].min()), float(np.nanmin(np.where(ok, nll_s, np.inf))))
            return set(rk for rk, o, v in zip(sus[0], ok, nll_s) if o and v <= gmin + 0.5)
        assert relevant(sa, a) == relevant(sb, b)


def test_bench_shape_packed_pass_dismissal_and_probe_do_not_change_the_finalists(monkeypatch):
    """
    m=50, n=3, k=6 (the bench's shape): a 2^26-candidate range searched (a) as shipped -- probe, packed single-precision
    pass, dismissal by the lower bound -- and (b) with FP64 iterations and no dismissal, piecewise, must return the
    same finalists; and the probe's hint must be attained inside the range.
    """
    import bench
    import theta_amd
    ctx = theta_amd.Context(0)
    r, rN, order = bench.synth()
    m, k = 50, 6
    span = 1 << 26
    out = []
    for force in ("0", "1"):
        monkeypatch.setenv("THETA_N3_FORCE_F64", force)
        monkeypatch.setenv("THETA_N3_NO_DISMISS", force)           # (b): the bench's headline leg, full_solve_f64
        p = theta_amd.Problem(ctx, 3, m, 2, r, rN, [0] * m, [k] * m, 1.0)
        b = p.count // 5
        if force == "1":
            monkeypatch.setattr(theta_amd.Problem, "PROBE_MIN_RANGE", 1 << 62)      # (b) without the probe
        res = p.search(b, b + span, window=0.5)
        out.append(res)
    a, f = out
    assert a["stats"]["dismissed"] > 0.9 * span and f["stats"]["dismissed"] == 0
    assert a["rank"] == f["rank"] and len(a["rank"]) >= 1
    assert np.array_equal(a["C"], f["C"])
    assert np.allclose(a["nll"], f["nll"], rtol=1e-12, atol=0)
    # (c) the packed pass with the dismissal alone switched off (bench.py's `full_solve` leg)
    monkeypatch.setenv("THETA_N3_FORCE_F64", "0")
    monkeypatch.setenv("THETA_N3_NO_DISMISS", "1")
    p = theta_amd.Problem(ctx, 3, m, 2, r, rN, [0] * m, [k] * m, 1.0)
    g = p.search(b, b + span, window=0.5)
    assert g["stats"]["dismissed"] == 0 and g["stats"]["evaluated"] == span
    assert g["rank"] == a["rank"] and np.array_equal(g["C"], a["C"])
    assert np.allclose(g["nll"], a["nll"], rtol=1e-12, atol=0)


def test_every_leaf_level_setting_returns_the_same_finalists(monkeypatch):
    """The number of lane-enumerated rows (template parameter L of the n=3 search kernel; THETA_N3_LEAF_LEVELS) is a
    tuning knob: every instantiation, incl. the ones with the prefix shorter than a wave task, must agree."""
    import bench
    import theta_amd
    ctx = theta_amd.Context(0)
    r, rN, order = bench.synth(seed=5, m=12, n=3, k=3)
    ref = None
    for L in (8, 6, 5, 3, 1):
        monkeypatch.setenv("THETA_N3_LEAF_LEVELS", str(L))
        p = theta_amd.Problem(ctx, 3, 12, 2, r, rN, [0] * 12, [3] * 12, 1.0)
        res = p.search(0, p.count, window=0.5)
        assert res["stats"]["evaluated"] == p.count
        key = (res["rank"], np.round(res["nll"], 6).tolist(), res["C"].tolist())
        if ref is None:
            ref = key
        assert key == ref, L
