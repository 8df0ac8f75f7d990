"""
Pins the CPU oracle (oracle/theta_oracle.py) to fixtures produced by the reference itself
(tests/golden/make_golden.py) and to the reference's own CalcAllC known-answer pickles.
CPU only.
"""
import hashlib
import math

import numpy as np
import pytest

import theta_oracle as orc
from conftest import load_json, unfl


def _close(a, b, rel=1e-9, abs_=1e-12):
    a, b = unfl(a), unfl(b)
    if isinstance(a, float) and math.isnan(a):
        return isinstance(b, float) and math.isnan(b)
    return a == b or abs(a - b) <= max(abs_, rel * max(abs(a), abs(b)))


def test_kat_calcallc_pickles_bit_exact():
    kat = load_json("kat_calcallc.json")
    e = kat["L2"]
    nll, vals = orc.calc_L2(e["mu"], np.array(e["C"], float), e["m"], np.array(e["r"]))
    assert nll == e["branch"]["nll"]                      # the fork's own answer, bit for bit
    assert abs(nll - e["master"]["nll"]) < 1e-11           # upstream loops differ by ~2e-12
    for v, w in zip(vals, e["branch"]["vals"]):
        assert v == w
    e = kat["L3"]
    nll, vals = orc.calc_L3(e["mu"], np.array(e["C"], float), e["m"], np.array(e["r"]), e["n"])
    assert nll == e["branch"]["nll"]
    assert abs(nll - e["master"]["nll"]) < 1e-11
    for v, w in zip(vals, e["branch"]["vals"]):
        assert v == w
    q = kat["Q10"]
    assert math.isnan(orc.calc_L3(q["mu"], np.array(q["C_nan"]), 4, np.array(q["r"]), 3)[0])
    assert orc.calc_L3(q["mu"], np.array(q["C_ok"]), 4, np.array(q["r"]), 3)[0] == unfl(q["nll_ok"])


def test_calc_L2_mutates_like_reference():
    C = np.array([[2., 1], [2, 3]])
    orc.calc_L2(0.25, C, 2, np.array([1, 2]))
    assert C.tolist() == [[0.5, 0.75], [0.5, 2.25]]      # quirk Q7
    with pytest.raises(ValueError):
        orc.calc_L2(0.5, np.ones((3, 2)), 2, np.ones(2))
    with pytest.raises(ValueError):
        orc.calc_L3([.2, .3, .5], np.ones((2, 2)), 2, np.ones(2), 3)


def _enum(case):
    n, m, tau, lb, ub = case["n"], case["m"], case["tau"], case["lb"], case["ub"]
    if n == 2:
        return [[[c] for c in col] for col in orc.enumerate_n2(m, tau, lb, ub)]
    return [[list(rw) for rw in rows] for rows in orc.enumerate_n3(m, tau, lb, ub)]


def test_enumeration_order_matches_reference():
    for case in load_json("enum_order.json")["cases"]:
        seq = _enum(case)
        assert len(seq) == case["count"]
        if "seq" in case:
            assert seq == case["seq"]
        else:
            h = hashlib.sha256()
            for c in seq:
                h.update(bytes(np.asarray(c, dtype=np.uint8).reshape(-1)))
            assert h.hexdigest() == case["sha256_u8"]
            assert seq[:50] == case["first"] and seq[-50:] == case["last"]
        if case["n"] == 2:
            assert orc.count_n2(case["m"], case["lb"], case["ub"]) == case["count"] == case.get("count_ref", case["count"])
        elif "count_ref_upper" in case:
            assert orc.count_n3_upper(case["m"], case["lb"], case["ub"], case["tau"]) == case["count_ref_upper"]
        if case["n"] == 3:
            assert orc.count_n3_exact(case["m"], case["tau"], case["lb"], case["ub"]) == case["count"]
            if "rows" in case:
                assert [list(x) for x in orc.row_graph(max(orc.check_bound_order(case["lb"], case["ub"])[1]), 2)[0]] == case["rows"]


def _check_table(n, case, limit=None):
    m, r, rN = case["m"], case["r"], case["rN"]
    gen = orc.enumerate_n2(m, 2, case["lb"], case["ub"]) if n == 2 else orc.enumerate_n3(m, 2, case["lb"], case["ub"])
    for idx, (cand, ref) in enumerate(zip(gen, case["table"])):
        if limit is not None and idx >= limit:
            break
        C = orc.col_to_matrix_n2(cand, 2) if n == 2 else orc.rows_to_matrix_n3(cand, 2)
        s = orc.solve(C, r, rN, case.get("max_normal", 1))
        assert (s is None) == (ref is None), (idx, cand)
        if s is None:
            continue
        for a, b in zip(s[0], ref[0]):
            assert _close(a, b), (idx, cand, s[0], ref[0])
        assert _close(s[1], ref[1])
        if len(ref) > 2:
            for a, b in zip(s[2], ref[2]):
                assert _close(a, b)


def test_solve_n2_tables():
    g = load_json("solve_n2.json")
    for case in g["cases"]:
        _check_table(2, case, limit=400)
    d = g["degenerate"]
    for c in d["cases"]:
        s = orc.solve_n2(orc.col_to_matrix_n2(c["col"], 2), d["r"], d["rN"], c["max_normal"])
        assert (s is None) == (c["soln"] is None), c
        if s is not None:
            assert _close(s[0][0], c["soln"][0][0]) and _close(s[1], c["soln"][1])


def test_solve_n3_tables():
    for case in load_json("solve_n3_small.json")["cases"]:
        _check_table(3, case, limit=120)
        s0 = orc.solve(orc.first_matrix_n3(case["m"], 2), case["r"], case["rN"])
        assert (s0 is None) == (case["q1_first"] is None)
        if s0 is not None:
            assert _close(s0[1], case["q1_first"][1])


def test_best_matches_reference_driver():
    for case in load_json("best_synth.json")["cases"]:
        if case["n"] == 3 and case["m"] > 5:
            continue  # kept for the GPU parity test; too slow for the CPU suite
        best, count = orc.search_single(case["n"], case["m"], 2, case["lb"], case["ub"], case["r"], case["rN"],
                                        case["max_normal"], case["order"])
        assert len(best) == len(case["best"])
        for b, ref in zip(best, case["best"]):
            assert np.array_equal(b[0], np.array(ref["C"]))
            for a, c in zip(b[1], ref["mu"]):
                assert _close(a, c)
            assert _close(b[2], ref["nll"])
            for a, c in zip(b[3], ref["vals"]):
                assert _close(a, c)


def test_oracle_driver_matches_reference_best_lists_on_campaign_fixtures():
    """tests/golden/best_campaign.json holds `best` of the REFERENCE's do_optimization_single on seeded random instances --
    complete lists, NaN entries included.  The oracle's port of the driver reproduces them (the smaller instances here; the
    -m gpu suite runs the HIP path against all of them)."""
    import warnings

    import campaign
    cases = [c for c in load_json("best_campaign.json")["cases"] if c["count"] <= (600 if c["n"] == 3 else 6000)]
    cases += [c for c in load_json("best_campaign2.json")["cases"] if c["n"] == 2 and c["count"] <= 6000]     # (the second set: larger spaces)
    cases += sorted(load_json("best_tau.json")["cases"], key=lambda c: c["count"])[:8]                        # (n=3 under tau = 1 and 3)
    cases += sorted(load_json("best_campaign5.json")["cases"], key=lambda c: c["count"])[:1]                  # (copy numbers above 7: bounds of the reference's own heuristic)
    assert len(cases) >= 31 and max(max(c["ub"]) for c in cases) >= 8
    n_nan = 0
    for c in cases:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            best, cnt = orc.search_single(c["n"], c["m"], c["tau"], list(c["lb"]), list(c["ub"]), c["r"], c["rN"], c["mx"], c["order"])
        ref = [(b["C"], [unfl(x) for x in b["mu"]], unfl(b["nll"])) for b in c["best"]]
        n_nan += sum(1 for b in ref if b[2] != b[2])
        assert campaign.compare_best(campaign.best_to_plain(best), ref, tol=1e-9) == "", (c["n"], c["shape"], c["seed"])
