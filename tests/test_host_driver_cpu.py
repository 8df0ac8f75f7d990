"""
The host side of the search without a GPU: theta_amd.search.do_optimization_single (Problem.search's walk and merge, the
reference-order re-solve of the finalists, the nu = 1/3 fallback and all-zero-column records, the tie replay) over the
stand-in device of tests/standin_device.py, against the oracle's port of the reference driver on the same instances --
complete `best` lists, NaN entries included.  Also with every search cut into many short pieces.
"""
import numpy as np
import pytest

import campaign
import standin_device as sd
import theta_oracle as orc
from theta_amd import _lib, search as S


@pytest.fixture
def standin(monkeypatch):
    ctx = sd.StandinContext()
    made = []

    def make(c, *a, **k):
        p = sd.StandinProblem(c, *a, **k)
        made.append(p)
        return p
    monkeypatch.setattr(_lib, "Problem", make)
    monkeypatch.setattr(_lib, "default_context", lambda: ctx)
    return ctx, made


def _instances(n, want, lo, hi, seeds):
    out = []
    for seed in seeds:
        inst = campaign.instance(seed, n, "toy")
        cnt = campaign.count_candidates(inst)
        if lo <= cnt <= hi:
            out.append(inst)
            if len(out) == want:
                break
    assert len(out) == want
    return out


def _reference(inst):
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        best, cnt = orc.search_single(inst["n"], inst["m"], inst["tau"], list(inst["lb"]), list(inst["ub"]), inst["r"], inst["rN"],
                                      inst["mx"], inst["order"])
    return campaign.best_to_plain(best), cnt


def _driver(inst):
    try:
        best = S.do_optimization_single(inst["n"], inst["m"], inst["k"], inst["tau"], list(inst["lb"]), list(inst["ub"]), inst["r"],
                                        inst["rN"], inst["mx"], inst["order"])
    except SystemExit:
        best = []
    return campaign.best_to_plain(best)


@pytest.mark.parametrize("n,pieces", [(2, False), (2, True), (3, False), (3, True)])
def test_driver_over_the_standin_device_equals_the_reference_driver(standin, monkeypatch, n, pieces):
    ctx, made = standin
    if pieces:
        monkeypatch.setattr(sd.StandinProblem, "MAX_PER_CALL", {2: 37, 3: 29})      # every search becomes a walk over pieces
    seen_fallback = seen_degenerate = 0
    insts = _instances(n, 5, 60, 700, range(7000 + 1000 * n, 9000 + 1000 * n))
    if n == 3:      # plus instances whose finalists include nu = 1/3 fallback records, all-zero-column records and NaN entries
        insts = insts[:2] + [campaign.instance(seed, 3, "toy") for seed in (10010, 10044, 10103)]
    for inst in insts:
        ref, cnt = _reference(inst)
        got = _driver(inst)
        if not ref and not got:
            continue
        assert campaign.compare_best(got, ref) == "", (n, inst["seed"])
        p = made[-1]
        assert p.count in (cnt, cnt - 1)                                            # (quirk Q1: the reference's extra first matrix)
        if pieces:
            assert len(p.search_calls) >= p.count // p.MAX_PER_CALL[n]
            hints = [h for _, _, h in p.search_calls]
            assert all(b <= a for a, b in zip(hints, hints[1:]))                    # later pieces start from the minimum so far
        rep = S.last_report
        seen_fallback += rep.fallback_finalists
        seen_degenerate += rep.degenerate
    if n == 3:
        assert seen_fallback + seen_degenerate > 0          # the instances exercised the records that come from the side lists


def test_a_space_beyond_reach_exits_with_a_message(standin, monkeypatch, capsys):
    ctx, made = standin

    class Huge(sd.StandinProblem):
        def __init__(self, c, n, m, tau, r, rN, lb, ub, max_normal=1.0):
            sd.StandinProblem.__init__(self, c, n, 4, tau, r[:4], rN[:4], lb[:4], ub[:4], max_normal)
            self.m = m
            self.count = 25344449490209970329508701131116975                       # n=3, m=70, bounds [0, 2]
    monkeypatch.setattr(_lib, "Problem", Huge)
    rng = np.random.RandomState(4)
    r, rN = rng.randint(1000, 5000, 70).tolist(), rng.randint(1000, 5000, 70).tolist()
    with pytest.raises(SystemExit):
        S.do_optimization_single(3, 70, 2, 2, [0] * 70, [2] * 70, r, rN, 1.0, list(range(70)))
    assert "ERROR" in capsys.readouterr().out


@pytest.mark.parametrize("n,m,k", [(2, 7, 3), (3, 5, 2)])
def test_get_values_dump_line_for_line_over_the_standin_device(standin, tmp_path, n, m, k):
    """The --GET_VALUES dump against the oracle's trace of the reference driver, first-matrix lines included: the check the
    GPU test (tests/test_gpu_cli_matrix.py) makes, here over the stand-in device."""
    from test_gpu_cli_matrix import test_get_values_dump_line_for_line as check
    check(tmp_path, n, m, k)


def test_enumerator_mirror_exposes_the_row_graph(standin):
    """Enumerator.get_graph() (Enumerator.py:166-170): the reference's time estimate walks it (TimeEstimate.py:124)."""
    from theta_amd.Enumerator import Enumerator
    for K, tau in ((2, 2), (3, 2), (4, 2), (3, 3)):
        e = Enumerator(3, 4, K, tau, [0] * 4, [K] * 4)
        rows, edges = e.get_graph()
        want_rows, want_edges = orc.row_graph(K, tau)
        assert [list(r) for r in rows] == [list(r) for r in want_rows]
        assert [list(x) for x in edges] == [list(x) for x in want_edges]
    with pytest.raises(AttributeError):
        Enumerator(2, 4, 3, 2, [0] * 4, [3] * 4).get_graph()          # (the reference builds the graph for n=3 only)
