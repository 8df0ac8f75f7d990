"""
The command line without a GPU: theta_amd.RunTHetA.main over the stand-in device of tests/standin_device.py (every candidate
through the CPU oracle) against the FILES the reference's own command line wrote (tests/golden/cli/, make_golden_cli.py):
the n=2 run of the 14-interval synthetic file, and the two-stage pipeline n=2 -> n=3 -> model selection with
-k 3 --NUM_INTERVALS 9 (n=3 stage: 7 selected intervals, 3 576 matrices).  Bounds files byte for byte, result files by value
(C of all intervals exact, NLL / mu / p* to 1e-6 -- the same comparison the GPU tests use).
"""
import os

import pytest

import standin_device as sd
from conftest import GOLD
from test_gpu_cli import _compare_results
from theta_amd import _lib

CLI = os.path.join(GOLD, "cli")


@pytest.fixture
def standin(monkeypatch):
    ctx = sd.StandinContext()
    made = []
    cache = {}

    def make(c, n, m, tau, r, rN, lb, ub, max_normal=1.0):
        key = (n, m, tau, tuple(int(x) for x in r), tuple(int(x) for x in rN), tuple(int(x) for x in lb), tuple(int(x) for x in ub),
               float(max_normal))
        p = sd.StandinProblem(c, n, m, tau, r, rN, lb, ub, max_normal)
        p._table = cache.setdefault(key, {})             # the time estimate and the search proper solve the same candidates
        made.append(p)
        return p
    monkeypatch.setattr(_lib, "Problem", make)
    monkeypatch.setattr(_lib, "default_context", lambda: ctx)
    return made


def _run(argv, tmp_path):
    from theta_amd import RunTHetA
    RunTHetA.main(argv + ["-d", str(tmp_path)])


def test_cli_n2_over_the_standin_device_writes_the_reference_files(standin, tmp_path):
    _run([os.path.join(CLI, "syn14.intervals"), "-n", "2", "-k", "3", "-p", "syn14"], tmp_path)
    assert open(tmp_path / "syn14.n2.withBounds").read() == open(os.path.join(CLI, "syn14.n2.withBounds")).read()
    _compare_results(tmp_path / "syn14.n2.results", os.path.join(CLI, "syn14.n2.results"))
    assert os.path.exists(tmp_path / "syn14.RunN3.bash")
    assert [(p.n, p.m, p.count) for p in standin] == [(2, 14, 315)] * 2          # the time estimate, then the search


def test_cli_two_stage_pipeline_over_the_standin_device_writes_the_reference_files(standin, tmp_path):
    _run([os.path.join(CLI, "syn14.intervals"), "-k", "3", "--NUM_INTERVALS", "9", "-p", "s", "--FORCE"], tmp_path)
    for kind in ("n2", "n3"):
        mine = [l.split("\t") for l in open(tmp_path / ("s.%s.withBounds" % kind)) if not l.startswith("#")]
        ref = [l.split("\t") for l in open(os.path.join(CLI, "syn14s.%s.withBounds" % kind)) if not l.startswith("#")]
        assert [[x.strip() for x in l] for l in mine] == [[x.strip() for x in l] for l in ref]
        _compare_results(tmp_path / ("s.%s.results" % kind), os.path.join(CLI, "syn14s.%s.results" % kind))
    _compare_results(tmp_path / "s.BEST.results", os.path.join(CLI, "syn14s.BEST.results"))
    assert [(p.n, p.m, p.count) for p in standin] == [(2, 9, 105)] * 2 + [(3, 7, 3576)] * 2


# ---------------------------------------------------------------------------------------------------
# a matrix of flags: every case of tests/golden/cli_matrix.json (written by the reference's own command line,
# tests/golden/make_golden_cli_matrix.py) replayed over the stand-in device
# ---------------------------------------------------------------------------------------------------
def _matrix():
    import json
    with open(os.path.join(GOLD, "cli_matrix.json")) as f:
        return json.load(f)


def _rows(text):
    return [[x.strip() for x in l.split("\t")] for l in text.splitlines() if l and not l.startswith("#")]


def _same(x, y, rel=1e-6):
    return (x != x and y != y) or abs(x - y) <= rel * abs(y) or abs(x - y) < 1e-12


def _compare_results_nan_aware(mine, ref):
    """test_gpu_cli._compare_results, with a NaN likelihood / p* equal to a NaN (the reference writes 'nan' when an interval
    outside the search has no reads: log(0) * 0 in CalcAllC.L2/L3)."""
    from test_gpu_cli import _parse_results
    a, b = _parse_results(mine), _parse_results(ref)
    assert len(a) == len(b)
    for (n1, m1, c1, p1), (n2, m2, c2, p2) in zip(a, b):
        assert c1 == c2
        assert _same(n1, n2) and len(m1) == len(m2) and all(_same(x, y, 0) or abs(x - y) < 1e-6 for x, y in zip(m1, m2))
        assert len(p1) == len(p2)
        for x, y in zip(p1, p2):
            assert (x == "X") == (y == "X")
            if x != "X":
                assert _same(float(x), float(y))


def _compare_likelihoods(mine, ref):
    a, b = _rows(mine), _rows(ref)
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert x[0] == y[0]                                                  # the copy-number digits, in enumeration order
        for u, v in zip(x[1:], y[1:]):
            fu, fv = float(u), float(v)
            assert (fu != fu and fv != fv) or abs(fu - fv) <= 1e-6 * max(abs(fv), 1e-300) or abs(fu - fv) < 1e-9, (x, y)


@pytest.mark.parametrize("case", sorted(_matrix().keys()))
def test_cli_flag_matrix_over_the_standin_device(standin, tmp_path, monkeypatch, case):
    gold = _matrix()[case]
    monkeypatch.chdir(tmp_path)              # (like the reference, the --GET_VALUES dump is written to the working directory)
    argv = [os.path.join(GOLD, a) if a.startswith("cli" + os.sep) else a for a in gold["args"]] + ["-p", "c"]
    rc = 0
    try:
        _run(argv, tmp_path)
    except SystemExit as e:
        rc = e.code if isinstance(e.code, int) else (0 if e.code is None else 1)
    assert rc == gold["rc"]
    written = sorted(f[2:] for f in os.listdir(tmp_path)
                     if f.startswith("c.") and (f.endswith(".results") or f.endswith(".withBounds") or f.endswith(".likelihoods")))
    assert written == sorted(gold["files"].keys())
    for suffix, text in gold["files"].items():
        mine = open(tmp_path / ("c." + suffix)).read()
        if suffix.endswith(".withBounds"):
            assert _rows(mine) == _rows(text)
        elif suffix.endswith(".results"):
            ref_path = tmp_path / ("ref." + suffix)
            ref_path.write_text(text)
            _compare_results_nan_aware(tmp_path / ("c." + suffix), ref_path)
        else:
            _compare_likelihoods(mine, text)


def test_calc_all_c_variants_over_the_standin_device(standin):
    """calc_all_c_2 / _3 / _3_multi_event (CalcAllC.py:92-328) on the reference's own inputs and outputs
    (tests/golden/calc_all_c.json): the GPU test's check, with the literal scorer of the stand-in device."""
    from test_gpu_cli import test_calc_all_c_variants_match_reference_vectors as check
    check()
