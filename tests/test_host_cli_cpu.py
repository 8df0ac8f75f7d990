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
