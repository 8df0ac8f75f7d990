"""
The RunTHetA command line on the GPU search against the output FILES the reference's own CLI wrote
(tests/golden/cli/, produced by tests/golden/make_golden_cli.py).  Files are compared by value
(Python 2/3 float formatting differs): bounds and C bit-exact, NLL / mu / p* within 1e-6 relative.
"""
import os

import numpy as np
import pytest

from conftest import GOLD, ROOT

pytestmark = pytest.mark.gpu
CLI = os.path.join(GOLD, "cli")


def _parse_results(path):
    out = []
    with open(path) as f:
        for line in f:
            if line.startswith("#"):
                continue
            nll, mu, C, p = line.rstrip("\n").split("\t")
            out.append((float(nll), [float(x) for x in mu.split(",")], C, p.split(",")))
    return out


def _compare_results(mine, ref):
    a, b = _parse_results(mine), _parse_results(ref)
    assert len(a) == len(b)
    for (n1, m1, c1, p1), (n2, m2, c2, p2) in zip(a, b):
        assert c1 == c2                                      # the copy-number profile of ALL intervals, bit-exact
        assert abs(n1 - n2) <= 1e-6 * abs(n2)
        assert np.allclose(m1, m2, rtol=0, atol=1e-6)
        assert len(p1) == len(p2)
        for x, y in zip(p1, p2):
            assert (x == "X") == (y == "X")
            if x != "X":
                assert abs(float(x) - float(y)) <= 1e-6 * abs(float(y))


def _run(argv, tmp_path):
    from theta_amd import RunTHetA
    RunTHetA.main(argv + ["-d", str(tmp_path)])


def test_cli_synthetic_n2_matches_reference_files(tmp_path):
    _run([os.path.join(CLI, "syn14.intervals"), "-n", "2", "-k", "3", "-p", "syn14"], tmp_path)
    assert open(tmp_path / "syn14.n2.withBounds").read() == open(os.path.join(CLI, "syn14.n2.withBounds")).read()
    _compare_results(tmp_path / "syn14.n2.results", os.path.join(CLI, "syn14.n2.results"))
    assert os.path.exists(tmp_path / "syn14.RunN3.bash")


@pytest.mark.skipif(not os.path.exists(os.path.join(CLI, "Example.n2.results")), reason="Example golden not generated")
def test_cli_config1_example_matches_reference_files(tmp_path):
    """BASELINE config 1 end to end: Example.intervals -n 2 -k 3 (incl. the Q7-affected calc_all_c_2)."""
    src = os.path.join(GOLD, "cli", "Example.intervals")
    _run([src, "-n", "2", "-k", "3", "-p", "Example"], tmp_path)
    assert open(tmp_path / "Example.n2.withBounds").read() == open(os.path.join(CLI, "Example.n2.withBounds")).read()
    _compare_results(tmp_path / "Example.n2.results", os.path.join(CLI, "Example.n2.results"))


def test_cli_default_pipeline_n2_then_n3_then_model_selection(tmp_path):
    """`RunTHetA <file>` with no -n: n=2, n=3 from the n=2 bounds/results, ModelSelection (RunTHetA.py:290-295)."""
    from theta_amd import CalcAllC
    _run([os.path.join(CLI, "syn14.intervals"), "-k", "3", "-p", "s", "--FORCE"], tmp_path)
    for f in ("s.n2.withBounds", "s.n2.results", "s.n3.withBounds", "s.n3.results", "s.BEST.results"):
        assert os.path.exists(tmp_path / f), f
    res3 = _parse_results(tmp_path / "s.n3.results")
    assert len(res3) >= 1
    nll, mu, C, p = res3[0]
    assert abs(sum(mu) - 1) < 1e-9 and len(mu) == 3
    # the reported NLL is CalcAllC.L3 of the reported C over all intervals with the reported mu
    rows = [r.split(",") for r in C.split(":")]
    counts = [l.split("\t") for l in open(os.path.join(CLI, "syn14.intervals")) if not l.startswith("#")]
    tum = np.array([float(c[4]) for c in counts])
    nrm = np.array([float(c[5]) for c in counts])
    Cm = np.array([[2.0] + [(-1.0 if v == "X" else float(v)) for v in r] for r in rows])
    again = CalcAllC.L3(np.array(mu), Cm * nrm[:, None], len(rows), tum, 3)[0]
    assert abs(again - nll) <= 1e-9 * abs(nll)
    best = _parse_results(tmp_path / "s.BEST.results")
    assert best[0][0] in (res3[0][0], _parse_results(tmp_path / "s.n2.results")[0][0])


def test_cli_default_pipeline_matches_reference_files_n3(tmp_path):
    """
    The default two-stage pipeline against the files the reference's own CLI wrote for the same command
    (`RunTHetA syn14.intervals -k 3 --FORCE`; tests/golden/cli/syn14d.*, make_golden_cli.py): n=2 stage, the n=3 bounds file
    derived from it, the n=3 result of 1 369 938 candidate matrices (chosen C of all intervals bit-exact, NLL / mu / p*
    1e-6) and the model-selection output.
    """
    import theta_amd.search as S
    _run([os.path.join(CLI, "syn14.intervals"), "-k", "3", "-p", "s3", "--FORCE"], tmp_path)
    assert S.last_report.candidates == 1369938
    _compare_results(tmp_path / "s3.n2.results", os.path.join(CLI, "syn14d.n2.results"))
    for kind in ("n2", "n3"):
        mine = [l.split("\t") for l in open(tmp_path / ("s3.%s.withBounds" % kind)) if not l.startswith("#")]
        ref = [l.split("\t") for l in open(os.path.join(CLI, "syn14d.%s.withBounds" % kind)) if not l.startswith("#")]
        assert [[x.strip() for x in l] for l in mine] == [[x.strip() for x in l] for l in ref]
    _compare_results(tmp_path / "s3.n3.results", os.path.join(CLI, "syn14d.n3.results"))
    _compare_results(tmp_path / "s3.BEST.results", os.path.join(CLI, "syn14d.BEST.results"))


def test_cli_default_pipeline_on_example_with_force(tmp_path):
    """
    `RunTHetA example/Example.intervals -n 2`, then `-n 3 --RESULTS ... --FORCE` on its bounds file (what the generated
    RunN3.bash runs): the n=2 stage must reproduce the reference CLI's files; the n=3
    stage -- 16 selected intervals, 98 846 979 candidate matrices, which the reference cannot finish (stopped
    after minutes at 30-800 candidates/s) -- must complete, and its reported solution must be self-consistent: the search
    winner is an optimum the oracle's solver confirms, and the written NLL is CalcAllC.L3 of the written C and mu.
    """
    import sys
    import theta_amd.search as S
    from theta_amd import CalcAllC
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import theta_oracle as orc
    # n=2, then n=3 the way the generated ex.RunN3.bash does it: -n 3 on the n=2 bounds file with --RESULTS (with -n 3 the
    # parser's default is 20 intervals, FileIO.py:170; the one-command pipeline would select up to 100)
    _run([os.path.join(CLI, "Example.intervals"), "-n", "2", "-k", "3", "-p", "ex"], tmp_path)
    _compare_results(tmp_path / "ex.n2.results", os.path.join(CLI, "Example.n2.results"))
    _run([str(tmp_path / "ex.n2.withBounds"), "-n", "3", "-k", "3", "-p", "ex", "--FORCE", "--RESULTS", str(tmp_path / "ex.n2.results")], tmp_path)
    rep = S.last_report                                       # of the n=3 search (the last one run)
    assert rep.candidates == 98846979 and rep.stats["evaluated"] == rep.candidates
    assert rep.certificate_complete and not rep.parity_uncertain
    res3 = _parse_results(tmp_path / "ex.n3.results")
    assert len(res3) >= 1
    nll, mu, C, p = res3[0]
    assert abs(sum(mu) - 1) < 1e-9 and len(mu) == 3 and min(mu) >= 0
    rows = [r.split(",") for r in C.split(":")]
    counts = [l.split("\t") for l in open(os.path.join(CLI, "Example.intervals")) if not l.startswith("#")]
    tum = np.array([float(c[4]) for c in counts])
    nrm = np.array([float(c[5]) for c in counts])
    Cm = np.array([[2.0] + [(-1.0 if v == "X" else float(v)) for v in r] for r in rows])
    again = CalcAllC.L3(np.array(mu), Cm * nrm[:, None], len(rows), tum, 3)[0]
    assert abs(again - nll) <= 1e-9 * abs(nll)


def test_calc_all_c_variants_match_reference_vectors():
    """calc_all_c_2 / _3 / _3_multi_event (CalcAllC.py:92-328) on the reference's own inputs and outputs."""
    import json
    import math
    from theta_amd import CalcAllC
    cases = json.load(open(os.path.join(GOLD, "calc_all_c.json")))["cases"]
    seen = set()
    for c in cases:
        n = c["n"]
        mu = tuple(c["mu"]) if n == 2 else np.array(c["mu"])
        for name, ref in c["out"].items():
            fn = getattr(CalcAllC, name)
            best = [(np.array(c["c"]), mu, 0.0, [])]
            (c_all, mu_o, nll, vals), = fn(best, list(c["r"]), list(c["rN"]), list(c["all_tumor"]), list(c["all_normal"]),
                                          list(c["used"]))[0]
            seen.add(name)
            assert np.array_equal(np.asarray(c_all), np.array(ref["c_all"])), name      # chosen copy numbers: bit-exact
            if ref["nll"] == "nan":
                assert math.isnan(nll)                                                  # zero-normal interval, quirk Q10
            else:
                assert abs(nll - ref["nll"]) <= 1e-9 * abs(ref["nll"])
                for a, b in zip(vals, ref["vals"]):
                    assert (a == "X") == (b == "X")
                    if a != "X":
                        assert abs(a - b) <= 1e-9 * abs(b)
    assert seen == {"calc_all_c_2", "calc_all_c_3", "calc_all_c_3_multi_event"}
