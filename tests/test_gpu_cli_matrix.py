"""
The command line and the --GET_VALUES dump on the device (ordinary -m gpu tests since round 3; their CPU twins -- the same
command lines and comparisons over the stand-in device -- are tests/test_host_cli_cpu.py):

  * the command line over a matrix of flags against the files the reference's own command line wrote
    (tests/golden/cli_matrix.json);
  * the --GET_VALUES dump line for line against the oracle's trace of the reference driver, quirk-Q1 lines included.
"""
import json
import os

import numpy as np
import pytest

import theta_oracle as orc
from conftest import GOLD
from test_host_cli_cpu import _compare_likelihoods, _compare_results_nan_aware, _rows

pytestmark = pytest.mark.gpu


def _matrix():
    with open(os.path.join(GOLD, "cli_matrix.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("case", sorted(_matrix().keys()))
def test_cli_flag_matrix_on_the_gpu(tmp_path, monkeypatch, case):
    from theta_amd import RunTHetA
    gold = _matrix()[case]
    monkeypatch.chdir(tmp_path)              # (like the reference, the --GET_VALUES dump is written to the working directory)
    argv = [os.path.join(GOLD, a) if a.startswith("cli" + os.sep) else a for a in gold["args"]] + ["-p", "c", "-d", str(tmp_path)]
    rc = 0
    try:
        RunTHetA.main(argv)
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


@pytest.mark.parametrize("n,m,k", [(2, 7, 3), (3, 6, 2), (3, 7, 2)])
def test_get_values_dump_line_for_line(tmp_path, n, m, k):
    """RunTHetA.py:210-215 over the whole evaluation sequence of the reference driver (oracle trace), first matrix included.
    (n=3: both spaces hold candidates with a singular Jacobian whose outcome hangs on libm's pow -- refpow.hpp; they were
    red in round 2.)"""
    import warnings
    import theta_amd.search as S
    r, rN, L, Ct, mu = orc.synth_counts(m, n, k, 100 + n)
    rs, rNs, order = orc.sort_r(rN, r)
    lb, ub = [0] * m, [k] * m
    S.pre = str(tmp_path / ("dump%d" % n))
    try:
        S.do_optimization_single(n, m, k, 2, list(lb), list(ub), rs, rNs, 1.0, order, False, True)
    finally:
        pre, S.pre = S.pre, "theta"
    lines = [l.rstrip("\n").split("\t") for l in open(pre + ".likelihoods")]
    trace = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        orc.search_single(n, m, 2, lb, ub, rs, rNs, 1.0, order, trace=trace)
    want = [("".join(str(int(v)) for v in Cm[:, 1]), float(soln[0][0]), float(soln[1])) for Cm, soln in trace if soln is not None]
    assert len(lines) == len(want)
    for (col, mu0, nll), (wcol, wmu0, wnll) in zip(lines, want):
        assert col == wcol
        assert (float(nll) != float(nll) and wnll != wnll) or abs(float(nll) - wnll) <= 1e-6 * abs(wnll)
        assert abs(float(mu0) - wmu0) < 1e-6 or (float(mu0) != float(mu0) and wmu0 != wmu0)
