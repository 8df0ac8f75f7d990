"""
refpow::square (theta_amd/csrc/refpow.hpp) -- glibc's pow(x, 2.0) restated, the square the reference's Jacobian takes
(Optimizer.py:308, `**2` on a numpy float64) -- against the libm of this machine, bit for bit, host build of the very header
the device compiles (tools/hybrj_check.cpp).  The oracle calls the same libm through numpy.
"""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


def _both(x):
    import hybrj_check as hc
    x = np.ascontiguousarray(x, np.float64)
    a, b = np.zeros_like(x), np.zeros_like(x)
    hc.lib.refpow_check_square(len(x), x.ctypes.data_as(hc.dp), a.ctypes.data_as(hc.dp))
    hc.lib.refpow_check_libm(len(x), x.ctypes.data_as(hc.dp), b.ctypes.data_as(hc.dp))
    return a, b


def test_numpy_scalar_square_is_libm_pow_not_a_product():
    """the premise: np.float64 ** 2 differs from x*x in the last bit now and then, and equals libm's pow"""
    rng = np.random.RandomState(0)
    x = rng.rand(200000)
    a, b = _both(x)
    ns = np.array([np.float64(v) ** 2 for v in x[:50000]])
    assert np.array_equal(ns, b[:50000])
    if np.array_equal(b, x * x):
        pytest.skip("this libm's pow(x, 2) is correctly rounded: nothing to restate on this machine")
    assert 0 < (b != x * x).sum() < 1000


def test_restated_square_is_libm_pow_bit_for_bit():
    rng = np.random.RandomState(5)
    for x in (rng.rand(1000000), 10.0 ** rng.uniform(-30, 30, 500000), -rng.rand(50000), 1 + rng.uniform(-1e-3, 1e-3, 300000),
              np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, 5e-324, 1e-310, 1e-160, 1e160, 2.0, 0.5, 1 / 3.0])):
        a, b = _both(x)
        with np.errstate(all="ignore"):
            if len(x) > 100 and np.array_equal(b, np.asarray(x) * np.asarray(x)):
                pytest.skip("this libm's pow(x, 2) is correctly rounded")
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64))
    a, b = _both(np.array([np.nan]))
    assert a[0] != a[0] and b[0] != b[0]


def test_tables_regenerate_from_the_published_formulas():
    """refpow_tables.inc is what tools/gen_refpow_tables.py writes (mpmath, glibc's formulas), not an extract of a binary"""
    import subprocess
    inc = os.path.join(ROOT, "theta_amd", "csrc", "refpow_tables.inc")
    before = open(inc).read()
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_refpow_tables.py")], stdout=subprocess.DEVNULL)
    assert open(inc).read() == before
