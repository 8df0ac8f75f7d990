"""
The scorers' table-driven logarithm (theta_amd/csrc/smx_log.hpp: 128-entry table + degree-6 log1p, the code
score_masked_mfma_kernel and score_plain_kernel run per interval and per (candidate, mask) pair) evaluated on the CPU against
60-digit arithmetic: absolute error below 4e-16 + 2.3e-16 |ln x| (one ulp of the result) over the whole range of positive normal
doubles -- 5e-15 at most where C.mu and its sums live --, and the
library's result for everything else (zero, subnormals, negatives, inf, NaN).
"""
import ctypes as C
import os
import subprocess

import mpmath
import numpy as np
import pytest

from conftest import ROOT

LIB = os.path.join(ROOT, "build_ab", "libsmx_log_check.so")


@pytest.fixture(scope="module")
def smx():
    src = os.path.join(ROOT, "tools", "smx_log_check.hip")
    deps = [src] + [os.path.join(ROOT, "theta_amd", "csrc", f) for f in ("smx_log.hpp", "smx_log_table.inc")]
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(f) for f in deps):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-ffp-contract=off", src,
                        "-o", LIB], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    lib = C.CDLL(LIB)
    lib.smx_log_eval.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]

    def f(x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.zeros_like(x)
        lib.smx_log_eval(x.ctypes.data_as(C.POINTER(C.c_double)), y.ctypes.data_as(C.POINTER(C.c_double)), len(x))
        return y
    return f


def test_table_matches_its_generator():
    import struct
    text = open(os.path.join(ROOT, "theta_amd", "csrc", "smx_log_table.inc")).read()
    words = [int(t.strip().rstrip("ULL"), 16) for line in text.splitlines() if not line.startswith("//")
             for t in line.split("//")[0].split(",") if t.strip()]
    assert len(words) == 256
    mpmath.mp.dps = 60
    for j in range(128):
        inv_c = struct.unpack("<d", struct.pack("<Q", words[2 * j]))[0]
        log_c = struct.unpack("<d", struct.pack("<Q", words[2 * j + 1]))[0]
        assert inv_c == float(mpmath.mpf(1) / (1 + (mpmath.mpf(j) + mpmath.mpf("0.5")) / 128))
        assert log_c == float(-mpmath.log(mpmath.mpf(inv_c)))


def test_absolute_error_over_the_normal_range(smx):
    mpmath.mp.dps = 40
    rng = np.random.RandomState(12)
    xs = np.concatenate([
        np.exp(rng.uniform(np.log(2.3e-308), np.log(1.7e308), 20000)),        # the whole exponent range
        rng.uniform(1000.0, 5e6, 20000),                                        # where C.mu and its sums live
        1.0 + rng.uniform(-0.3, 0.4, 5000), 2.0 ** rng.randint(-1000, 1000, 500).astype(float),
        np.array([1.0, 2.0, 0.5, np.nextafter(1.0, 2.0), np.nextafter(1.0, 0.0), 2.2250738585072014e-308, 1.7976931348623157e308]),
        (1.0 + (np.arange(129) / 128.0)) * (1 - 1e-16), (1.0 + (np.arange(129) / 128.0)) * (1 + 1e-15),   # the table's interval edges
    ])
    xs = xs[(xs >= 2.2250738585072014e-308) & np.isfinite(xs)]
    ys = smx(xs)
    worst = 0.0
    for x, y in zip(xs.tolist(), ys.tolist()):
        t = mpmath.log(mpmath.mpf(x))
        err = abs(float(mpmath.mpf(y) - t))
        bound = 4e-16 + 2.3e-16 * abs(float(t))
        assert err <= bound, (x, y, float(t), err)
        worst = max(worst, err / bound)
    assert worst > 0.01                      # (the check is not vacuous)


def test_everything_else_goes_to_the_library(smx):
    xs = np.array([0.0, -0.0, -1.0, 5e-324, 1e-310, np.inf, -np.inf, np.nan])
    with np.errstate(all="ignore"):
        want = np.log(xs)
    got = smx(xs)
    for g, w in zip(got, want):
        assert (g != g and w != w) or g == w


def test_power_of_two_row_scaling_of_the_masked_scorer_is_exact():
    """score_masked_mfma_kernel feeds the mask bit of MFMA step st as the double whose high word is the single exponent bit
    27 + st (2^-895, 2^-767, 2^-511, 2) and stores the rows that step consumes multiplied by the inverse power of two
    (batch.hip): every product must be EXACTLY the unscaled term.  Checked here in IEEE double arithmetic over the magnitudes
    row terms can take (|x| from 1e-300 to 2^126: C.mu of read counts up to 2^63, r ln(C.mu), M3's 1e-26 residues), for zeros,
    infinities and NaN."""
    import struct
    a = [struct.unpack("<d", struct.pack("<Q", (1 << (27 + st)) << 32))[0] for st in range(4)]
    assert a == [2.0 ** -895, 2.0 ** -767, 2.0 ** -511, 2.0]
    ex = [895, 767, 511, -1]
    rng = np.random.RandomState(3)
    x = np.concatenate([np.exp(rng.uniform(np.log(1e-300), np.log(2.0 ** 126), 20000)) * rng.choice([-1.0, 1.0], 20000),
                        rng.uniform(0, 1e7, 5000), -rng.uniform(0, 1e9, 5000), np.array([0.0, -0.0, 1e-26, 2.0 ** 126, -(2.0 ** 126)])])
    with np.errstate(all="ignore"):
        for st in range(4):
            stored = np.ldexp(x, ex[st])
            assert np.isfinite(stored).all()
            assert np.array_equal(a[st] * stored, x)
            for special in (np.inf, -np.inf):
                assert a[st] * np.ldexp(special, ex[st]) == special
            assert np.isnan(a[st] * np.ldexp(np.nan, ex[st]))
            assert np.isnan(0.0 * np.ldexp(-np.inf, ex[st]))        # a masked row with ln 0: NaN, like log(0) * 0 in CalcAllC.L3 (Q10)
