"""
The n=2 "render" generator (theta_amd/csrc/n2_render.hpp: records produced 128 bytes at a time by scatter + prefix sum) on the
CPU: tools/n2_render_emul.hip executes the kernel body of n2_enumerate_render_kernel lane by lane -- the same N2_HD code the
GPU kernel compiles -- and its output is compared byte for byte with the oracle's port of Enumerator._generate_next_C_2
(Enumerator.py:119-152), over interval counts, alphabets, bounds and rank ranges that hit every boundary case of the scheme
(records straddling lines and words, runs cut short by the end of the range, break-points at 0 and at equal positions).
The kernel is off by default (THETA_N2_ENUM_RENDER=1) until it has run on the GPU.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import theta_oracle as orc
from conftest import ROOT

LIB = os.path.join(ROOT, "build_ab", "libn2_emul.so")


@pytest.fixture(scope="module")
def emul():
    src = os.path.join(ROOT, "tools", "n2_render_emul.hip")
    hdrs = [os.path.join(ROOT, "theta_amd", "csrc", f) for f in ("n2_render.hpp", "n2_cand.hpp", "n2.hpp")]
    prod = os.path.join(ROOT, "theta_amd", "libtheta_hip.so")
    if not os.path.exists(prod):
        pytest.skip("libtheta_hip.so not built")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(f) for f in [src, prod] + hdrs):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", src,
                        "-L" + os.path.join(ROOT, "theta_amd"), "-ltheta_hip", "-Wl,-rpath," + os.path.join(ROOT, "theta_amd"), "-o", LIB],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    lib = C.CDLL(LIB)
    lib.n2_emul_enumerate.argtypes = [C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_ulonglong, C.c_ulonglong, C.c_int,
                                      C.POINTER(C.c_uint8), C.POINTER(C.c_ulonglong)]
    return lib


def _emulate(lib, m, lb, ub, begin, count, T=0):
    lbv, ubv = np.asarray(lb, np.int32), np.asarray(ub, np.int32)
    out = np.full(count * m + 64, 0xEE, np.uint8)                     # (a guard zone behind the output)
    total = C.c_ulonglong()
    rc = lib.n2_emul_enumerate(m, lbv.ctypes.data_as(C.POINTER(C.c_int32)), ubv.ctypes.data_as(C.POINTER(C.c_int32)), begin, count, T,
                               out.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(total))
    assert rc == 0, rc
    assert (out[count * m:] == 0xEE).all()                           # nothing is written past the last record
    return out[:count * m].reshape(count, m), total.value


def _reference(m, lb, ub):
    return np.array(list(orc.enumerate_n2(m, 2, list(lb), list(ub))), np.uint8)


CASES = [
    # m, lb, ub                                   (the whole space is compared, then sub-ranges)
    (4, [0] * 4, [3] * 4),
    (7, [0] * 7, [3] * 7),
    (8, [0] * 8, [7] * 8),                        # KV = 8: the largest alphabet of the 8-value kernel
    (9, [0] * 9, [4] * 9),
    (13, [0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 2, 3], [1, 2, 2, 2, 3, 3, 4, 4, 4, 5, 5, 5, 5]),
    (16, [0] * 16, [2] * 16),                     # m a divisor of 128: records never straddle a line
    (25, [0] * 25, [2] * 25),
    (12, [0] * 12, [9] * 12),                     # KV = 16 kernel (values up to 9)
    (6, [0] * 6, [15] * 6),                       # the full 16-value alphabet
    (50, [0] * 44 + [1] * 6, [1] * 20 + [2] * 30),
    (33, [2] * 33, [3] * 33),                     # a lower bound above 0: every record starts with break-points at 0
    (100, [0] * 100, [1] * 100),
    (130, [0] * 130, [1] * 130),                  # a record longer than a line
    (5, [0, 0, 0, 0, 0], [0, 0, 0, 0, 3]),       # almost everything pinned
]


@pytest.mark.parametrize("m,lb,ub", CASES)
def test_rendered_records_equal_the_reference_enumeration(emul, m, lb, ub):
    ref = _reference(m, lb, ub)
    got, total = _emulate(emul, m, lb, ub, 0, len(ref))
    assert total == len(ref) == orc.count_n2(m, list(lb), list(ub))
    assert np.array_equal(got, ref)
    # sub-ranges: ragged begins and counts (runs cut short, a last run of one record, single-record ranges)
    rng = np.random.RandomState(m)
    for _ in range(12):
        b = int(rng.randint(0, len(ref)))
        c = int(rng.randint(1, min(len(ref) - b, 5000) + 1))
        got, _ = _emulate(emul, m, lb, ub, b, c)
        assert np.array_equal(got, ref[b:b + c]), (b, c)
    got, _ = _emulate(emul, m, lb, ub, len(ref) - 1, 1)
    assert np.array_equal(got, ref[-1:])


@pytest.mark.parametrize("m,T", [(7, 128), (7, 256), (25, 128), (50, 64), (50, 192), (12, 32), (12, 160)])
def test_other_run_lengths(emul, m, T):
    """Any run length with T m a multiple of 128 (the launcher picks the smallest >= 32)."""
    lb, ub = [0] * m, [3 if m < 20 else 2] * m
    ref = _reference(m, lb, ub)
    n = min(len(ref), 40000)
    got, _ = _emulate(emul, m, lb, ub, 3, n - 3, T)
    assert np.array_equal(got, ref[3:n])


@pytest.mark.parametrize("seed", range(12))
def test_random_ragged_bounds(emul, seed):
    """Random per-interval bounds (not monotone as given: _check_bound_order adjusts them, Enumerator.py:90-113): both bound
    tables of the successor (first position with lb >= w / ub >= w) at work."""
    import itertools
    rng = np.random.RandomState(100 + seed)
    total = 0
    while total == 0:                               # (ragged bounds often leave nothing: draw until the space is not empty)
        m = int(rng.randint(5, 40))
        lb = np.maximum.accumulate(rng.randint(0, 3, m)) - rng.randint(0, 2, m)      # mostly rising, with dips
        lb = np.maximum(lb, 0)
        ub = np.maximum.accumulate(lb) + rng.randint(0, 4, m)                          # above the adjusted lower bounds, ragged
        total = orc.count_n2(m, lb.tolist(), ub.tolist())
    n = int(min(total, 60000))
    ref = np.array(list(itertools.islice(orc.enumerate_n2(m, 2, lb.tolist(), ub.tolist()), n)), np.uint8)
    got, tot = _emulate(emul, m, lb.tolist(), ub.tolist(), 0, n)
    assert tot == total and np.array_equal(got, ref)
    if total > n:                                   # a range that does not start at rank 0
        b = int(rng.randint(1, n - 1))
        got, _ = _emulate(emul, m, lb.tolist(), ub.tolist(), b, n - b)
        assert np.array_equal(got, ref[b:])
