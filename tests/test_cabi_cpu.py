"""CPU-only checks of the C-ABI boundary: the library loads without a GPU and exports every symbol
include/theta_hip.h declares; compute entry points fail loudly (no CPU fallback) when no GPU exists."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

import theta_amd
from theta_amd import _lib


def _declared():
    src = open(os.path.join(ROOT, "include", "theta_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(theta_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 13
    for nm in names:
        assert hasattr(lib, nm), nm
    assert sorted(_lib.EXPORTS) == names


def test_error_string_and_no_cpu_fallback():
    lib = _lib.load()
    h = ctypes.c_void_p()
    rc = lib.theta_create(0, ctypes.byref(h))
    if rc == _lib.THETA_OK:
        lib.theta_destroy(h)
        pytest.skip("GPU present: the no-device path cannot be exercised")
    assert rc == _lib.ERR_HIP
    assert b"HIP device" in lib.theta_last_error() or b"hip" in lib.theta_last_error().lower()
    with pytest.raises(theta_amd.ThetaError):
        theta_amd.Context(0)
    # the operator mirrors must raise too -- nothing silently computes on the CPU
    from theta_amd.search import do_optimization_single
    with pytest.raises(theta_amd.ThetaError):
        do_optimization_single(2, 3, 2, 2, [0, 0, 0], [2, 2, 2], [5, 6, 7], [5, 5, 5], 1.0, [0, 1, 2])


def test_host_tie_replay_semantics():
    """The sequential isClose rule (RunTHetA.py:194-206) on synthetic finalists -- pure host logic."""
    import numpy as np
    from theta_amd.search import replay_ties, isClose, find_mins
    def rec(rank, nll):
        return {"rank": rank, "c": np.array([rank % 3, 2, 2], np.uint8), "mu": np.array([.5, .5]), "nll": nll,
                "vals": np.array([.2, .3, .5])}
    order = [2, 0, 1]
    # first minimum is kept as the reference point: 10.0005 is appended, 9.9996 (within 1e-3 of 10.0) too
    best = replay_ties([rec(5, 10.0005), rec(1, 10.0), rec(9, 9.9996), rec(12, 10.3)], 2, 2, order, False)
    assert [b[2] for b in best] == [10.0, 10.0005, 9.9996]
    # a candidate lower by more than the margin replaces the list
    best = replay_ties([rec(1, 10.0), rec(2, 9.99), rec(3, 9.9905)], 2, 2, order, False)
    assert [b[2] for b in best] == [9.99, 9.9905]
    # quirk Q1: the rank-0 matrix is evaluated twice for n=2
    best = replay_ties([rec(0, 7.0), rec(4, 8.0)], 2, 2, order, True)
    assert [b[2] for b in best] == [7.0, 7.0]
    # rows go back to the original interval order
    C = best[0][0]
    assert C[order[0], 1] == 0 and C[:, 0].tolist() == [2, 2, 2]
    assert best[0][3] == [.3, .5, .2]
    assert isClose([float("nan")], [1.0]) and not isClose([1.0], [1.002])
    a = [(None, None, 5.0, None)]
    b = [(None, None, 5.0004, None), (None, None, 5.0, None)]
    assert find_mins([list(a), [], list(b)]) == a + b


def test_no_torch_in_the_product_path():
    """north_star: host code is Python + numpy over ctypes, no PyTorch -- neither theta_amd nor bench.py imports it."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import theta_amd, theta_amd.search, theta_amd.RunTHetA, theta_amd.CalcAllC; "
            "import importlib.util as u; s = u.spec_from_file_location('bench', %r); m = u.module_from_spec(s); s.loader.exec_module(m); "
            "print('torch' in sys.modules)" % (ROOT, os.path.join(ROOT, "bench.py")))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip() == "False"


def test_comm_argument_checks_without_a_gpu():
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.theta_comm_create(None, 0, 1, b"127.0.0.1", 0, 0, ctypes.byref(h)) == _lib.ERR_ARG      # RCCL needs a context
    assert lib.theta_comm_create(None, 2, 2, b"127.0.0.1", 1234, 1, ctypes.byref(h)) == _lib.ERR_ARG   # rank out of range
    assert lib.theta_comm_create(None, 0, 1, b"127.0.0.1", 0, 1, ctypes.byref(h)) == _lib.THETA_OK     # a world of one, host transport
    c = theta_amd.Comm.__new__(theta_amd.Comm)
    c._h, c.world, c.rank = h, 1, 0
    assert c.allreduce_min([4.0]).tolist() == [4.0] and c.info()["transport"] == "host"
    c.close()
