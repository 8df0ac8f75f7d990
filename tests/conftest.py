import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))  # the oracle is test infrastructure (tests only)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "perf: timing assertions, apart from the parity tests (THETA_RUN_PERF=1)")


# The full-size variants of the cases that were cut for the GPU suite's time (round-5 advice: kept, behind a switch):
#   THETA_RUN_SLOW=1 python -m pytest tests -m gpu -k "redo_ladder or tiny or tau3 or two_hundred or prefix_bound or fp64_sieve"
SLOW = os.environ.get("THETA_RUN_SLOW") == "1"


def unfl(x):
    """Inverse of make_golden.fl: JSON-safe float back to float."""
    if isinstance(x, str):
        return float(x)
    return x


def rank_deficient(C):
    """C: (B, m, 2) tumour columns.  True where the points (x_i, y_i) of a matrix lie on one line, i.e. (tau, x, y) are linearly
    dependent -- the candidates the search hands to the reference's own procedure (csrc/n3_core.hpp: N3Line).  Exact: the
    rank of the integer matrix [1, x, y] through its 3x3 Gram determinant."""
    import numpy as np
    C = np.asarray(C, np.int64)
    A = np.concatenate([np.ones(C.shape[:2] + (1,), np.int64), C], axis=2)          # (B, m, 3)
    G = np.einsum("bij,bik->bjk", A, A)                                               # (B, 3, 3) integer Gram matrices
    det = (G[:, 0, 0] * (G[:, 1, 1] * G[:, 2, 2] - G[:, 1, 2] * G[:, 2, 1])
           - G[:, 0, 1] * (G[:, 1, 0] * G[:, 2, 2] - G[:, 1, 2] * G[:, 2, 0])
           + G[:, 0, 2] * (G[:, 1, 0] * G[:, 2, 1] - G[:, 1, 1] * G[:, 2, 0]))
    return det == 0


def load_json(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def gold():
    return load_json
