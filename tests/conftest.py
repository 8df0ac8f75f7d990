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


def unfl(x):
    """Inverse of make_golden.fl: JSON-safe float back to float."""
    if isinstance(x, str):
        return float(x)
    return x


def load_json(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def gold():
    return load_json
