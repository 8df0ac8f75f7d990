"""
Timing assertions, apart from the parity tests: a loaded or cold box fails these for reasons that have nothing to do with
correctness, so they are not part of `-m gpu` (a `-x` run must not stop at them).  Run on the GPU box with
    THETA_RUN_PERF=1 python -m pytest tests/test_gpu_perf.py -m perf -q
The parity tests print the same times without asserting on them (tests/test_gpu_bnb.py).
"""
import os
import time

import pytest

pytestmark = [pytest.mark.perf,
              pytest.mark.skipif(os.environ.get("THETA_RUN_PERF") != "1", reason="timing assertions: set THETA_RUN_PERF=1 on a GPU box")]


@pytest.fixture(scope="module")
def ctx():
    import theta_amd
    return theta_amd.default_context()


@pytest.mark.parametrize("K,seed", [(4, 7), (6, 4242)])
def test_whole_space_of_configs_3_and_4_within_seconds(ctx, K, seed):
    """RunTHetA.py:173-220 over 4e27 / 2.6e38 matrices: a fraction of a second through the mixture-space search."""
    import bench
    from theta_amd import search as S
    r, rN, order = bench.synth(seed=seed, m=50, n=3, k=K)
    S.do_optimization_single(3, 50, K, 2, [0] * 50, [K] * 50, r, rN, 1.0, order, False, False)      # (warm: context, tables)
    t0 = time.time()
    S.do_optimization_single(3, 50, K, 2, [0] * 50, [K] * 50, r, rN, 1.0, order, False, False)
    assert time.time() - t0 < 5.0


def test_lazy_counting_table_halves_problem_creation(ctx, monkeypatch):
    """config 5's shape: theta_problem_create leaves the 2 GB counting table to the first call that takes ranks."""
    import bench
    import theta_amd
    r, rN, _order = bench.synth(seed=55, m=200, n=3, k=7)
    times = {}
    for lazy in ("1", "0"):
        monkeypatch.setenv("THETA_N3_LAZY_TABLE", lazy)
        t0 = time.time()
        p = theta_amd.Problem(ctx, 3, 200, 2, r, rN, [0] * 200, [7] * 200, 1.0)
        times[lazy] = time.time() - t0
        p.close()
    assert times["1"] < 0.5 * times["0"], times
