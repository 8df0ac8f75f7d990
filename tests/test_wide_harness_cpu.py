"""
The CPU half of tests/test_gpu_wide.py, runnable without a GPU: the instance generator for more than 64 intervals and the
oracle-side worker pool (start method `spawn`, as on the GPU box), on a short prefix of one instance.
"""
import multiprocessing as mp

import numpy as np

import theta_oracle as orc
import test_gpu_wide as wide


def test_wide_instance_and_spawned_oracle_pool():
    m = 72
    rs, rNs, order, truth, lb, ub = wide._wide_instance(m, 502, 1)
    assert len(rs) == m and all(l <= min(t) and max(t) <= u for l, u, t in zip(lb, ub, truth.tolist()))
    cnt = orc.count_n3_exact(m, 2, lb, ub)
    assert 50 < cnt <= 20000
    seq = []
    for rows in orc.enumerate_n3(m, 2, lb, ub):
        seq.append(rows)
        if len(seq) == 24:
            break
    seq = np.array(seq, dtype=np.uint8)
    chunks = np.array_split(np.arange(len(seq)), 3)
    with mp.get_context("spawn").Pool(3) as pool:
        parts = pool.map(wide._oracle_solve_chunk, [(seq[c], rs, rNs) for c in chunks], chunksize=1)
    table = [t for part in parts for t in part]
    assert len(table) == 24
    direct = wide._oracle_solve_chunk((seq[:4], rs, rNs))
    assert table[:4] == direct
