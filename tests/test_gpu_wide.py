"""
n = 3 searches over more than 64 intervals (run with `-m gpu`): the complete `best` list against the oracle's port of the
reference driver on instances of 72, 100 and 128 intervals.  The oracle side runs in worker processes started with `spawn`
(fresh interpreters: nothing of the GPU runtime of the pytest process is inherited).
"""
import multiprocessing as mp
import os
import warnings

import numpy as np
import pytest

import campaign
import theta_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import theta_amd
    return theta_amd.default_context()


def _oracle_solve_chunk(args):
    rows, rs, rNs = args
    warnings.simplefilter("ignore")
    out = []
    for c in rows:
        s = orc.solve_n3(orc.rows_to_matrix_n3([tuple(x) for x in c], 2), rs, rNs)
        out.append(None if s is None else ([float(x) for x in s[0]], float(s[1])))
    return out


def _wide_instance(m, seed, free, kmax=4):
    """m > 64 intervals: truth = a valid DFS path (non-decreasing a, b = a, the last three rows b = a + 1); the bounds pin every
    interval to its truth except around the places where the copy number steps up -- spread over the whole matrix, before and
    after depth 64 --, where the `free` rows before / after a step may take the neighbouring value too."""
    rng = np.random.RandomState(seed)
    a = np.sort(rng.randint(0, kmax, m))
    b = a.copy()
    b[m - 3:] += 1
    L = rng.randint(2_000_000, 20_000_000, m)
    rN = np.maximum(rng.poisson(L * 0.01), 1)
    mu = np.array([0.3, 0.45, 0.25])
    p = rN * (2 * mu[0] + a * mu[1] + b * mu[2])
    p = p / p.sum()
    r = rng.multinomial(int(rN.sum() * 1.2), p)
    # (the generated data is already in ratio order up to noise; sort_r decides, and bounds follow the sorted order)
    rs, rNs, order = orc.sort_r([int(x) for x in rN], [int(x) for x in r])
    truth = np.stack([a, b], 1)[order]
    lb = truth.min(axis=1).copy()
    ub = truth.max(axis=1).copy()
    t0 = truth[:, 0]
    for i in range(1, m):
        if t0[i] > t0[i - 1]:
            for j in range(free):
                if i + j < m:
                    lb[i + j] = min(lb[i + j], t0[i - 1])
                if i - 1 - j >= 0:
                    ub[i - 1 - j] = max(ub[i - 1 - j], t0[i])
    lb, ub = orc.check_bound_order(lb.tolist(), ub.tolist())
    return rs, rNs, order, truth, [int(x) for x in lb], [int(x) for x in ub]


@pytest.mark.parametrize("m,seed,free", [(100, 501, 1), (72, 502, 1), (128, 503, 1)])
def test_n3_search_over_more_than_64_intervals_against_the_oracle(ctx, m, seed, free):
    import theta_amd
    from theta_amd.search import do_optimization_single
    rs, rNs, order, truth, lb, ub = _wide_instance(m, seed, free)
    p = theta_amd.Problem(ctx, 3, m, 2, rs, rNs, lb, ub, 1.0)
    cnt = p.count
    assert cnt == orc.count_n3_exact(m, 2, lb, ub) and 50 < cnt <= 20000, cnt
    p.close()
    seq = np.array(list(orc.enumerate_n3(m, 2, lb, ub)), dtype=np.uint8)
    best = do_optimization_single(3, m, 4, 2, list(lb), list(ub), rs, rNs, 1.0, order, False, False)
    procs = max(1, min(64, (os.cpu_count() or 2) - 2))
    chunks = np.array_split(np.arange(cnt), procs)
    with mp.get_context("spawn").Pool(procs) as pool:
        parts = pool.map(_oracle_solve_chunk, [(seq[c], rs, rNs) for c in chunks], chunksize=1)
    table = [t for part in parts for t in part]
    first = orc.solve_n3(orc.first_matrix_n3(m, 2), rs, rNs)
    seq_solns = ([(orc.first_matrix_n3(m, 2), first)] if first is not None else []) + \
                [(orc.rows_to_matrix_n3([tuple(x) for x in seq[k]], 2), table[k]) for k in range(cnt) if table[k] is not None]
    ref, lowest = [], float("inf")
    for Cm, sol in seq_solns:
        L = sol[1]
        if orc.is_close(L, lowest):
            ref.append((orc.reverse_sort_C(Cm, order), sol[0], L, None))
        elif L < lowest:
            ref, lowest = [(orc.reverse_sort_C(Cm, order), sol[0], L, None)], L
    assert campaign.compare_best(campaign.best_to_plain(best), campaign.best_to_plain(ref)) == ""
    # the finalists' matrices come back whole (rows 64.. included), in the reference's enumeration order
    p = theta_amd.Problem(ctx, 3, m, 2, rs, rNs, lb, ub, 1.0)
    res = p.search(0, cnt, window=0.5)
    for rk, Cm in zip(res["rank"], res["C"]):
        assert np.array_equal(Cm, seq[rk])
    # the materialised generator holds 128 intervals too (round 3): the whole space, ragged sub-ranges, the device variant
    assert np.array_equal(p.enumerate(0, cnt), seq)
    for b, c in ((cnt // 3, min(777, cnt - cnt // 3)), (cnt - 5, 5), (1, 1)):
        assert np.array_equal(p.enumerate(b, c), seq[b:b + c])
    p.close()


def test_get_values_dump_over_more_than_64_intervals(ctx, tmp_path):
    """--GET_VALUES (RunTHetA.py:210-215) for an n=3 search of 72 intervals: line for line the oracle's trace of the reference
    driver (enumerate + theta_solve_batch; the fused kernel's dump stays at 64 intervals and is not on this path)."""
    import theta_amd.search as S
    m = 72
    rs, rNs, order, truth, lb, ub = _wide_instance(m, 502, 1)
    S.pre = str(tmp_path / "dumpwide")
    try:
        S.do_optimization_single(3, m, 4, 2, list(lb), list(ub), rs, rNs, 1.0, order, False, True)
    finally:
        pre, S.pre = S.pre, "theta"
    lines = [l.rstrip("\n").split("\t") for l in open(pre + ".likelihoods")]
    seq = np.array(list(orc.enumerate_n3(m, 2, lb, ub)), dtype=np.uint8)
    procs = max(1, min(64, (os.cpu_count() or 2) - 2))
    chunks = np.array_split(np.arange(len(seq)), procs)
    with mp.get_context("spawn").Pool(procs) as pool:
        parts = pool.map(_oracle_solve_chunk, [(seq[c], rs, rNs) for c in chunks], chunksize=1)
    table = [t for part in parts for t in part]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        first = orc.solve_n3(orc.first_matrix_n3(m, 2), rs, rNs)
    want = ([("0" * m, float(first[0][0]), float(first[1]))] if first is not None else []) + \
           [("".join(str(int(v)) for v in seq[k][:, 0]), t[0][0], t[1]) for k, t in enumerate(table) if t is not None]
    assert len(lines) == len(want) > 50
    for (col, mu0, nll), (wcol, wmu0, wnll) in zip(lines, want):
        assert col == wcol
        assert (float(nll) != float(nll) and wnll != wnll) or abs(float(nll) - wnll) <= 1e-6 * abs(wnll)
        assert abs(float(mu0) - wmu0) < 1e-6 or (float(mu0) != float(mu0) and wmu0 != wmu0)
