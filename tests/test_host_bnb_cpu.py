"""CPU tests of the host side of round 5's branch and bound (theta_amd/search.py): the membership test of the reference's n=3 space
(in_space_n3) against the oracle's generator, the order-adjusted bounds, and the split of rank ranges over the ranks of a sharded run."""
import itertools

import numpy as np
import pytest

import theta_oracle as orc
from theta_amd import search as S


@pytest.mark.parametrize("m,K,lb,ub,tau", [(3, 2, [0] * 3, [2] * 3, 2), (4, 3, [0, 0, 1, 1], [2, 3, 3, 3], 2), (3, 4, [0] * 3, [4] * 3, 2),
                                           (4, 2, [0, 1, 0, 0], [2, 2, 2, 2], 2), (3, 3, [0] * 3, [3] * 3, 1), (3, 3, [0] * 3, [3] * 3, 3)])
def test_in_space_is_exactly_the_set_the_reference_generator_yields(m, K, lb, ub, tau):
    """Every one of the (K + 1)^(2 m) integer matrices: in_space_n3 says yes exactly for those Enumerator._generate_next_C_3 yields
    (Enumerator.py:172-242, through the oracle's generator, which the golden enumeration orders pin)."""
    space = set(tuple(map(tuple, rows)) for rows in orc.enumerate_n3(m, tau, list(lb), list(ub)))
    lb2, ub2 = S.adjusted_bounds(lb, ub)
    assert len(space) > 10
    bad = [c for c in itertools.product(itertools.product(range(K + 1), repeat=2), repeat=m)
           if S.in_space_n3(np.array(c), lb2, ub2, tau) != (tuple(c) in space)]
    assert not bad, bad[:3]
    # the batch form (numpy, what the heuristic and mix_records call) says the same of every matrix
    every = np.array(list(itertools.product(itertools.product(range(K + 1), repeat=2), repeat=m)), np.int64)
    got = S.in_space_n3_batch(every, lb2, ub2, tau)
    want = np.array([tuple(map(tuple, c)) in space for c in every.tolist()])
    assert np.array_equal(got, want), np.nonzero(got != want)[0][:3]
    # ... and the generator's order is lexicographic in (b, a) per row, i.e. in the grid slot a + (K + 1) b: what theta_mix_search sorts by
    seq = [tuple(b * (K + 1) + a for a, b in rows) for rows in orc.enumerate_n3(m, tau, list(lb), list(ub))]
    assert seq == sorted(seq)


def test_adjusted_bounds_follow_check_bound_order():
    assert S.adjusted_bounds([1, 0, 2, 1], [3, 2, 3, 1]) == ([1, 1, 2, 2], [1, 1, 1, 1])
    assert S.adjusted_bounds([0, 0], [2, 3]) == ([0, 0], [2, 3])


def test_share_of_ranges_partitions_the_matrices_evenly_and_in_order():
    ranges = [(0, 10), (20, 30), (100, 141), (1000, 1001)]
    total = sum(e - b for b, e in ranges)
    for G in (1, 2, 3, 7):
        parts = [S._share_of_ranges(ranges, g, G) for g in range(G)]
        flat = [x for part in parts for b, e in part for x in range(b, e)]
        assert flat == [x for b, e in ranges for x in range(b, e)]
        sizes = [sum(e - b for b, e in part) for part in parts]
        assert sum(sizes) == total and max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("m,K,lb,ub,tau", [(3, 2, [0] * 3, [2] * 3, 2), (4, 3, [0, 0, 1, 1], [2, 3, 3, 3], 2), (3, 4, [0] * 3, [4] * 3, 2),
                                           (4, 2, [0, 1, 0, 0], [2, 2, 2, 2], 2), (3, 3, [0] * 3, [3] * 3, 1), (3, 3, [0] * 3, [3] * 3, 3),
                                           (5, 2, [0] * 5, [2] * 5, 2), (4, 3, [1, 0, 2, 1], [3, 2, 3, 3], 2), (4, 4, [0] * 4, [4] * 4, 2)])
def test_count_lower_bound_counts_matrices_of_the_reference_space(m, K, lb, ub, tau):
    """theta_count_lower_bound (what lets theta_problem_create defer the counting table of a space of 2^128 matrices or more): the
    number it returns is EXACTLY the number of integer matrices with valid rows within the adjusted bounds along which a + b never
    decreases and whose first off-diagonal row has a < b -- and every one of those is a matrix of the reference's space (the
    oracle's generator): a subset, so a lower bound of the count."""
    from theta_amd import _lib
    space = set(tuple(map(tuple, rows)) for rows in orc.enumerate_n3(m, tau, list(lb), list(ub)))
    lb2, ub2 = S.adjusted_bounds(lb, ub)
    rows = [(a, b) for a in range(K + 1) for b in range(K + 1) if (tau - a) * (tau - b) >= 0]
    family = []
    for c in itertools.product(rows, repeat=m):
        if not all(lb2[i] <= c[i][0] <= ub2[i] and lb2[i] <= c[i][1] <= ub2[i] for i in range(m)):
            continue
        if any(sum(c[i]) < sum(c[i - 1]) for i in range(1, m)):
            continue
        off = [r for r in c if r[0] != r[1]]
        if off and off[0][0] > off[0][1]:
            continue
        family.append(c)
    assert all(c in space for c in family), [c for c in family if c not in space][:3]
    lg = _lib.count_lower_bound_log2(m, tau, lb, ub)
    got = 0 if lg == float("-inf") else round(2.0 ** lg)
    assert got == len(family) <= len(space) and got > 0, (got, len(family), len(space))


def test_count_lower_bound_tells_the_spaces_beyond_2_to_the_128():
    from theta_amd import _lib
    assert _lib.count_lower_bound_log2(200, 2, [0] * 200, [7] * 200) > 300          # BASELINE config 5's shape: ~1e150 matrices
    assert _lib.count_lower_bound_log2(100, 2, [0] * 100, [7] * 100) > 130
    assert _lib.count_lower_bound_log2(50, 2, [0] * 50, [6] * 50) < 127.7             # config 4: 2.61e38 = 2^127.6 matrices, not saturated
