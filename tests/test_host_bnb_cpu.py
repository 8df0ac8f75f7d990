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
