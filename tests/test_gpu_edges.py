"""
Edge cases of the C-ABI path on the GPU: empty and tiny ranges, the smallest and largest supported
shapes, error codes, capacity growth, ragged bounds, window semantics, list-overflow recovery.
"""
import numpy as np
import pytest

import theta_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import theta_amd
    return theta_amd.default_context()


def test_error_codes_and_messages(ctx):
    import theta_amd
    from theta_amd import _lib
    with pytest.raises(theta_amd.ThetaError) as e:
        theta_amd.Problem(ctx, 4, 5, 2, [1] * 5, [1] * 5, [0] * 5, [2] * 5)
    assert e.value.code == _lib.ERR_ARG and "n must be 2 or 3" in str(e.value)
    with pytest.raises(theta_amd.ThetaError):
        theta_amd.Problem(ctx, 2, 1, 2, [1], [1], [0], [2])                      # m >= 2 (FileIO.py:437-439)
    with pytest.raises(theta_amd.ThetaError):
        theta_amd.Problem(ctx, 2, 3, 2, [1, 2, 3], [1, 0, 3], [0] * 3, [2] * 3)  # normal count 0
    wide = theta_amd.Problem(ctx, 3, 3, 2, [1, 2, 3], [1, 1, 3], [0] * 3, [9] * 3)  # n=3, 72 rows within the bounds: a mix-only problem (no ranks)
    assert wide.count == 2 ** 128 - 1
    for call in (lambda: wide.search(0, 10), lambda: wide.enumerate(0, 10), lambda: wide.values(0, 10), lambda: wide.bnb(float("inf"))):
        with pytest.raises(theta_amd.ThetaError) as e:
            call()
        assert e.value.code == _lib.ERR_ARG and "no ranks" in str(e.value)
    wide.close()
    with pytest.raises(theta_amd.ThetaError):
        theta_amd.Problem(ctx, 3, 3, 2, [1, 2, 3], [1, 1, 3], [0] * 3, [16] * 3)  # n=3: copy numbers up to 15
    with pytest.raises(theta_amd.ThetaError):
        theta_amd.Problem(ctx, 3, 257, 2, [1] * 257, [1] * 257, [0] * 257, [2] * 257)  # n=3: four prefix intervals per lane at most (256 intervals)
    p = theta_amd.Problem(ctx, 2, 3, 2, [5, 6, 7], [5, 5, 5], [2, 2, 2], [1, 1, 1])  # lb > ub: nothing to enumerate
    assert p.count == 0
    with pytest.raises(theta_amd.NoCandidates):
        p.search(0, 0)
    p2 = theta_amd.Problem(ctx, 2, 3, 2, [5, 6, 7], [5, 5, 5], [0] * 3, [2] * 3)
    with pytest.raises(theta_amd.ThetaError):
        p2.search(0, p2.count + 1)                                                # range beyond the end
    res = p2.search(4, 4)
    assert len(res["rank"]) == 0 and res["stats"]["evaluated"] == 0              # empty range is fine


def test_driver_exits_like_reference_when_no_candidates(ctx):
    from theta_amd.search import do_optimization_single
    with pytest.raises(SystemExit) as e:
        do_optimization_single(2, 3, 2, 2, [2, 2, 2], [1, 1, 1], [5, 6, 7], [5, 5, 5], 1.0, [0, 1, 2])
    assert e.value.code == 1                                                      # RunTHetA.py:217-219


def test_smallest_shapes_m2(ctx):
    import theta_amd
    r, rN = [900, 1400], [1000, 1000]
    for n in (2, 3):
        p = theta_amd.Problem(ctx, n, 2, 2, r, rN, [0, 0], [3, 3])
        seq = list(orc.enumerate_n2(2, 2, [0, 0], [3, 3])) if n == 2 else list(orc.enumerate_n3(2, 2, [0, 0], [3, 3]))
        assert p.count == len(seq)
        got = p.enumerate(0, p.count)
        assert np.array_equal(got, np.array(seq, dtype=np.uint8).reshape(got.shape))
        nll, mu, st = p.values(0, p.count)
        assert st["evaluated"] == p.count
        for i, c in enumerate(seq):
            C = orc.col_to_matrix_n2(c, 2) if n == 2 else orc.rows_to_matrix_n3(c, 2)
            s = orc.solve(C, r, rN, 1)
            if n == 2:
                assert (s is None) == bool(np.isnan(nll[i]))
            if s is not None and not np.isnan(nll[i]) and np.isfinite(s[1]):
                assert s[1] >= nll[i] * (1 - 1e-9)


def test_largest_shapes(ctx):
    """n=2 with m=256 intervals and copy numbers up to 15; n=3 with m=64 and the full K=7 alphabet."""
    import theta_amd
    rng = np.random.RandomState(3)
    m = 256
    rN = rng.randint(5000, 9000, m)
    c = np.sort(rng.randint(0, 16, m))                  # copy numbers up to THETA_MAX_COPY
    r = rng.poisson(rN * (2 * 0.4 + c * 0.6) / 3.0)
    rs, rNs, order = orc.sort_r([int(x) for x in rN], [int(x) for x in r])
    lbv, ubv = c.copy(), c.copy()
    lbv[-60:] = np.maximum(lbv[-60:] - 1, 0)            # the last 60 intervals are free within -1: 3 696 candidates
    p = theta_amd.Problem(ctx, 2, m, 2, rs, rNs, lbv.tolist(), ubv.tolist())
    assert p.count == orc.count_n2(m, lbv.tolist(), ubv.tolist()) == 3696
    res = p.search(0, p.count)
    assert res["stats"]["evaluated"] == p.count
    k = int(np.argmin(res["nll"]))
    s = orc.solve_n2(orc.col_to_matrix_n2(res["C"][k], 2), rs, rNs, 1)
    assert s is not None and abs(s[1] - res["nll"][k]) <= 1e-9 * abs(s[1])
    # n=3: m=64 with the full K=7 alphabet has more than 2^128 matrices: the count saturates (round 4; such a space is a legal
    # problem whose rank ranges below 2^128 are searched like any others -- tests/test_gpu_round4.py)
    m = 64
    r3 = [int(x) for x in rng.randint(1000, 5000, m)]
    rN3 = [int(x) for x in rng.randint(1000, 5000, m)]
    psat = theta_amd.Problem(ctx, 3, m, 2, r3, rN3, [0] * m, [7] * m)
    assert psat.count == 2 ** 128 - 1
    psat.close()
    # ... m=64, K=7 with rising lower bounds fits
    lb3 = [min(6, i // 9) for i in range(m)]
    ub3 = [min(7, v + 1) for v in lb3]
    p3 = theta_amd.Problem(ctx, 3, m, 2, r3, rN3, lb3, ub3)
    assert p3.count == orc.count_n3_exact(m, 2, lb3, ub3) > 10 ** 23      # 128-bit table vs Python integers
    for start in (0, p3.count // 2, p3.count - 5000):
        C = p3.enumerate(start, 300)
        assert C.shape == (300, m, 2) and C.max() <= 7 and (C.min(axis=2) >= np.array(lb3)[None, :]).all()
        nll, mu, st = p3.values(start, 5000)
        assert st["evaluated"] == 5000
        ok, mu_b, nll_b, _ = ctx.solve_batch(3, 2, r3, rN3, p3.enumerate(start, 5000), 1.0, want_vals=False)
        ok = ok & ~ctx.last_solve_fallback          # (optimum outside the simplex: the batch solver reports the nu = 1/3 fallback)
        both = ok & ~np.isnan(nll)
        assert (ok != ~np.isnan(nll)).sum() <= 10
        if both.any():
            assert (np.abs(nll_b[both] - nll[both]) / nll[both]).max() < 1e-10


def test_ragged_bounds_and_capacity_growth(ctx):
    """Bounds that need _check_bound_order; a window so wide that the first capacity guess is too small."""
    import theta_amd
    r, rN, L, Ct, mu = orc.synth_counts(9, 2, 4, 55)
    rs, rNs, order = orc.sort_r(rN, r)
    lb = [0, 1, 0, 2, 1, 0, 3, 1, 2]
    ub = [4, 2, 3, 4, 2, 4, 4, 3, 4]
    p = theta_amd.Problem(ctx, 2, 9, 2, rs, rNs, lb, ub)
    assert p.count == orc.count_n2(9, lb, ub)
    big = p.search(0, p.count, window=1e300, cap=4)          # every accepted candidate is "within the window"
    nll, mu_, st = p.values(0, p.count)
    assert len(big["rank"]) == int((~np.isnan(nll)).sum()) > 4
    assert big["rank"] == sorted(big["rank"])
    # (the search starts a candidate's Newton iteration where its bound evaluation left it, the dump from the previous root: the
    # same root to the last bits, not to the last bit)
    assert np.allclose(big["nll"], nll[~np.isnan(nll)], rtol=1e-13, atol=0)
    zero = p.search(0, p.count, window=0.0)
    assert len(zero["rank"]) >= 1 and abs(zero["nll"].min() - np.nanmin(nll)) <= 1e-13 * abs(np.nanmin(nll))


def test_tie_list_overflow_is_recovered(ctx):
    """More than 2^20 records within the window: the library re-runs with the tightened threshold."""
    import theta_amd
    m = 40
    rs, rNs = [1000 + 3 * i for i in range(m)], [1000] * m
    p = theta_amd.Problem(ctx, 2, m, 2, rs, rNs, [0] * m, [6] * m)   # C(46,6) = 9.4e6 candidates
    assert p.count == 9366819
    res = p.search(0, p.count, window=0.5)
    nll, mu_, st = p.values(0, p.count)
    assert res["stats"]["evaluated"] == p.count
    assert res["nll"].min() == np.nanmin(nll)
    assert res["rank"][int(np.argmin(res["nll"]))] == int(np.nanargmin(nll))


def test_ranges_beyond_one_call_are_walked_in_pieces(ctx):
    """Problem.search splits ranges larger than one theta_search call can take and merges the finalists."""
    import theta_amd
    r, rN, L, Ct, mu = orc.synth_counts(12, 3, 3, 91)
    rs, rNs, order = orc.sort_r(rN, r)
    p = theta_amd.Problem(ctx, 3, 12, 2, rs, rNs, [0] * 12, [3] * 12)
    whole = p.search(0, p.count, window=0.5)
    old = dict(theta_amd.Problem.MAX_PER_CALL)
    try:
        theta_amd.Problem.MAX_PER_CALL[3] = max(1000, p.count // 7)
        pieces = p.search(0, p.count, window=0.5)
    finally:
        theta_amd.Problem.MAX_PER_CALL.update(old)
    assert pieces["rank"] == whole["rank"]
    assert np.array_equal(pieces["nll"], whole["nll"]) and np.array_equal(pieces["C"], whole["C"])
    assert pieces["stats"]["evaluated"] == whole["stats"]["evaluated"] == p.count


def test_enumerate_all_shapes_and_the_device_variant(ctx):
    """
    The wave-cooperative generator against the oracle's enumeration: even and odd m (32-bit and 16-bit store paths),
    a prefix shorter than a store word, ranges that start and end inside a prefix, requests of several tasks --
    and theta_enumerate_device (same bytes, written to device memory the caller owns; here from hipMalloc via ctypes).
    """
    import ctypes as C
    import theta_amd
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    rng = np.random.RandomState(99)
    for (m, k) in [(7, 2), (8, 2), (9, 3), (10, 2), (11, 2)]:
        r = rng.randint(1000, 5000, m).tolist()
        rN = rng.randint(1000, 5000, m).tolist()
        lb, ub = [0] * m, [k] * m
        ref = np.array(list(orc.enumerate_n3(m, 2, lb, ub)), dtype=np.uint8)
        p = theta_amd.Problem(ctx, 3, m, 2, r, rN, lb, ub, 1.0)
        assert p.count == len(ref)
        assert np.array_equal(p.enumerate(0, p.count), ref)
        for b, c in [(1, 1), (p.count // 3, min(20000, p.count - p.count // 3)), (p.count - 2, 2)]:
            assert np.array_equal(p.enumerate(b, c), ref[b:b + c])
        nbytes = p.count * m * 2 + 64
        dptr = C.c_void_p()
        assert hip.hipMalloc(C.byref(dptr), nbytes) == 0
        try:
            assert hip.hipMemset(dptr, 0xEE, nbytes) == 0
            ms = p.enumerate_device(5, p.count - 9, dptr.value + 16)
            assert ms > 0
            with pytest.raises(theta_amd.ThetaError):                     # the output pointer must be 4-byte aligned
                p.enumerate_device(5, 10, dptr.value + 2)
            got = np.zeros(nbytes, np.uint8)
            assert hip.hipMemcpy(got.ctypes.data_as(C.c_void_p), dptr, nbytes, 2) == 0          # device -> host
        finally:
            hip.hipFree(dptr)
        assert (got[:16] == 0xEE).all() and (got[16 + (p.count - 9) * m * 2:] == 0xEE).all()       # nothing outside
        assert np.array_equal(got[16:16 + (p.count - 9) * m * 2].reshape(-1, m, 2), ref[5:p.count - 4])


def test_burst_generator_against_oracle_and_lane_private_generator(ctx, monkeypatch):
    """
    The burst generator (n3_enum.hip: one output stream per wave, breadth-first over the last rows) against the oracle on
    every small shape class -- m = 2..6 (1, 2 and 4 expanded rows, 32-bit and 16-bit units), ragged bounds, windows that
    start and end inside a prefix -- and against the lane-private generator (THETA_ENUM_LEGACY=1) on the bench shape,
    where a request spans many wave tasks and bursts are cut by the list capacities.
    """
    import theta_amd
    rng = np.random.RandomState(17)
    shapes = [(2, 2, None), (3, 2, None), (4, 3, None), (5, 2, None), (6, 3, None), (7, 3, None), (12, 2, None),
              (9, 4, ([0, 0, 0, 1, 1, 1, 2, 2, 2], [1, 2, 2, 2, 3, 3, 3, 4, 4])),
              (10, 5, ([0, 0, 1, 1, 1, 2, 2, 2, 2, 3], [1, 1, 2, 3, 3, 3, 4, 5, 5, 5]))]
    for m, k, bounds in shapes:
        lb, ub = bounds if bounds else ([0] * m, [k] * m)
        r = rng.randint(1000, 5000, m).tolist()
        rN = rng.randint(1000, 5000, m).tolist()
        ref = np.array(list(orc.enumerate_n3(m, 2, lb, ub)), dtype=np.uint8)
        p = theta_amd.Problem(ctx, 3, m, 2, r, rN, lb, ub, 1.0)
        assert p.count == len(ref)
        for levels in (None, "6"):
            if levels:
                monkeypatch.setenv("THETA_ENUM_LEVELS", levels)
            assert np.array_equal(p.enumerate(0, p.count), ref), (m, k, levels)
            for _ in range(4):
                b = int(rng.randint(0, p.count))
                c = int(rng.randint(1, min(p.count - b, 70000) + 1))
                assert np.array_equal(p.enumerate(b, c), ref[b:b + c]), (m, k, b, c, levels)
            monkeypatch.delenv("THETA_ENUM_LEVELS", raising=False)
        p.close()
    # bench shape: 3 tasks' worth from an odd start, both generators
    import bench
    for m, k in ((50, 6), (49, 5), (24, 7)):
        r, rN, order = bench.synth(seed=3, m=m, n=3, k=k)
        p = theta_amd.Problem(ctx, 3, m, 2, r, rN, [0] * m, [k] * m, 1.0)
        for b, c in ((p.count // 7 + 12345, 3 * 8192 + 777), (0, 50000), (p.count - 40001, 40001)):
            monkeypatch.setenv("THETA_ENUM_LEGACY", "1")
            old = p.enumerate(b, c)
            monkeypatch.delenv("THETA_ENUM_LEGACY")
            for levels in (None, "6", "4", "2", "1"):          # expanded rows: the instance's own choice, and forced
                if levels:
                    monkeypatch.setenv("THETA_ENUM_LEVELS", levels)
                new = p.enumerate(b, c)
                monkeypatch.delenv("THETA_ENUM_LEVELS", raising=False)
                assert np.array_equal(new, old), (m, k, b, c, levels)
        p.close()


# (the --GET_VALUES dump is compared line for line, with no allowance, in tests/test_gpu_cli_matrix.py)


def test_driver_exits_cleanly_when_the_search_is_beyond_the_library(ctx, capsys):
    """n=3 with 260 intervals (the library holds 256), a space beyond 2^128 matrices (64 intervals, bounds [0, 7]: a legal problem since round 4, not a legal exhaustive search), or one no
    search can finish (70 intervals, bounds [0, 2]: 2.5e34 matrices -- refused at once, nothing of that size is materialised
    on the host): a message and exit(1) like the reference's other input errors, not a traceback."""
    import theta_amd
    from theta_amd.search import do_optimization_single
    rng = np.random.RandomState(4)
    p = theta_amd.Problem(ctx, 3, 70, 2, [1] * 70, [1] * 70, [0] * 70, [2] * 70)
    assert 2 ** 100 < p.count < 2 ** 128                    # (2.5e34: a legal problem, not a legal exhaustive search)
    with pytest.raises(theta_amd.ThetaError) as e:
        p.search(0, p.count)
    assert e.value.code == theta_amd._lib.ERR_OVERFLOW
    p.close()
    for m, k in ((260, 2), (130, 2), (70, 2), (64, 7)):
        r = rng.randint(1000, 5000, m).tolist()
        rN = rng.randint(1000, 5000, m).tolist()
        with pytest.raises(SystemExit):
            do_optimization_single(3, m, k, 2, [0] * m, [k] * m, r, rN, 1.0, list(range(m)))
        assert "ERROR" in capsys.readouterr().out
