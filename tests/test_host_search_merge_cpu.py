"""
Host side of Problem.search (theta_amd/_lib.py) without a GPU: the walk over the pieces of a rank range and the running
merge of their lists, on a stand-in for the device call.  Ranges no search can finish are refused up front -- nothing of
their size is ever materialised on the host (a 70-interval space with bounds [0, 2] holds 2.5e34 matrices).
"""
import time

import numpy as np
import pytest

from theta_amd import _lib


class _FakeProblem(_lib.Problem):
    """Problem.search over synthetic pieces: piece i holds a few finalists, suspects and all-zero-column entries."""

    def __init__(self, n, m, count, overflow_first_pass=()):
        self.n, self.m, self.count = n, m, count
        self.calls = []
        self.overflow_first_pass = set(overflow_first_pass)
        self.seen = set()
        self.suspects_dropped = 0

    def close(self):
        pass

    def _probe(self, begin, end):
        return float("inf")

    def _data(self, b, e):
        rng = np.random.RandomState(b % (1 << 31))
        k = rng.randint(0, 4)
        nll = 1000.0 + rng.rand(k) * 3.0 - (b % 7) * 0.1 - (b // 1000 % 40) * 0.05     # (minima keep falling for a while)
        ranks = [b + int(x) for x in rng.randint(0, max(1, e - b), k)]
        shape = (k, self.m) if self.n == 2 else (k, self.m, 2)
        res = {"nll": nll, "mu": rng.rand(k, self.n), "rank": ranks, "C": rng.randint(0, 3, shape).astype(np.uint8),
               "stats": {"evaluated": e - b, "accepted": k, "degenerate": 0, "iterations": 1, "terms": 2, "list_overflow": 0, "flops": 3.0,
                         "flops_f32": 1.0, "dismissed": 0, "survivors": 0, "fallback_candidates": 0, "kernel_ms": 0.5, "setup_ms": 0.1,
                         "phase_cycles": [1, 2], "best_nll": float(nll.min()) if k else float("inf"), "rejected_bound": 5.0 + b,
                         "rejected_rank": b}}
        ks = rng.randint(0, 3)
        sus = ([b + 1 + i for i in range(ks)], 1000.0 + rng.rand(ks) * 3.0, rng.randint(0, 3, (ks, self.m, 2)).astype(np.uint8))
        kd = rng.randint(0, 2)
        deg = ([b + 2 + i for i in range(kd)], rng.randint(0, 3, (kd, self.m, 2)).astype(np.uint8))
        return res, sus, deg

    def _piece(self, b, e, window, cap, hint):
        self.calls.append((b, e, hint))
        res, sus, deg = self._data(b, e)
        dropped = 0
        if b in self.overflow_first_pass and b not in self.seen:
            self.seen.add(b)
            dropped = 3
        if self.n != 3:
            return res, ([], np.zeros(0), None), 0, ([], None)
        return res, sus, dropped, deg

    def _search_once(self, begin, end, window, cap):
        raise AssertionError("empty ranges only")


def _brute(p, begin, end, step, window):
    parts = [p._data(b, min(b + step, end)) for b in range(begin, end, step)]
    nll = np.concatenate([q[0]["nll"] for q in parts])
    gmin = nll.min()
    keep = nll <= gmin + window
    ranks = [r for q in parts for r in q[0]["rank"]]
    sl = np.concatenate([q[1][1] for q in parts])
    srk = [r for q in parts for r in q[1][0]]
    return ([r for r, k in zip(ranks, keep) if k], nll[keep], [r for r, l in zip(srk, sl) if l <= gmin + window],
            [r for q in parts for r in q[2][0]], sum(q[0]["stats"]["evaluated"] for q in parts))


@pytest.mark.parametrize("n", [2, 3])
def test_streaming_merge_equals_the_merge_of_all_pieces(n, monkeypatch):
    monkeypatch.setattr(_lib.Problem, "MAX_PER_CALL", {2: 1000, 3: 1000})
    p = _FakeProblem(n, 5, 300 * 1000 + 17)
    out = p.search(0, p.count, window=0.5)
    ranks, nll, srk, drk, evaluated = _brute(p, 0, p.count, 1000, 0.5)
    assert out["rank"] == ranks and np.array_equal(out["nll"], nll)
    assert out["mu"].shape == (len(ranks), n) and out["C"].shape[0] == len(ranks)
    assert out["stats"]["evaluated"] == evaluated == p.count
    assert len(p.calls) == 301 and p.calls[-1][1] == p.count
    # later pieces start from the minimum found so far
    assert all(h2 <= h1 for (_, _, h1), (_, _, h2) in zip(p.calls, p.calls[1:]))
    if n == 3:
        assert list(p.last_suspects[0]) == srk and len(p.last_suspects[1]) == len(srk) == len(p.last_suspects[2])
        assert list(p.last_degenerate[0]) == drk and len(p.last_degenerate[1]) == len(drk)
    else:
        assert p.last_suspects[0] == [] and p.last_degenerate[0] == []


def test_overflowed_pieces_are_searched_again_with_the_final_minimum(monkeypatch):
    monkeypatch.setattr(_lib.Problem, "MAX_PER_CALL", {2: 1000, 3: 1000})
    probe = _FakeProblem(3, 5, 50 * 1000)
    mins = [(float(probe._data(b, b + 1000)[0]["nll"].min()) if len(probe._data(b, b + 1000)[0]["nll"]) else np.inf)
            for b in range(0, 50 * 1000, 1000)]
    last_drop = max(i for i in range(50) if mins[i] < min(mins[:i] + [np.inf]))      # the piece that sets the final minimum
    assert last_drop >= 3
    # two pieces searched with a hint above the final minimum lose suspects in their first pass
    p = _FakeProblem(3, 5, 50 * 1000, overflow_first_pass=(0, (last_drop - 1) * 1000))
    out = p.search(0, p.count, window=0.5)
    ranks, nll, srk, drk, evaluated = _brute(p, 0, p.count, 1000, 0.5)
    assert out["rank"] != [] and out["rank"] == ranks and np.array_equal(out["nll"], nll)    # piece order, second passes included
    assert list(p.last_suspects[0]) == srk and list(p.last_degenerate[0]) == drk
    assert p.suspect_reruns == 2 and len(p.calls) == 52
    gmin = float(out["nll"].min())
    assert p.calls[-1][2] == gmin and p.calls[-2][2] == gmin                    # second pass: the minimum of the whole range
    assert out["stats"]["evaluated"] == evaluated                                # (first-pass numbers of redone pieces are dropped)


def test_a_range_beyond_reach_is_refused_at_once():
    count = 25344449490209970329508701131116975                                  # n=3, m=70, bounds [0, 2] (the exact count)
    p = _FakeProblem(3, 70, count)
    t0 = time.time()
    with pytest.raises(_lib.ThetaError) as e:
        p.search(0, count)
    assert e.value.code == _lib.ERR_OVERFLOW and "candidate matrices" in str(e.value)
    assert time.time() - t0 < 1.0 and p.calls == []
    # the largest range one call walks is accepted (checked without walking it: the first piece is asked for)
    lim = _lib.Problem.MAX_PER_CALL[3] * _lib.Problem.MAX_PIECES

    class Stop(Exception):
        pass

    def first_piece(b, e, window, cap, hint):
        raise Stop()
    p._piece = first_piece
    with pytest.raises(Stop):
        p.search(5, 5 + lim)
    with pytest.raises(_lib.ThetaError):
        p.search(5, 6 + lim)


# ---- the recovery ladders (round-3 advice: none of them had a test) -------------------------------------------------------
class _DegOverflowProblem(_FakeProblem):
    """A piece longer than `limit` candidates holds more rank-deficient candidates than the device list."""
    limit = 1 << 21

    def _piece(self, b, e, window, cap, hint):
        if e - b > self.limit:
            self.calls.append((b, e, "overflow"))
            raise _lib.DegenerateOverflow(_lib.ERR_CAPACITY, "rank-deficient candidates did not fit the device list")
        return super()._piece(b, e, window, cap, hint)


def test_a_piece_with_too_many_rank_deficient_candidates_is_halved(monkeypatch):
    monkeypatch.setattr(_lib.Problem, "MAX_PER_CALL", {2: 1 << 23, 3: 1 << 23})
    p = _DegOverflowProblem(3, 5, (1 << 24) + 5)
    out = p.search(0, p.count, window=0.5)
    walked = [(b, e) for b, e, h in p.calls if h != "overflow"]
    assert walked[0][0] == 0 and walked[-1][1] == p.count
    assert all(x[1] == y[0] for x, y in zip(walked, walked[1:]))                 # the halves in rank order, nothing twice, nothing lost
    assert all(e - b <= _DegOverflowProblem.limit for b, e in walked)
    assert out["stats"]["evaluated"] == p.count
    # ... also in the second pass of a piece whose suspect list overflowed
    p = _DegOverflowProblem(3, 5, 3 << 21, overflow_first_pass=(1 << 21,))
    monkeypatch.setattr(_lib.Problem, "MAX_PER_CALL", {2: 1 << 21, 3: 1 << 21})
    hints = iter([900.0])

    orig = _FakeProblem._data

    def falling(self, b, e):                                                      # (the last piece lowers the minimum: the overflowed one is redone)
        res, sus, deg = orig(self, b, e)
        if b == 2 << 21 and len(res["nll"]):
            res["nll"] = res["nll"] - 50.0
        return res, sus, deg
    monkeypatch.setattr(_FakeProblem, "_data", falling)
    out = p.search(0, p.count, window=0.5)
    assert out["stats"]["evaluated"] == p.count


def test_overflow_kinds_are_exceptions_and_the_driver_narrows_the_window_for_each(monkeypatch):
    from theta_amd import search as S

    class Narrow:
        n, m, tau = 3, 4, 2
        suspect_reruns = 0

        def __init__(self, kind, fits_below):
            self.kind, self.fits_below, self.windows = kind, fits_below, []

        def search(self, begin, end, window=0.5):
            self.windows.append(window)
            if window > self.fits_below:
                if self.kind == "degenerate":
                    raise _lib.DegenerateOverflow(_lib.ERR_CAPACITY, "too many")
                raise _lib.ListOverflow("too many", self.kind)
            return {"rank": [], "nll": np.zeros(0), "mu": np.zeros((0, 3)), "C": np.zeros((0, 4, 2), np.uint8), "stats": {}}

    for kind in ("ties", "suspects", "degenerate"):
        rep = S.SearchReport()
        prob = Narrow(kind, 0.06)
        recs, stats = S.collect_finalists(prob, None, [1] * 4, [1] * 4, 1.0, 0, 10, report=rep)
        assert prob.windows == [0.5, 0.05] and rep.window == 0.05 and recs == []
        prob = Narrow(kind, 0.001)                                                # never fits: the third attempt's error comes out
        with pytest.raises(_lib.ListOverflow) as e:
            S.collect_finalists(prob, None, [1] * 4, [1] * 4, 1.0, 0, 10, report=S.SearchReport())
        assert prob.windows == [0.5, 0.05, 0.01] and e.value.kind == kind and e.value.code == _lib.ERR_CAPACITY
    # any other error is not retried
    class Broken(Narrow):
        def search(self, begin, end, window=0.5):
            self.windows.append(window)
            raise _lib.ThetaError(_lib.ERR_HIP, "device lost")
    prob = Broken("ties", 0)
    with pytest.raises(_lib.ThetaError):
        S.collect_finalists(prob, None, [1] * 4, [1] * 4, 1.0, 0, 10)
    assert prob.windows == [0.5]
