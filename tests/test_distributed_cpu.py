"""
The N>1 path on CPU: two (and three) processes call the library's OWN exchange -- theta_comm_create +
theta_exchange_finalists (theta_amd/csrc/comm.hip, the C entry points a sharded search uses) -- over its host transport
(the TCP star that also bootstraps RCCL), then replay the tie rule on the merged list.  The per-shard finalists are
synthetic in the first tests (no GPU in this container); on the GPU box the same entry points run over RCCL
(tests/test_gpu_zz_comm.py).  The last test runs the sharded DRIVER end to end over a stand-in device.
No torch anywhere in this path.
"""
import multiprocessing as mp
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard_records(rank, n, m):
    """Deterministic fake finalists of shard `rank` (ranks beyond 2^64 to exercise the 128-bit transport)."""
    rng = np.random.RandomState(100 + rank)
    recs = []
    base = (1 << 70) * (rank + 1)
    nlls = {0: [5000.0 + 0.0004 * rank, 5000.2, 5003.0 + rank], 1: [5000.0003, 4999.9998, 5000.45]}.get(rank, [6000.0, float("nan")])
    for j, v in enumerate(nlls):
        c = rng.randint(0, 5, (m, 2) if n == 3 else (m,)).astype(np.uint8)
        recs.append({"rank": base + 17 * j + rank, "c": c, "mu": rng.dirichlet(np.ones(n)), "nll": v,
                     "vals": rng.dirichlet(np.ones(m))})
    return recs


def _worker(rank, world, port, n, m, q):
    try:
        sys.path.insert(0, ROOT)
        import theta_amd
        from theta_amd.search import replay_ties
        comm = theta_amd.Comm(None, rank=rank, world=world, addr="127.0.0.1", port=port, transport="host")
        assert comm.info()["transport"] == "host" and comm.info()["world"] == world
        # the small collectives the drivers use
        assert comm.allreduce_min([3.0 + rank, -rank])[1] == -(world - 1)
        assert comm.allreduce_sum([1.0])[0] == world
        g = comm.allgather(np.array([rank, 10 * rank], np.int64))
        assert g.tolist() == [[k, 10 * k] for k in range(world)]
        comm.barrier()
        merged, gmin = comm.exchange_finalists(n, m, _shard_records(rank, n, m), 0.5)
        best = replay_ties(merged, n, 2, list(range(m)), first_duplicate=False)
        # an exchange in which NO rank has anything
        empty, gmin0 = comm.exchange_finalists(n, m, [], 0.5)
        assert empty == [] and gmin0 == float("inf")
        ncoll = comm.info()["collectives"]
        comm.close()
        q.put((rank, [(t["rank"], t["nll"], t["c"].tolist(), t["mu"].tolist(), t["vals"].tolist()) for t in merged],
               [(b[2], b[0].tolist()) for b in best], gmin, ncoll))
    except Exception as e:      # a failed rank must not leave the parent waiting for ever
        q.put((rank, "error: %r" % (e,), None, None, None))


def _run(world, n, m):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, m, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = {}
    for _ in range(world):
        item = q.get(timeout=120)
        out[item[0]] = item[1:]
    for p in procs:
        p.join(30)
    return out


def _same(a, b):
    return a == b or (a != a and b != b)


@pytest.mark.parametrize("n,m", [(3, 7), (2, 5)])
def test_exchange_and_replay_two_ranks(n, m):
    world = 2
    out = _run(world, n, m)
    assert set(out.keys()) == {0, 1}
    assert not isinstance(out[0][0], str) and not isinstance(out[1][0], str), (out[0][0], out[1][0])
    assert out[0][:3] == out[1][:3]                            # every rank ends with the same answer
    merged, best, gmin, ncoll = out[0]
    assert gmin == 4999.9998
    # expected: everything within the collection window of the global minimum (4999.9998), from both shards
    want = []
    for rk in range(world):
        for t in _shard_records(rk, n, m):
            if t["nll"] <= 4999.9998 + 0.5:
                want.append(t)
    assert len(merged) == len(want) == 5
    for got, t in zip(merged, want):
        assert got[0] == t["rank"] and got[1] == t["nll"]
        assert got[2] == t["c"].tolist()
        assert np.allclose(got[3], t["mu"]) and np.allclose(got[4], t["vals"])
    # sequential tie rule in rank order: shard 0's 5000.0 comes first, 5000.2 is dropped by the gap cut,
    # shard 1's 5000.0003 and 4999.9998 are within 1e-3 of the running minimum and are appended
    assert [b[0] for b in best] == [5000.0, 5000.0003, 4999.9998]


def test_exchange_three_ranks_keeps_nan_records():
    """A shard whose finalists are all beyond the window still contributes its NaN-likelihood record (isClose(NaN))."""
    n, m, world = 3, 6, 3
    out = _run(world, n, m)
    assert set(out.keys()) == {0, 1, 2}
    for r in range(world):
        assert not isinstance(out[r][0], str), out[r][0]
    m0, b0 = out[0][0], out[0][1]
    for r in (1, 2):
        assert len(out[r][0]) == len(m0)
        for x, y in zip(out[r][0], m0):
            assert x[0] == y[0] and _same(x[1], y[1])
    assert len(m0) == 6 and sum(1 for t in m0 if t[1] != t[1]) == 1
    assert m0[-1][1] != m0[-1][1]                              # rank order: shard 2's records come last
    # the replay appends the NaN record to the list it finds (RunTHetA.py:198-201 with Misc.py:44-46)
    nl = [b[0] for b in b0]
    assert nl[:3] == [5000.0, 5000.0003, 4999.9998] and len(nl) == 4 and nl[3] != nl[3]


def _uneven_worker(rank, world, port, counts, q):
    try:
        sys.path.insert(0, ROOT)
        import theta_amd
        n, m = 3, 5
        comm = theta_amd.Comm(None, rank=rank, world=world, addr="127.0.0.1", port=port, transport="host")
        rng = np.random.RandomState(7 + rank)
        recs = [{"rank": (rank << 40) + j, "c": rng.randint(0, 3, (m, 2)).astype(np.uint8), "mu": rng.dirichlet(np.ones(n)),
                 "nll": float("nan") if j % 2 else 100.0 + 1e-4 * j, "vals": rng.dirichlet(np.ones(m))} for j in range(counts[rank])]
        merged, gmin = comm.exchange_finalists(n, m, recs, 0.5)
        merged2, _ = comm.exchange_finalists(n, m, recs[: counts[rank] // 3], 0.5)      # and again, other counts
        comm.close()
        q.put((rank, [t["rank"] for t in merged], len(merged2), gmin, None))
    except Exception as e:
        q.put((rank, "error: %r" % (e,), None, None, None))


@pytest.mark.parametrize("counts", [(100, 0), (0, 300, 7)])
def test_exchange_with_uneven_shards_is_one_collective_decision(counts):
    """Round-2 advice (high): each rank sized its output room from its OWN record count, so with 100 records on one shard and
    none on the other the ranks disagreed on whether to come back with more room -- one raised, the other hung in the next
    collective.  The room is now agreed on by an all-reduce of the counts before the exchange."""
    world = len(counts)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_uneven_worker, args=(r, world, port, counts, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = {}
    for _ in range(world):
        item = q.get(timeout=120)
        out[item[0]] = item[1:]
    for p in procs:
        p.join(30)
    for r in range(world):
        assert not isinstance(out[r][0], str), out[r][0]
        assert out[r][0] == out[0][0] and out[r][1] == out[0][1]
    assert len(out[0][0]) == sum(counts)                   # all within the window or NaN: every record of every shard, in rank order
    assert out[0][0] == sorted(out[0][0])
    assert out[0][1] == sum(c // 3 for c in counts)


# ---------------------------------------------------------------------------------------------------
# the sharded DRIVER end to end: do_optimization_distributed over the stand-in device (tests/standin_device.py: every
# candidate through the CPU oracle) and the library's host transport -- rank-range sharding, the probe-minimum all-reduce,
# theta_exchange_finalists, the tie replay -- against the oracle's port of the reference's single-process driver
# ---------------------------------------------------------------------------------------------------
def _driver_worker(rank, world, port, inst, q):
    try:
        for pth in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
            sys.path.insert(0, pth)
        import warnings
        warnings.simplefilter("ignore")
        import theta_amd
        import campaign as cp
        import standin_device as sd
        from theta_amd import _lib, search as S
        ctx = sd.StandinContext()
        made = []

        def make(c, *a, **k):
            made.append(sd.StandinProblem(c, *a, **k))
            return made[-1]
        _lib.Problem = make
        comm = theta_amd.Comm(None, rank=rank, world=world, addr="127.0.0.1", port=port, transport="host")
        best = S.do_optimization_distributed(inst["n"], inst["m"], inst["k"], inst["tau"], inst["lb"], inst["ub"], inst["r"], inst["rN"],
                                             inst["mx"], inst["order"], comm, ctx=ctx)
        ncoll = comm.info()["collectives"]
        comm.close()
        lo, hi = made[-1].count * rank // world, made[-1].count * (rank + 1) // world
        searched = [(b, e) for b, e, _ in made[-1].search_calls]
        q.put((rank, cp.best_to_plain(best), (lo, hi), searched, ncoll))
    except BaseException as e:
        q.put((rank, "error: %r" % (e,), None, None, None))


@pytest.mark.parametrize("n,seed,world", [(2, 9512, 2), (3, 10044, 2), (3, 10010, 3), (3, 10044, 8)])
def test_sharded_driver_over_the_standin_device_equals_the_reference_driver(n, seed, world):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import warnings
    import campaign
    import theta_oracle as orc
    inst = campaign.instance(seed, n, "toy")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref, cnt = orc.search_single(n, inst["m"], inst["tau"], list(inst["lb"]), list(inst["ub"]), inst["r"], inst["rN"], inst["mx"],
                                     inst["order"])
    ref = campaign.best_to_plain(ref)
    assert ref and 40 <= cnt <= 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_driver_worker, args=(rk, world, port, inst, q)) for rk in range(world)]
    for p in procs:
        p.start()
    out = {}
    for _ in range(world):
        item = q.get(timeout=600)
        out[item[0]] = item[1:]
    for p in procs:
        p.join(30)
    for rk in range(world):
        best, shard, searched, ncoll = out[rk]
        assert not isinstance(best, str), best
        assert campaign.compare_best(best, ref) == "", (rk, seed)                  # every rank returns the reference's list
        assert searched and searched[0][0] == shard[0] and searched[-1][1] == shard[1]      # and searched only its own ranks
        assert ncoll >= 2                                                            # the hint all-reduce and the exchange


def _mix_driver_worker(rank, world, port, inst, q, fail_rank=None):
    try:
        for pth in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
            sys.path.insert(0, pth)
        import warnings
        warnings.simplefilter("ignore")
        import theta_amd
        import campaign as cp
        import standin_device as sd
        from theta_amd import _lib, search as S
        S.BNB_MIN_CANDIDATES = 0                 # every n=3 space goes to the mixture-space search, as a 1e27-matrix one would
        ctx = sd.StandinContext()
        made = []

        def make(c, *a, **k):
            made.append(sd.StandinProblem(c, *a, **k))
            made[-1].standin_mix = True
            made[-1].mix_fail_rank = fail_rank
            return made[-1]
        _lib.Problem = make
        comm = theta_amd.Comm(None, rank=rank, world=world, addr="127.0.0.1", port=port, transport="host")
        best = S.do_optimization_distributed(inst["n"], inst["m"], inst["k"], inst["tau"], inst["lb"], inst["ub"], inst["r"], inst["rN"],
                                             inst["mx"], inst["order"], comm, ctx=ctx)
        ncoll = comm.info()["collectives"]
        comm.close()
        q.put((rank, cp.best_to_plain(best), made[-1].mix_calls, made[-1].mix_listed, ncoll, S.last_report.mix))
    except BaseException as e:
        import traceback
        q.put((rank, "error: %r %s" % (e, traceback.format_exc()[-600:]), None, None, None, None))


@pytest.mark.parametrize("seed,world", [(10044, 2), (10010, 8)])
def test_sharded_mixture_space_search_over_the_standin_device(seed, world):
    """The whole-space search of a space no walk finishes, on several ranks (round 6: search._search_local deals the boxes out,
    mix_records agrees on the attainable NLL by all-reduce after every step that can lower it, the records meet in
    theta_exchange_finalists and merge_mix_records restores the enumeration order): over the host transport with the stand-in
    device, whose mix_search deals the matrices within a threshold out over the ranks.  Every rank goes through the same
    collectives and returns the reference's list (its NaN entries apart: no bound reaches those); the ranks' lists are disjoint,
    their thresholds equal."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import warnings
    import campaign
    import theta_oracle as orc
    inst = campaign.instance(seed, 3, "toy")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref, cnt = orc.search_single(3, inst["m"], inst["tau"], list(inst["lb"]), list(inst["ub"]), inst["r"], inst["rN"], inst["mx"],
                                     inst["order"])
    ref = [b for b in campaign.best_to_plain(ref) if b[2] == b[2]]
    assert ref and 40 <= cnt <= 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mix_driver_worker, args=(rk, world, port, inst, q)) for rk in range(world)]
    for p in procs:
        p.start()
    out = {}
    for _ in range(world):
        item = q.get(timeout=900)
        out[item[0]] = item[1:]
    for p in procs:
        p.join(30)
    thresholds, seen, ncolls = set(), [], set()
    for rk in range(world):
        best, calls, listed, ncoll, mix = out[rk]
        assert not isinstance(best, str), best
        got = [b for b in best if b[2] == b[2]]
        assert campaign.compare_best(got, ref) == "", (rk, seed)
        assert mix["shard"] == [rk, world] and all(c[2] == (rk, world) for c in calls)
        finals = [c for c in calls if c[0] == "list"]
        assert len(finals) >= 1
        thresholds.add(round(finals[-1][1], 9))
        seen.append(set(listed[-1]))
        ncolls.add(ncoll)
    assert len(thresholds) == 1, thresholds                  # one attainable NLL for all ranks
    assert len(ncolls) == 1, ncolls                          # ... reached through the same collectives
    for a in range(world):
        for b in range(a + 1, world):
            assert not (seen[a] & seen[b])                   # disjoint shares
    assert sum(len(x) for x in seen) >= len(ref)


def test_a_rank_whose_share_of_the_boxes_is_too_much_takes_all_ranks_to_the_walks():
    """Whether the mixture-space search gives up depends on a rank's OWN share of the boxes (round 6): one rank of three gives up in
    its thresholded walks -- every rank must leave that search at the same collective and fall back to the rank walk together
    (else the ranks' collectives pair up wrongly, or hang).  Every rank returns the reference's list."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import warnings
    import campaign
    import theta_oracle as orc
    inst = campaign.instance(10010, 3, "toy")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref, cnt = orc.search_single(3, inst["m"], inst["tau"], list(inst["lb"]), list(inst["ub"]), inst["r"], inst["rN"], inst["mx"],
                                     inst["order"])
    ref = campaign.best_to_plain(ref)
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mix_driver_worker, args=(rk, world, port, inst, q, 1)) for rk in range(world)]
    for p in procs:
        p.start()
    out = {}
    for _ in range(world):
        item = q.get(timeout=900)
        out[item[0]] = item[1:]
    for p in procs:
        p.join(30)
    ncolls = set()
    for rk in range(world):
        best, calls, listed, ncoll, mix = out[rk]
        assert not isinstance(best, str), best
        assert campaign.compare_best(best, ref) == "", rk                       # (the walk's list: NaN entries included)
        assert mix is not None and "gave_up" in mix, mix
        ncolls.add(ncoll)
    assert len(ncolls) == 1, ncolls


def _failing_worker(rank, world, port, inst, fail_rank, fail_in_probe, q):
    try:
        for pth in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
            sys.path.insert(0, pth)
        import warnings
        warnings.simplefilter("ignore")
        import theta_amd
        import standin_device as sd
        from theta_amd import _lib, search as S

        class Failing(sd.StandinProblem):
            def _probe(self, begin, end):
                if rank == fail_rank and fail_in_probe:
                    raise _lib.ThetaError(_lib.ERR_CAPACITY, "device list full (injected)")
                return super()._probe(begin, end)

            def search(self, *a, **k):
                if rank == fail_rank:
                    raise _lib.ThetaError(_lib.ERR_CAPACITY, "device list full (injected)")
                return super().search(*a, **k)

        _lib.Problem = Failing
        comm = theta_amd.Comm(None, rank=rank, world=world, addr="127.0.0.1", port=port, transport="host")
        try:
            S.do_optimization_distributed(inst["n"], inst["m"], inst["k"], inst["tau"], inst["lb"], inst["ub"], inst["r"], inst["rN"],
                                          inst["mx"], inst["order"], comm, ctx=sd.StandinContext())
            q.put((rank, "returned"))
        except _lib.ThetaError as e:
            q.put((rank, "ThetaError %d" % e.code))
        comm.barrier()                                   # the communicator is still in step on every rank
        comm.close()
    except BaseException as e:
        q.put((rank, "error: %r" % (e,)))


@pytest.mark.parametrize("fail_in_probe", [False, True])
def test_a_failing_shard_takes_every_rank_out_together(fail_in_probe):
    """Round-2 advice (medium): a data-dependent failure of ONE shard (ERR_CAPACITY) used to leave the other ranks waiting in
    theta_exchange_finalists for ever.  Now every rank raises the same status after an all-reduce of a flag."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import campaign
    from theta_amd import _lib
    inst = campaign.instance(10044, 3, "toy")
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(rk, world, port, inst, 1, fail_in_probe, q)) for rk in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(30)
    assert out == {rk: "ThetaError %d" % _lib.ERR_CAPACITY for rk in range(world)}, out


def test_a_rank_that_never_connects_is_reported_not_waited_for(monkeypatch):
    """theta_comm_create: the rendezvous has a time-out (THETA_COMM_TIMEOUT_S) and says who is missing."""
    import time
    import theta_amd
    monkeypatch.setenv("THETA_COMM_TIMEOUT_S", "2")
    t = time.time()
    with pytest.raises(theta_amd.ThetaError) as e:
        theta_amd.Comm(None, rank=0, world=2, addr="127.0.0.1", port=_free_port(), transport="host")
    assert "only 1 of 2 ranks" in str(e.value) and time.time() - t < 30
    t = time.time()
    with pytest.raises(theta_amd.ThetaError) as e:
        theta_amd.Comm(None, rank=1, world=2, addr="127.0.0.1", port=_free_port(), transport="host")
    assert "cannot reach rank 0" in str(e.value) and time.time() - t < 30


# ---------------------------------------------------------------------------------------------------
# the reference's OWN parallel entry: do_optimization(..., max_processes) (RunTHetA.py:124-171) starts one worker process per
# further GPU itself (theta_amd/shard_worker.py) -- here over the stand-in device and the host transport
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,seed,procs", [(3, 10044, 4), (2, 9512, 3), (3, 10010, 2)])
def test_do_optimization_with_max_processes_spawns_the_shards_itself(monkeypatch, n, seed, procs):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import warnings
    import campaign
    import standin_device as sd
    import theta_oracle as orc
    from theta_amd import _lib, search as S
    inst = campaign.instance(seed, n, "toy")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref, cnt = orc.search_single(n, inst["m"], inst["tau"], list(inst["lb"]), list(inst["ub"]), inst["r"], inst["rN"], inst["mx"],
                                     inst["order"])
    ref = campaign.best_to_plain(ref)
    monkeypatch.setattr(_lib, "Problem", sd.StandinProblem)
    monkeypatch.setattr(S, "WORKER_INIT", "standin_device:worker_context")
    monkeypatch.setenv("THETA_NGPU", str(procs))            # (no GPU to count here: the number of ranks is given)
    best = S.do_optimization(inst["n"], inst["m"], inst["k"], inst["tau"], list(inst["lb"]), list(inst["ub"]), inst["r"], inst["rN"],
                             inst["mx"], inst["order"], procs)
    assert campaign.compare_best(campaign.best_to_plain(best), ref) == ""
    assert S.last_report.gpus == procs and S.last_report.transport == "host"
    # ... and equals the single-process driver's list, entry by entry
    single = S.do_optimization_single(inst["n"], inst["m"], inst["k"], inst["tau"], list(inst["lb"]), list(inst["ub"]), inst["r"],
                                      inst["rN"], inst["mx"], inst["order"], _ctx=sd.StandinContext())
    assert campaign.compare_best(campaign.best_to_plain(best), campaign.best_to_plain(single)) == ""


def test_do_optimization_caps_the_gpus_by_the_size_of_the_space(monkeypatch):
    from theta_amd import _lib, search as S
    monkeypatch.delenv("THETA_NGPU", raising=False)
    monkeypatch.setattr(_lib, "device_count", lambda: 8)
    assert S.gpus_for(8) == 8 and S.gpus_for(3) == 3 and S.gpus_for(1) == 1 and S.gpus_for(64) == 8
    assert S.gpus_for(8, count=10 ** 6) == 1                       # a small space is not worth a second process
    assert S.gpus_for(8, count=3 * S.MIN_CANDIDATES_PER_GPU) == 3
    assert S.gpus_for(8, count=1 << 100) == 8
    monkeypatch.setenv("THETA_NGPU", "2")
    assert S.gpus_for(8, count=10) == 2
    assert S.gpus_for(1, count=1 << 100) == 1                       # (a variable left over from a test never shards a max_processes = 1 call)


def test_a_failing_worker_is_reported_by_do_optimization(monkeypatch):
    """A worker that cannot start (its init hook raises) must end the call with an error, not leave rank 0 in the rendezvous."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import campaign
    import standin_device as sd
    from theta_amd import _lib, search as S
    inst = campaign.instance(10044, 3, "toy")
    monkeypatch.setattr(_lib, "Problem", sd.StandinProblem)
    monkeypatch.setattr(S, "WORKER_INIT", "standin_device:worker_context")
    monkeypatch.setenv("THETA_NGPU", "2")
    monkeypatch.setenv("THETA_COMM_TIMEOUT_S", "5")
    real = S._spawn_shards

    def broken(world, args, transport, ndev):
        monkeypatch.setattr(S, "WORKER_INIT", "standin_device:no_such_function")
        try:
            return real(world, args, transport, ndev)
        finally:
            monkeypatch.setattr(S, "WORKER_INIT", "standin_device:worker_context")
    monkeypatch.setattr(S, "_spawn_shards", broken)
    with pytest.raises(_lib.ThetaError):
        S.do_optimization(inst["n"], inst["m"], inst["k"], inst["tau"], list(inst["lb"]), list(inst["ub"]), inst["r"], inst["rN"],
                          inst["mx"], inst["order"], 2)


def test_bench_with_eight_ranks_over_the_host_transport():
    """bench.py launched as the driver launches it for N = 8 -- eight processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* --
    on this GPU-less machine: every rank on the CPU stand-in device (THETA_BENCH_INIT, a space the oracle can walk through
    THETA_BENCH_SHAPE), the collectives over the library's host transport.  The line must carry n_gpus = 8, the candidates of all
    eight ranks in `value`, the transport and the ranks' devices at its top level; and with the transport left at its default
    (RCCL) the run must FAIL rather than downgrade silently (round-4 verdict, Next 5)."""
    import json
    import subprocess
    import sys as _sys
    port = _free_port()
    here = os.path.dirname(os.path.abspath(__file__))
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="8", THETA_BENCH_TRANSPORT="host",
                THETA_BENCH_INIT="standin_device:worker_context", THETA_BENCH_SHAPE="6,2", THETA_COMM_TIMEOUT_S="120",
                PYTHONPATH=os.pathsep.join([here, os.path.join(ROOT, "oracle"), ROOT, os.environ.get("PYTHONPATH", "")]))
    args = [_sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--batch", "12",
            "--leg", "full_solve_f64_tight_certified"]
    procs = [subprocess.Popen(args, env=dict(base, RANK=str(rk), LOCAL_RANK=str(rk)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for rk in range(8)]
    try:
        outs = [p.communicate(timeout=500) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert all(p.returncode == 0 for p in procs), [o[1][-600:] for o in outs]
    assert all(o[0].strip() == "" for o in outs[1:])                       # only rank 0 prints
    line = json.loads(outs[0][0].strip().splitlines()[-1])
    assert line["n_gpus"] == 8 and line["steps"] == 2 and line["scaling"] == "weak"
    assert line["transport"] == "host" and line["rccl_version"] == 0 and len(line["rank_devices"]) == 8
    assert line["candidates_per_rank"] == [24.0] * 8                        # 2 timed steps x 12 candidates on every rank
    # round 6 (verdict, Next 9): rank g runs on device LOCAL_RANK = g, searches inside its own shard, and no two shards overlap
    assert line["rank_devices"] == list(range(8))
    rr = line["rank_ranges"]
    assert len(rr) == 8
    for g in range(8):
        a, b = rr[g]["searched"]
        c, d = rr[g]["shard"]
        assert abs(c - g / 8) < 1e-3 and abs(d - (g + 1) / 8) < 1e-3 and c <= a < b <= d, (g, rr[g])
        if g:
            assert rr[g - 1]["shard"][1] <= c + 1e-12 and rr[g - 1]["searched"][1] <= a, (g, rr[g - 1], rr[g])
    assert abs(line["value"] * line["ms_per_step"] * 1e-3 * 2 - 8 * 24) < 1e-6 * 8 * 24      # value = all ranks' candidates / the slowest rank's time
    assert line["comm"]["world"] == 8 and line["comm"]["collectives"] >= 4
    # the default transport (RCCL) cannot be set up here (no GPU, no context): the run must end non-zero, not downgrade
    port2 = _free_port()
    base2 = dict(base, MASTER_PORT=str(port2), WORLD_SIZE="2")
    base2.pop("THETA_BENCH_TRANSPORT")
    procs = [subprocess.Popen(args[:2] + ["--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "8"], env=dict(base2, RANK=str(rk), LOCAL_RANK=str(rk)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for rk in range(2)]
    try:
        outs = [p.communicate(timeout=300) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert all(p.returncode != 0 for p in procs), [o[1][-300:] for o in outs]
    assert all("not falling back" in o[1] for o in outs), [o[1][-300:] for o in outs]
    assert all(o[0].strip() == "" for o in outs)                              # and no JSON line


def test_launch_node_sets_what_the_drivers_launcher_sets(tmp_path):
    """tools/launch_node.sh (one process per GPU without torchrun) against `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port P` -- the command the driver runs: every variable bench.py reads
    (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT) has the same value rank by rank under both."""
    import json
    import subprocess
    import sys as _sys
    script = tmp_path / "print_env.py"
    script.write_text("import json, os, sys\n"
                      "keys = ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')\n"
                      "open(os.path.join(sys.argv[1], 'rank%s.json' % os.environ['RANK']), 'w').write(json.dumps({k: os.environ.get(k) for k in keys}))\n")
    got = {}
    for name, cmd in (("launch_node", ["bash", os.path.join(ROOT, "tools", "launch_node.sh"), "3", str(script)]),
                      ("torchrun", [_sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
                                    "--master-port", "%d", str(script)])):
        out = tmp_path / name
        out.mkdir()
        port = _free_port()
        cmd = [c % port if c == "%d" else c for c in cmd] + [str(out)]
        env = dict(os.environ, MASTER_PORT=str(port), MASTER_ADDR="127.0.0.1")
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, (name, res.stderr[-800:])
        got[name] = {rk: json.loads((out / ("rank%d.json" % rk)).read_text()) for rk in range(3)}
        for rk in range(3):
            assert got[name][rk] == {"RANK": str(rk), "LOCAL_RANK": str(rk), "WORLD_SIZE": "3", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)}, (name, rk)
