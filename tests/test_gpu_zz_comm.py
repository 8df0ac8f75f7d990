"""
The library's communicator on the GPU box (run with `-m gpu`): RCCL with world = 1 on this one-GPU box, and two processes
sharing the GPU over the host transport.  The file sorts last on purpose: the oracle-side worker pools of the other GPU
tests are forked from the pytest process, and nothing is forked from a process that has initialised RCCL.
"""
import multiprocessing as mp
import os
import socket
import sys

import numpy as np
import pytest

import campaign
from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import theta_amd
    return theta_amd.default_context()


def _gpu_best(inst):
    from theta_amd.search import do_optimization_single
    try:
        best = do_optimization_single(inst["n"], inst["m"], inst["k"], inst["tau"], list(inst["lb"]), list(inst["ub"]), inst["r"],
                                      inst["rN"], inst["mx"], inst["order"])
    except SystemExit:
        best = []
    return campaign.best_to_plain(best)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard_worker(rank, world, port, inst, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import theta_amd
        from theta_amd import search as S
        import campaign as cp
        c = theta_amd.Context(0)
        comm = theta_amd.Comm(c, rank=rank, world=world, addr="127.0.0.1", port=port, transport="host")
        best = S.do_optimization_distributed(inst["n"], inst["m"], inst["k"], inst["tau"], inst["lb"], inst["ub"], inst["r"], inst["rN"],
                                             inst["mx"], inst["order"], comm, ctx=c)
        comm.close()
        q.put((rank, cp.best_to_plain(best)))
    except BaseException as e:
        q.put((rank, "error: %r" % (e,)))


def test_two_processes_share_the_gpu_and_exchange_through_the_library(ctx):
    """do_optimization_distributed with world = 2 on this box's single GPU (host transport: RCCL refuses two ranks on one
    device): both ranks return the single-GPU `best`, n=2 and n=3."""
    mpc = mp.get_context("spawn")
    done = 0
    for n, seeds in ((2, range(9500, 9600)), (3, range(9600, 9800))):
        got = 0
        for seed in seeds:
            inst = campaign.instance(seed, n, "mid" if seed % 2 else "toy")
            cnt = campaign.count_candidates(inst)
            if not (300 <= cnt <= 100000):
                continue
            single = _gpu_best(inst)
            q = mpc.Queue()
            port = _free_port()
            procs = [mpc.Process(target=_shard_worker, args=(rk, 2, port, inst, q)) for rk in range(2)]
            for pr in procs:
                pr.start()
            out = dict(q.get(timeout=300) for _ in range(2))
            for pr in procs:
                pr.join(60)
            for rk in range(2):
                assert not isinstance(out[rk], str), out[rk]
                assert campaign.compare_best(out[rk], single) == "", (n, seed, rk)
            got += 1
            done += 1
            if got >= 2:
                break
    assert done == 4


def test_rccl_communicator_world_of_one(ctx):
    """ncclCommInitRank / ncclAllReduce / ncclAllGather through the library (librccl.so is dlopened here): one rank, this GPU."""
    import theta_amd
    comm = theta_amd.Comm(ctx, rank=0, world=1, transport="rccl")
    info = comm.info()
    assert info["transport"] == "rccl" and info["rccl_version"] > 20000
    assert comm.allreduce_min([3.5, -1.0]).tolist() == [3.5, -1.0]
    assert comm.allreduce_sum([2.0]).tolist() == [2.0]
    assert comm.allgather(np.arange(5, dtype=np.int32)).tolist() == [[0, 1, 2, 3, 4]]
    comm.barrier()
    recs = [{"rank": (1 << 70) + 3, "c": np.ones((6, 2), np.uint8), "mu": np.array([.2, .3, .5]), "nll": 10.0, "vals": np.ones(6)},
            {"rank": 5, "c": np.zeros((6, 2), np.uint8), "mu": np.array([.1, .1, .8]), "nll": float("nan"), "vals": np.ones(6)},
            {"rank": 9, "c": np.zeros((6, 2), np.uint8), "mu": np.array([.1, .1, .8]), "nll": 11.0, "vals": np.ones(6)}]
    merged, gmin = comm.exchange_finalists(3, 6, recs, 0.5)
    assert gmin == 10.0 and [t["rank"] for t in merged] == [5, (1 << 70) + 3]
    assert merged[0]["nll"] != merged[0]["nll"] and merged[1]["c"].tolist() == [[1, 1]] * 6
    assert comm.info()["collectives"] >= 6
    comm.close()


def test_bench_with_two_ranks_on_one_gpu():
    """
    bench.py itself as the driver launches it for N = 2 -- two processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their
    environment (what `python -m torch.distributed.run` provides; nothing imports torch) -- sharing this box's single GPU over the
    library's host transport (THETA_BENCH_NDEV=1, THETA_BENCH_TRANSPORT=host; on an 8-GPU node the same code runs one rank per
    GPU over RCCL).  Checks the JSON line: n_gpus, the hint all-reduce + the exchange happened, every rank's candidates counted,
    and the two time-shared ranks together are not slower than half of one rank alone.
    """
    import json
    import subprocess
    port = _free_port()
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", THETA_BENCH_NDEV="1",
                THETA_BENCH_TRANSPORT="host", THETA_COMM_TIMEOUT_S="120")
    args = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", str(1 << 29)]
    procs = [subprocess.Popen(args, env=dict(base, RANK=str(rk), LOCAL_RANK=str(rk)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for rk in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-800:] for o in outs]
    assert outs[1][0].strip() == ""                                   # only rank 0 prints
    line = json.loads(outs[0][0].strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak" and line["dtype"] == "f64"
    assert line["comm"]["world"] == 2 and line["comm"]["collectives"] >= 4      # hint all-reduce, exchange (all-reduces + all-gather), barriers
    assert line["cpu_baseline"] is None and line["roofline"]["traffic"] is None
    assert line["config"]["leg"] == "full_solve_f64_tight_certified"          # the default headline: the tolerance-meeting leg
    leg = line["roofline"]["legs"][line["config"]["leg"]]
    assert leg["launches"] == 3 and leg["candidates_per_launch"] == 1 << 29
    assert line["transport"] == "host" and line["rank_devices"] == [0, 0] and len(line["candidates_per_rank"]) == 2
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--batch", str(1 << 29),
                          "--no-cpu-baseline", "--no-legs", "--no-traffic", "--no-extras"], capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-800:]
    single = json.loads(one.stdout.strip().splitlines()[-1])
    assert single["n_gpus"] == 1
    # whole-job rate of two ranks time-sharing ONE GPU: about the single-rank rate (not twice, the GPU is the same); within 2x
    assert 0.5 * single["value"] <= line["value"] <= 2.0 * single["value"], (line["value"], single["value"])


def test_bench_falls_back_to_the_host_transport_when_rccl_refuses_the_communicator():
    """
    Two ranks on this box's ONE GPU with the RCCL transport (the driver's default): rank 0's ncclUniqueId travels over the
    library's bootstrap, both ranks enter ncclCommInitRank, RCCL's own bootstrap gathers the peers -- and refuses ("Duplicate GPU
    detected", ncclInvalidUsage; the furthest an RCCL communicator of world > 1 gets on a one-GPU box).  bench.py then carries its
    two collectives over the host transport and says so on the line (`transport`, top level) -- but only where the caller allows
    it (THETA_BENCH_TRANSPORT=rccl_or_host).  Under the driver's launcher (the variable unset) the same situation is an ERROR: both
    ranks exit non-zero and no JSON line is printed, so a scaling record can never be green on a downgraded transport.
    """
    import json
    import subprocess
    port = _free_port()
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", THETA_BENCH_NDEV="1", THETA_COMM_TIMEOUT_S="120",
                THETA_BENCH_TRANSPORT="rccl_or_host")
    args = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", str(1 << 28)]
    procs = [subprocess.Popen(args, env=dict(base, RANK=str(rk), LOCAL_RANK=str(rk)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for rk in range(2)]
    try:
        outs = [p.communicate(timeout=500) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert all(p.returncode == 0 for p in procs), [o[1][-800:] for o in outs]
    assert all("falling back to the host transport" in o[1] for o in outs), [o[1][-400:] for o in outs]
    line = json.loads(outs[0][0].strip().splitlines()[-1])       # (RCCL's version banner, also on stdout, comes before: bench.py flushes it first)
    assert line["n_gpus"] == 2 and line["comm"]["world"] == 2 and line["comm"]["transport"] == "host" and line["comm"]["collectives"] >= 4
    assert line["transport"] == "host"
    # the driver's launcher: no THETA_BENCH_TRANSPORT -- RCCL or nothing
    strict = dict(base, MASTER_PORT=str(_free_port()))
    strict.pop("THETA_BENCH_TRANSPORT")
    procs = [subprocess.Popen(args, env=dict(strict, RANK=str(rk), LOCAL_RANK=str(rk)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for rk in range(2)]
    try:
        outs = [p.communicate(timeout=500) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert all(p.returncode != 0 for p in procs), [o[1][-400:] for o in outs]
    assert all("not falling back" in o[1] for o in outs), [o[1][-400:] for o in outs]
    assert not any(l.startswith("{") for o in outs for l in o[0].splitlines())


def test_do_optimization_with_max_processes_shards_over_worker_processes(ctx, monkeypatch):
    """The reference's parallel entry (RunTHetA.py:124-171): do_optimization(..., max_processes) starts its own worker
    process per further GPU.  On this one-GPU box THETA_NGPU=2 / 3 makes the ranks share the device (host transport); every
    list equals do_optimization_single's, n=2 and n=3."""
    from theta_amd import search as S
    done = 0
    for n, seeds, world in ((2, range(9500, 9600), 2), (3, range(9600, 9800), 3)):
        for seed in seeds:
            inst = campaign.instance(seed, n, "mid" if seed % 2 else "toy")
            cnt = campaign.count_candidates(inst)
            if not (300 <= cnt <= 100000):
                continue
            single = _gpu_best(inst)
            monkeypatch.setenv("THETA_NGPU", str(world))
            best = S.do_optimization(inst["n"], inst["m"], inst["k"], inst["tau"], list(inst["lb"]), list(inst["ub"]), inst["r"],
                                     inst["rN"], inst["mx"], inst["order"], 8)
            monkeypatch.delenv("THETA_NGPU")
            assert S.last_report.gpus == world and S.last_report.transport == "host"
            assert len(S.last_report.shard_kernel_ms) == world
            assert campaign.compare_best(campaign.best_to_plain(best), single) == "", (n, seed)
            done += 1
            break
    assert done == 2
    # without the override: one GPU here, so max_processes = 8 runs on it alone (and a small space would anyway)
    assert S.gpus_for(8) == 1
    best = S.do_optimization(inst["n"], inst["m"], inst["k"], inst["tau"], list(inst["lb"]), list(inst["ub"]), inst["r"],
                             inst["rN"], inst["mx"], inst["order"], 8)
    assert campaign.compare_best(campaign.best_to_plain(best), single) == "" and S.last_report.gpus == 1


def test_a_space_too_large_to_walk_is_searched_by_every_rank_of_a_sharded_run(ctx, monkeypatch):
    """BASELINE config 3 (m = 50, n = 3, k = 4: 4e27 matrices) through do_optimization(..., max_processes) with two ranks sharing this
    box's GPU: every rank takes the dive whole, then walks ITS share of the boxes of the mixture-space branch and bound (round 6: dealt
    out by position where they become small; the attainable NLL agreed on by the library's all-reduce after every step that can lower
    it), every rank contributes its records to the ONE exchange, and every rank holds the `best` of do_optimization_single."""
    import bench
    from theta_amd import search as S
    r, rN, order = bench.synth(seed=7, m=50, n=3, k=4)
    single = campaign.best_to_plain(S.do_optimization_single(3, 50, 4, 2, [0] * 50, [4] * 50, r, rN, 1.0, order, False, False))
    assert S.last_report.mix is not None and S.last_report.candidates > 1e27
    monkeypatch.setenv("THETA_NGPU", "2")
    best = S.do_optimization(3, 50, 4, 2, [0] * 50, [4] * 50, r, rN, 1.0, order, 8)
    assert S.last_report.gpus == 2 and S.last_report.mix is not None and S.last_report.mix["shard"] == [0, 2]
    assert campaign.compare_best(campaign.best_to_plain(best), single) == ""


def test_cli_num_processes_reaches_the_sharded_driver(tmp_path, monkeypatch):
    """`RunTHetA <file> -n 2 --NUM_PROCESSES 2` (the reference's flag) under THETA_NGPU=2: same result file as one process."""
    import subprocess
    src = os.path.join(ROOT, "tests", "golden", "cli", "Example.intervals")
    outs = []
    for tag, extra, env in (("one", [], {}), ("two", ["--NUM_PROCESSES", "2"], {"THETA_NGPU": "2"})):
        d = tmp_path / tag
        d.mkdir()
        p = subprocess.run([sys.executable, "-m", "theta_amd.RunTHetA", src, "-n", "2", "-k", "3", "-d", str(d), "-p", "ex"] + extra,
                           cwd=ROOT, env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout[-600:] + p.stderr[-600:]
        outs.append((p.stdout, open(d / "ex.n2.results").read()))
    assert outs[0][1] == outs[1][1]
    assert "on 2 GPU ranks" in outs[1][0]
