"""
The N>1 path on CPU: two processes over gloo exercise the ONE exchange step of a sharded search
(all-reduce(min) + all-gather of finalists, theta_amd.search.exchange_finalists) and the tie replay
on the merged list.  The per-shard finalists are synthetic here (no GPU in this container); on the
GPU box the same function runs over RCCL.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

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
    nlls = [5000.0 + 0.0004 * rank, 5000.2, 5003.0 + rank] if rank == 0 else [5000.0003, 4999.9998, 5000.45]
    for j, v in enumerate(nlls):
        c = rng.randint(0, 5, (m, 2) if n == 3 else (m,)).astype(np.uint8)
        recs.append({"rank": base + 17 * j + rank, "c": c, "mu": rng.dirichlet(np.ones(n)), "nll": v,
                     "vals": rng.dirichlet(np.ones(m))})
    return recs


def _worker(rank, world, port, n, m, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from theta_amd.search import exchange_finalists, replay_ties
    merged = exchange_finalists(_shard_records(rank, n, m), n, m, torch.device("cpu"))
    best = replay_ties(merged, n, 2, list(range(m)), first_duplicate=False)
    out[rank] = ([(t["rank"], t["nll"], t["c"].tolist(), t["mu"].tolist(), t["vals"].tolist()) for t in merged],
                 [(b[2], b[0].tolist()) for b in best])
    dist.destroy_process_group()


@pytest.mark.parametrize("n,m", [(3, 7), (2, 5)])
def test_exchange_and_replay_two_ranks(n, m):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, m, out), nprocs=world, join=True)
    assert set(out.keys()) == {0, 1}
    assert out[0] == out[1]                                    # every rank ends with the same answer
    merged, best = out[0]
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
