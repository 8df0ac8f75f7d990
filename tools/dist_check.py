"""
Two (or more) ranks sharing the box's GPU, gloo as the transport: do_optimization_distributed (rank-range shards + one
exchange) must return the `best` of do_optimization_single.  Run on the GPU box:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/dist_check.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import numpy as np
import torch
import torch.distributed as dist

dist.init_process_group("gloo")
rank = dist.get_rank()
os.environ["LOCAL_RANK"] = str(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))   # ranks share the box's GPU(s)
import parity_campaign as pc
from theta_amd.search import do_optimization_single, do_optimization_distributed

ok = True
for shape, seeds in (("toy", range(1001, 1041)), ("mid", range(2001, 2031))):
    pc.SHAPE = shape
    for seed in seeds:
        n = 2 if seed & 1 else 3
        inst = pc.instance(seed, n)
        args = (n, inst["m"], inst["k"], inst["tau"], list(inst["lb"]), list(inst["ub"]), inst["r"], inst["rN"], inst["mx"], inst["order"])
        try:
            single = do_optimization_single(*args)
        except SystemExit:
            single = []
        except Exception as e:      # e.g. a count that overflows: same on every rank
            print("rank", rank, "single failed", shape, seed, n, repr(e)[:100], flush=True)
            continue
        try:
            shard = do_optimization_distributed(*args, device=torch.device("cpu"))
        except SystemExit:
            print("rank", rank, "distributed exit", shape, seed, n, flush=True)
            shard = []
        same = len(single) == len(shard) and all(np.array_equal(a[0], b[0]) and abs(a[2] - b[2]) <= 1e-9 * abs(a[2]) for a, b in zip(single, shard))
        if not same:
            ok = False
            print("rank", rank, "MISMATCH", shape, seed, n, len(single), len(shard), flush=True)
print("rank %d: %s" % (rank, "all instances agree" if ok else "MISMATCHES"), flush=True)
dist.destroy_process_group()
