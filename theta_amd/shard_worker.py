"""
One shard of `do_optimization(..., max_processes)` (python/RunTHetA.py:124-171): the process `theta_amd.search` starts per
additional GPU.  The reference forks max_processes - 1 CPU workers that pull matrices from a multiprocessing.Queue
(RunTHetA.py:53-105, 136-141) and merges their lists with find_mins (:107-122); here every worker owns ONE GPU and a
contiguous rank range of the candidate space, and the merge is the library's exchange (theta_exchange_finalists: RCCL over
xGMI, csrc/comm.hip).  Rank 0 is the calling process itself.

    python -m theta_amd.shard_worker <payload.pickle> <status.pickle>

The payload is written by search._spawn_shards: rank, world, rendezvous port, transport, device index and the arguments of
do_optimization.  The worker prints nothing (rank 0 speaks for the job); its status -- "ok" or the error's code and text --
goes to the status file.  A worker is started as a fresh interpreter, not forked: the parent has a live HIP context.
"""
import os
import pickle
import sys


def main(argv):
    payload_path, status_path = argv[1], argv[2]
    with open(payload_path, "rb") as f:
        job = pickle.load(f)
    status = {"rank": job["rank"], "state": "error", "code": -1, "message": "worker did not finish"}
    try:
        for p in reversed(job.get("sys_path", [])):
            if p not in sys.path:
                sys.path.insert(0, p)
        from theta_amd import _lib, search
        if job.get("init"):
            # (tests: a module-level function "module:function" that returns the context to use -- a stand-in device on
            # machines without a GPU, tests/standin_device.py; never set by the package itself)
            import importlib
            mod, fn = job["init"].split(":")
            ctx = getattr(importlib.import_module(mod), fn)(job["rank"])
        else:
            ctx = _lib.Context(job["device"])
        comm = _lib.Comm(ctx if job["transport"] != "host" or hasattr(ctx, "_h") else None, rank=job["rank"], world=job["world"],
                         addr="127.0.0.1", port=job["port"], transport=job["transport"])
        try:
            best = search.do_optimization_distributed(*job["args"], comm=comm, ctx=ctx)
            rep = search.last_report
            status = {"rank": job["rank"], "state": "ok", "entries": len(best), "window": rep.window,
                      "kernel_ms": float(rep.stats.get("kernel_ms", 0.0)) if rep.stats else 0.0,
                      "evaluated": int(rep.stats.get("evaluated", 0)) if rep.stats else 0}
        finally:
            comm.close()
    except SystemExit as e:
        status = {"rank": job["rank"], "state": "exit", "code": e.code, "message": "exit"}
    except BaseException as e:          # the parent reads the status file: never leave it without one
        status = {"rank": job["rank"], "state": "error", "code": getattr(e, "code", -1), "message": repr(e)}
    with open(status_path + ".tmp", "wb") as f:
        pickle.dump(status, f)
    os.replace(status_path + ".tmp", status_path)
    return 0 if status["state"] == "ok" else 1


if __name__ == "__main__":
    devnull = open(os.devnull, "w")
    sys.stdout = devnull
    os.dup2(devnull.fileno(), 1)          # (the C stdout as well: RCCL prints a version banner there, and fd 1 is the parent's)
    sys.exit(main(sys.argv))
