"""
Parity campaign, run ON THE GPU BOX: many seeded random instances, the GPU driver (do_optimization_single through the
C ABI) against the CPU oracle's port of the reference driver (oracle.search_single: every candidate through the
reference's solver, sequential isClose rule), the oracle side spread over the host cores.

    python tools/parity_campaign.py [--n3 240] [--n2 240] [--seconds 420] > gpurun_out/parity_campaign.json

Checks per instance (the parity bar of BASELINE.json): chosen C identical, |d mu| < 1e-6, NLL within 1e-6 relative.
The COMPLETE `best` lists are compared entry by entry, n=2 and n=3, entries with a NaN likelihood included
(tests/campaign.py: compare_best).  Instances whose oracle side does not finish inside the budget are reported
as "unfinished", not as passes.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import numpy as np

SHAPE = "toy"
sys.path.insert(0, os.path.join(ROOT, "tests"))
import campaign


def instance(seed, n):
    return campaign.instance(seed, n, SHAPE)


def oracle_side(inst):
    import warnings
    warnings.simplefilter("ignore")
    import theta_oracle as orc
    t = time.time()
    best, cnt = orc.search_single(inst["n"], inst["m"], inst["tau"], list(inst["lb"]), list(inst["ub"]), inst["r"], inst["rN"],
                                  inst["mx"], inst["order"])
    return inst["seed"], inst["n"], cnt, campaign.best_to_plain(best), time.time() - t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n3", type=int, default=240)
    ap.add_argument("--n2", type=int, default=240)
    ap.add_argument("--seconds", type=float, default=420.0)
    ap.add_argument("--max-candidates", type=int, default=25000)
    ap.add_argument("--shape", choices=["toy", "mid", "low", "amp"], default="toy")
    ap.add_argument("--seed0", type=int, default=1000, help="first seed - 1 (other rounds' campaigns started at 1000)")
    a = ap.parse_args()
    global SHAPE
    SHAPE = a.shape
    import theta_amd
    from theta_amd.search import do_optimization_single
    import theta_amd.search as S
    ctx = theta_amd.Context(0)
    insts, gpu = [], {}
    t0 = time.time()
    seed = a.seed0
    want = {2: a.n2, 3: a.n3}
    got = {2: 0, 3: 0}
    while got[2] < want[2] or got[3] < want[3]:
        seed += 1
        n = 2 if (seed & 1) else 3
        if got[n] >= want[n]:
            continue
        inst = instance(seed, n)
        try:
            p = theta_amd.Problem(ctx, n, inst["m"], inst["tau"], inst["r"], inst["rN"], inst["lb"], inst["ub"], inst["mx"])
        except theta_amd.ThetaError:
            continue
        cnt = p.count
        p.close()
        if cnt < 50 or cnt > a.max_candidates * (8 if n == 2 else 1):
            continue
        try:
            best = do_optimization_single(n, inst["m"], inst["k"], inst["tau"], list(inst["lb"]), list(inst["ub"]), inst["r"],
                                          inst["rN"], inst["mx"], inst["order"])
        except SystemExit:
            best = []
        rep = S.last_report
        gpu[(seed, n)] = dict(best=[(np.asarray(b[0]).tolist(), [float(x) for x in b[1]], float(b[2])) for b in best], count=cnt,
                              uncertain=bool(getattr(rep, "parity_uncertain", False)) if best else False)
        insts.append(inst)
        got[n] += 1
    gpu_seconds = time.time() - t0
    insts.sort(key=lambda i: -gpu[(i["seed"], i["n"])]["count"] * (30 if i["n"] == 3 else 1))      # longest oracle jobs first
    cores = os.cpu_count() or 1
    res = {}
    pool = mp.get_context("fork").Pool(max(1, cores - 2))
    pending = [pool.apply_async(oracle_side, (i,)) for i in insts]
    deadline = time.time() + a.seconds
    for h in pending:
        try:
            s, n, cnt, best, dt = h.get(timeout=max(0.1, deadline - time.time()))
            res[(s, n)] = (cnt, best, dt)
        except mp.TimeoutError:
            pass
    pool.terminate()
    out = {"shape": a.shape, "instances": len(insts), "cores": cores, "gpu_seconds_all_instances": gpu_seconds, "unfinished": 0,
           "n2": {"checked": 0, "agree": 0, "both_empty": 0, "candidates": 0}, "n3": {"checked": 0, "agree": 0, "both_empty": 0,
                                                                                 "candidates": 0, "gpu_extra_tie_entries": 0,
                                                                                 "parity_uncertain_flagged": 0},
           "disagreements": []}
    for inst in insts:
        key = (inst["seed"], inst["n"])
        if key not in res:
            out["unfinished"] += 1
            continue
        cnt, ref, dt = res[key]
        g = gpu[key]
        tag = "n%d" % inst["n"]
        out[tag]["checked"] += 1
        out[tag]["candidates"] += cnt
        ok = True
        why = ""
        if cnt not in (g["count"], g["count"] + 1):                # (+1: quirk Q1, the first matrix is evaluated twice / extra)
            ok, why = False, "candidate counts %d vs %d" % (cnt, g["count"])
        elif not ref and not g["best"]:
            out[tag]["both_empty"] += 1
        elif bool(ref) != bool(g["best"]):
            ok, why = False, "one side empty (ref %d, gpu %d)" % (len(ref), len(g["best"]))
        else:
            # COMPLETE lists, entry by entry, NaN entries included (tests/campaign.py)
            why = campaign.compare_best(g["best"], ref)
            ok = why == ""
            if inst["n"] == 3:
                out[tag]["parity_uncertain_flagged"] += int(g["uncertain"])
        if ok:
            out[tag]["agree"] += 1
        else:
            out["disagreements"].append({"seed": inst["seed"], "n": inst["n"], "m": inst["m"], "k": inst["k"], "why": why,
                                         "gpu_first": g["best"][:1], "ref_first": ref[:1]})
    out["oracle_seconds_wall"] = time.time() - t0 - gpu_seconds
    print(json.dumps(out))


if __name__ == "__main__":
    main()
