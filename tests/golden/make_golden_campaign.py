#!/usr/bin/env python3
"""
`best` of the REFERENCE ITSELF (python/RunTHetA.py do_optimization_single, imported the way make_golden.py imports it:
lib2to3 copy under /tmp, three shims) on seeded random instances of tests/campaign.py -- complete lists, including the
entries with a NaN likelihood the reference appends through isClose(NaN) (Misc.py:44-46) for matrices with an all-zero tumour
column.  Runs only in the build container (needs /root/reference).  Data only is written: tests/golden/best_campaign.json.

    python tests/golden/make_golden_campaign.py              # best_campaign.json  (seeds 5001.., 50..15 000 candidates per n=3 instance)
    python tests/golden/make_golden_campaign.py second       # best_campaign2.json (seeds 7001.., larger spaces: up to 60 000 / 200 000)
    python tests/golden/make_golden_campaign.py third        # best_campaign3.json (seeds 8001.., n=3, the low-coverage shape + mid)
    python tests/golden/make_golden_campaign.py fourth       # best_campaign4.json (whole spaces with full-rank NaN outcomes, tools/nan_hunt.py)
    python tests/golden/make_golden_campaign.py fifth        # best_campaign5.json (seeds 9001.., n=3, bounds from the reference's own heuristic
                                                             #   on counts with a strongly amplified interval: copy numbers 8 - 10)
"""
import json
import multiprocessing as mp
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "oracle"))

import numpy as np

import campaign
from make_golden import fl, import_reference

WANT = {(2, "toy"): 30, (2, "mid"): 20, (3, "toy"): 40, (3, "mid"): 30}
LIMIT = {2: (50, 40000), 3: (50, 15000)}


def run(inst):
    R = import_reference()
    t = time.time()
    try:
        best = R.do_optimization_single(inst["n"], inst["m"], inst["k"], inst["tau"], list(inst["lb"]), list(inst["ub"]),
                                        list(inst["r"]), list(inst["rN"]), inst["mx"], list(inst["order"]), True, False)
    except SystemExit:
        best = []
    out = dict(inst)
    out["best"] = [{"C": np.asarray(b[0]).tolist(), "mu": [fl(x) for x in b[1]], "nll": fl(b[2])} for b in best]
    out["ref_seconds"] = time.time() - t
    return out


SECOND = {"want": {(2, "toy"): 20, (2, "mid"): 20, (3, "toy"): 40, (3, "mid"): 40}, "limit": {2: (2000, 200000), 3: (2000, 60000)},
          "seed0": 7000, "out": "best_campaign2.json"}


THIRD = {"want": {(3, "low"): 80, (3, "mid"): 30}, "limit": {3: (500, 40000)}, "seed0": 8000, "out": "best_campaign3.json"}


# instances tools/nan_hunt.py found to hold FULL-RANK matrices the reference reports with a NaN likelihood (whole spaces of 4e5..5e5
# matrices: a quarter of an hour of the reference each)
FOURTH = {"list": [(20036, 3, "mid"), (20123, 3, "mid")], "out": "best_campaign4.json"}


def main():
    insts = []
    want_tab, limit, seed0, out_name = WANT, LIMIT, 5000, "best_campaign.json"
    if len(sys.argv) > 1 and sys.argv[1] == "second":
        want_tab, limit, seed0, out_name = SECOND["want"], SECOND["limit"], SECOND["seed0"], SECOND["out"]
    if len(sys.argv) > 1 and sys.argv[1] == "third":
        want_tab, limit, seed0, out_name = THIRD["want"], THIRD["limit"], THIRD["seed0"], THIRD["out"]
    if len(sys.argv) > 1 and sys.argv[1] == "fourth":
        out_name, want_tab = FOURTH["out"], {}
        for seed, n, shape in FOURTH["list"] + [(int(a), 3, "mid") for a in sys.argv[2:]]:
            inst = campaign.instance(seed, n, shape)
            inst["count"] = int(campaign.count_candidates(inst))
            insts.append(inst)
    if len(sys.argv) > 1 and sys.argv[1] == "fifth":
        out_name, want_tab = "best_campaign5.json", {}
        seed = 9000
        while len(insts) < 14:
            seed += 1
            inst = campaign.instance(seed, 3, "amp")
            cnt = campaign.count_candidates(inst)
            if max(inst["ub"]) < 8 or cnt > 70000:
                continue
            inst["count"] = int(cnt)
            insts.append(inst)
    for (n, shape), want in want_tab.items():
        seed, got = seed0, 0
        while got < want:
            seed += 1
            inst = campaign.instance(seed, n, shape)
            cnt = campaign.count_candidates(inst)
            if not (limit[n][0] <= cnt <= limit[n][1]):
                continue
            inst["count"] = int(cnt)
            insts.append(inst)
            got += 1
    insts.sort(key=lambda i: -i["count"] * (40 if i["n"] == 3 else 1))
    with mp.get_context("fork").Pool(os.cpu_count() or 1) as pool:
        res = pool.map(run, insts, chunksize=1)
    res.sort(key=lambda i: (i["n"], i["shape"], i["seed"]))
    with open(os.path.join(HERE, out_name), "w") as f:
        json.dump({"cases": res}, f, separators=(",", ":"))
    nan_entries = sum(1 for c in res for b in c["best"] if b["nll"] == "nan")
    print("wrote " + out_name + ": %d instances, %d candidates, %d NaN entries in the best lists, %.0f s of reference time"
          % (len(res), sum(c["count"] for c in res), nan_entries, sum(c["ref_seconds"] for c in res)))


if __name__ == "__main__":
    main()
