#!/usr/bin/env python3
"""
`best` of the REFERENCE ITSELF on small spaces built AROUND matrices of full rank that the reference reports with a NaN likelihood
(found by tools/nan_hunt.py in the seeded instances mid/20036, mid/20123, mid/20312): the instance's counts, bounds tightened
around the matrix (a random subset of the intervals keeps one copy number of slack) until the space holds 3 000 .. 40 000
matrices.  Where the NaN matrix stands behind the last replacement of the minimum, the reference's `best` carries its NaN tuple
(isClose(NaN), Misc.py:44-46) -- the case of the search's NaN sweep.  Build container only; data only is written:
tests/golden/best_nan_cases.json.

    python tests/golden/make_golden_nan_cases.py
"""
import json
import multiprocessing as mp
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "oracle"))

import numpy as np

import campaign
from make_golden_campaign import run

# (seed of the mid-shape instance, the matrix's tumour columns in the instance's -- sorted -- interval order)
FOUND = [
    (20036, [[0, 0], [0, 1], [2, 0], [2, 1], [2, 1], [2, 2], [2, 2], [2, 2], [2, 2], [3, 2], [3, 2], [3, 3], [3, 3]]),
    (20123, [[0, 0], [0, 1], [1, 1], [1, 1], [2, 1], [2, 2], [3, 2], [3, 2], [3, 3], [3, 3], [3, 3], [3, 3], [4, 4], [5, 5], [5, 5]]),
    (20123, [[0, 0], [0, 1], [1, 1], [1, 1], [2, 1], [3, 2], [3, 2], [2, 3], [3, 3], [4, 3], [4, 4], [4, 4], [4, 4], [5, 5], [5, 5]]),
    (20123, [[0, 0], [0, 1], [1, 1], [1, 1], [2, 1], [3, 2], [3, 2], [4, 3], [4, 3], [4, 3], [3, 4], [4, 3], [4, 4], [5, 5], [5, 5]]),
    (20312, [[0, 0], [1, 1], [1, 1], [1, 1], [0, 2], [1, 2], [2, 1], [2, 2], [2, 2], [2, 2], [2, 4], [3, 3]]),
    (20312, [[0, 0], [1, 1], [0, 2], [2, 0], [1, 1], [2, 1], [1, 2], [2, 2], [2, 2], [2, 2], [4, 2], [3, 3]]),
]


def main():
    insts = []
    for idx, (seed, rows) in enumerate(FOUND):
        base = campaign.instance(seed, 3, "mid")
        C = np.array(rows)
        assert len(C) == base["m"]
        rng = np.random.RandomState(1000 + idx)
        made = 0
        for _try in range(400):
            free = rng.rand(base["m"]) < rng.choice([0.25, 0.35, 0.5])
            inst = dict(base)
            inst["lb"] = [int(max(0, min(a, b) - (1 if f else 0))) for (a, b), f in zip(C, free)]
            inst["ub"] = [int(min(base["k"], max(a, b) + (1 if f else 0))) for (a, b), f in zip(C, free)]
            cnt = campaign.count_candidates(inst)
            if 3000 <= cnt <= 40000:
                inst["count"] = int(cnt)
                inst["shape"] = "nan%d_%d" % (idx, made)
                inst["around"] = rows
                insts.append(inst)
                made += 1
                if made == 3:
                    break
    insts.sort(key=lambda i: -i["count"])
    with mp.get_context("fork").Pool(max(1, (os.cpu_count() or 2) - 2)) as pool:
        res = pool.map(run, insts, chunksize=1)
    res.sort(key=lambda i: i["shape"])
    with open(os.path.join(HERE, "best_nan_cases.json"), "w") as f:
        json.dump({"cases": res}, f, separators=(",", ":"))
    nan_entries = sum(1 for c in res for b in c["best"] if b["nll"] == "nan")
    print("wrote best_nan_cases.json: %d instances, %d candidates, %d NaN entries in the best lists" % (len(res), sum(c["count"] for c in res), nan_entries))


if __name__ == "__main__":
    main()
