#!/usr/bin/env python3
"""
`best` of the REFERENCE ITSELF for n=3 with tau = 1 and tau = 3 (--TAU; every other fixture has the default 2): the toy and mid
instances of tests/campaign.py regenerated under another tau (rows of the truth scaled accordingly by the generator's own rule).
Build container only; data only is written: tests/golden/best_tau.json.

    python tests/golden/make_golden_tau.py
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


def retau(inst, tau, seed):
    """the same bounds and normal counts, tumour counts redrawn from a truth inside the bounds under the new tau"""
    import theta_oracle as orc
    rng = np.random.RandomState(seed)
    m = inst["m"]
    rN = np.array(inst["rN"], float)
    C = np.full((m, 3), float(tau))
    for j in (1, 2):
        C[:, j] = [rng.randint(l, u + 1) for l, u in zip(inst["lb"], inst["ub"])]
    mu = rng.dirichlet(np.ones(3) * 3)
    p = (C * rN[:, None]) @ mu
    p = p / p.sum()
    r = np.maximum(rng.multinomial(int(rN.sum() * rng.uniform(0.8, 1.5)), p), 1)
    out = dict(inst)
    out["tau"] = tau
    out["r"] = [int(x) for x in r]           # (kept in the instance's interval order: `order` is what the driver un-sorts with)
    return out


def main():
    insts = []
    for shape, want in (("toy", 16), ("mid", 12)):
        seed, got = 11000, 0
        while got < want:
            seed += 1
            base = campaign.instance(seed, 3, shape)
            inst = retau(base, 1 if seed % 2 else 3, seed)
            try:
                cnt = campaign.count_candidates(inst)
            except Exception:
                continue
            if not (200 <= cnt <= 12000):
                continue
            inst["count"] = int(cnt)
            insts.append(inst)
            got += 1
    insts.sort(key=lambda i: -i["count"])
    with mp.get_context("fork").Pool(max(1, (os.cpu_count() or 2) - 1)) as pool:
        res = pool.map(run, insts, chunksize=1)
    res.sort(key=lambda i: (i["shape"], i["seed"]))
    with open(os.path.join(HERE, "best_tau.json"), "w") as f:
        json.dump({"cases": res}, f, separators=(",", ":"))
    print("wrote best_tau.json: %d instances, %d candidates, %d with an empty list" % (len(res), sum(c["count"] for c in res), sum(1 for c in res if not c["best"])))


if __name__ == "__main__":
    main()
