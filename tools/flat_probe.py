"""Where a flat likelihood's seconds go: one of the instances earlier campaigns gave up on, through mix_records with the clock off.
   python tools/flat_probe.py [seed]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import theta_amd
from theta_amd import search as S
import bnb_campaign as BC
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 70010
S.MIX_WALKABLE = 0
S.MIX_MAX_MS_LARGE = 120000.0
inst = BC.instance(seed)
ctx = theta_amd.default_context()
p = theta_amd.Problem(ctx, 3, inst["m"], inst["tau"], inst["r"], inst["rN"], inst["lb"], inst["ub"], 1.0)
rep = S.SearchReport()
t0 = time.time()
recs, _ = S.mix_records(p, ctx, inst["r"], inst["rN"], 1.0, (list(inst["lb"]), list(inst["ub"])), report=rep)
k = rep.mix
print("seed %d m=%d K=%d sum r %d: %.2f s, %d records" % (seed, inst["m"], inst["K"], sum(inst["r"]), time.time() - t0, len(recs)))
print("dive", k.get("dive"))
print("first walk", k.get("first_walk"))
print("heuristic", k.get("heuristic_seconds"))
for q in k["passes"]:
    print("pass", q)
print({x: k[x] for x in ("leaf_rel", "incumbent", "minimum", "threshold", "listed", "in_space", "records", "boxes_tested", "leaves", "kernel_ms", "search_ms", "syncs")})
