"""Round 5 scratch: are the counters of theta_search / theta_search_witness reproducible run to run and build to build?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench, theta_amd
ctx = theta_amd.default_context()
rr, rn, _ = bench.synth()
p = theta_amd.Problem(ctx, 3, 50, 2, rr, rn, [0] * 50, [6] * 50, 1.0)
span = 1 << 18
for where, b in (("start", 0), ("middle", p.count // 3)):
    probe = p.search(b, b + (1 << 16), window=0.0)
    known = float(probe["nll"].min())
    for name, opts in (("f32", {"n3_no_dismiss": 1}), ("f64", {"n3_no_dismiss": 1, "n3_force_f64": 1})):
        for k, v in opts.items():
            p.set_option(k, v)
        for rep in range(3):
            p.hint(known)
            a = p.search(b, b + span, window=0.5)["stats"]
            p.hint(known)
            _r, w = p.witness(b, b + span, 8, 0.5)
            print(where, name, rep, "plain", a["iterations"], a["terms"], a["survivors"], a["accepted"], "witness", w["iterations"], w["terms"], w["survivors"], w["accepted"])
        for k in opts:
            p.set_option(k, 0)
