"""
The hybrj restatement (theta_amd/csrc/hybrj4.hpp + n3_refsys.hpp: MINPACK's solver on the Lagrangian system of
Optimizer._solve_n3plus, in the reference's operation order) against scipy.optimize.fsolve, on the host: the same header
compiled with g++ (tools/hybrj_check.cpp).  Decides, per candidate, whether the reference reports the candidate's own
optimum or its nu = (1/3,1/3,1/3) fallback -- so the iterate has to be the same one, not just a root.
"""
import os
import sys

import numpy as np

from conftest import GOLD, ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_hybrj_restatement_follows_scipy_fsolve_on_the_reference_table():
    import hybrj_check as hc
    g = np.load(os.path.join(GOLD, "solve_n3_m6k3.npz"))
    r, rN = g["r"].astype(float), g["rN"].astype(float)
    acc, nll = g["accepted"].astype(bool), g["nll"]
    rng = np.random.RandomState(5)
    idx = rng.choice(len(g["C"]), 400, replace=False)
    same_iterate = same_class = 0
    for k in idx:
        a, ia, na = hc.mine(g["C"][k], r, rN)
        b, ib, nb = hc.scipy_side(g["C"][k], r, rN)
        same_iterate += np.allclose(a, b, rtol=1e-9, atol=1e-12, equal_nan=True)
        same_class += hc.in_range(a) == hc.in_range(b)
    assert same_class == len(idx) and same_iterate >= len(idx) - 2
    # ... and against the table itself: iterate in [0,1]^3 <=> the reference reported the candidate's own optimum
    C = g["C"].astype(float)
    B, m, _ = C.shape
    full = np.concatenate([np.full((B, m, 1), 2.0), C], axis=2)
    Cw = full * rN[None, :, None]
    with np.errstate(all="ignore"):
        Ch = Cw / Cw.sum(1)[:, None, :]
        F = -(r[None, :] * np.log(Ch.sum(2) / 3.0)).sum(1)
        ref_fb = acc & (np.abs(F - nll) <= 1e-9 * np.abs(nll))
    ref_own = acc & ~ref_fb & np.isfinite(nll)
    wrong = 0
    for k in idx:
        inr = hc.in_range(hc.mine(g["C"][k], r, rN)[0])
        if ref_own[k]:
            wrong += not inr
        elif ref_fb[k] or not acc[k]:
            wrong += inr
    assert wrong <= 1
