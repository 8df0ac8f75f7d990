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


def test_reference_outcome_class_of_every_table_entry():
    """
    hybrj restatement + the restated decision sequence of scipy's BFGS line search (n3_refbfgs.hpp), host build: the outcome
    class of EVERY entry of the reference's m=6, K=3 table -- own optimum / nu = 1/3 fallback / None -- is reproduced.
    """
    import ctypes as C
    import hybrj_check as hc
    hc.lib.hybrj_check_outcome.argtypes = [C.c_int, C.c_int, hc.dp, hc.dp, C.POINTER(C.c_uint8), hc.dp]
    g = np.load(os.path.join(GOLD, "solve_n3_m6k3.npz"))
    r, rN = g["r"].astype(float), g["rN"].astype(float)
    Cs, acc, nll = g["C"], g["accepted"].astype(bool), g["nll"]
    B, m, _ = Cs.shape
    full = np.concatenate([np.full((B, m, 1), 2.0), Cs.astype(float)], axis=2)
    Cw = full * rN[None, :, None]
    with np.errstate(all="ignore"):
        Ch = Cw / Cw.sum(1)[:, None, :]
        F = -(r[None, :] * np.log(Ch.sum(2) / 3.0)).sum(1)
        ref_fb = acc & (np.abs(F - nll) <= 1e-9 * np.abs(nll))
    ref = np.where(~acc, 0, np.where(acc & np.isnan(nll), -1, np.where(ref_fb, 2, 1)))
    wrong = 0
    counts = {0: 0, 1: 0, 2: 0}
    for k in range(B):
        c = np.ascontiguousarray(Cs[k], np.uint8)
        nu = np.zeros(3)
        o = hc.lib.hybrj_check_outcome(m, 2, r.ctypes.data_as(hc.dp), rN.ctypes.data_as(hc.dp), c.ctypes.data_as(C.POINTER(C.c_uint8)),
                                       nu.ctypes.data_as(hc.dp))
        if ref[k] >= 0:
            counts[o] += 1
            wrong += o != ref[k]
    assert counts[0] == 284 and counts[2] >= 4460 and counts[1] >= 16280
    assert wrong <= 2          # (one entry whose optimum IS the centre of the simplex carries the fallback value either way)
