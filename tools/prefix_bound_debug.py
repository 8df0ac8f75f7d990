"""Debug: n3_prefix_bound on / off on whole spaces; where the lists differ, the relaxed bound of the best candidate's prefix on the CPU."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import theta_amd, bench
from test_gpu_wide import _wide_instance
from scipy.optimize import minimize

ctx = theta_amd.default_context()


def cpu_bound(r, rN, C, tau, ML):
    """relaxed lower bound of the prefix of candidate C (rows (x, y)), leaf rows = last ML"""
    r = np.asarray(r, float); rN = np.asarray(rN, float)
    m = len(r); D = m - ML
    Nn = rN / rN.sum(); Rtot = r.sum(); K0 = -(r * np.log(Nn)).sum()
    pre = C[:D]
    Rp = r[:D].sum(); Nrem = Nn[D:].sum()
    zred = np.array([1 - Nrem, (Nn[:D] * pre[:, 0]).sum(), (Nn[:D] * pre[:, 1]).sum()])
    def F(v):
        w = np.array([1.0, v[0], v[1]])
        q = w[0] + pre[:, 0] * w[1] + pre[:, 1] * w[2]
        if (q <= 0).any() or zred @ w <= 0: return 1e300
        return K0 - (r[:D] * np.log(q)).sum() + Rp * np.log(zred @ w)
    fm = min(minimize(F, s, method='Nelder-Mead', options={'xatol': 1e-12, 'fatol': 1e-12, 'maxiter': 20000}).fun for s in ([.2, .2], [.05, .5], [.5, .05], [1, 1]))
    const = Rp * np.log(Rtot / Rp) - sum(r[l] * np.log(r[l] / (Rtot * Nn[l])) for l in range(D, m) if r[l] > 0)
    def nll(v):
        w = np.array([1.0, v[0], v[1]])
        q = w[0] + C[:, 0] * w[1] + C[:, 1] * w[2]
        z = np.array([1.0, (Nn * C[:, 0]).sum(), (Nn * C[:, 1]).sum()])
        if (q <= 0).any() or z @ w <= 0: return 1e300
        return K0 - (r * np.log(q)).sum() + Rtot * np.log(z @ w)
    cm = min(minimize(nll, s, method='Nelder-Mead', options={'xatol': 1e-12, 'fatol': 1e-12, 'maxiter': 20000}).fun for s in ([.2, .2], [.05, .5], [.5, .05], [1, 1]))
    return fm + const, cm


def run(name, m, tau, rr, rn, lb, ub, b=None, e=None):
    p = theta_amd.Problem(ctx, 3, m, tau, rr, rn, lb, ub, 1.0)
    b = 0 if b is None else b
    e = p.count if e is None else e
    out = {}
    for pb in (0, 1):
        p.set_option("n3_prefix_bound", pb)
        try:
            res = p.search(b, e, window=0.5)
            st = res["stats"]
            out[pb] = res
            print(name, "pb", pb, "count", e - b, "min", (res["nll"].min() if len(res["nll"]) else None), "finalists", len(res["rank"]), "suspects", len(p.last_suspects[0]),
                  "pruned", st["pruned"], "survivors", st["survivors"], "kernel ms %.2f" % st["kernel_ms"], flush=True)
        except Exception as ex:
            print(name, "pb", pb, "FAILED", str(ex)[:200], flush=True)
    if 0 in out and 1 in out and (out[0]["rank"] != out[1]["rank"]):
        print("  LISTS DIFFER", out[0]["rank"][:3], out[1]["rank"][:3])
    if 0 in out and len(out[0]["nll"]):
        k = int(np.argmin(out[0]["nll"]))
        C = np.asarray(out[0]["C"][k], float).reshape(m, 2)
        ML = 6 if m >= 10 else 4
        lbd, cm = cpu_bound(rr, rn, C, tau, ML)
        print("  best of pb=0: rank", out[0]["rank"][k], "nll", out[0]["nll"][k], "unconstrained min (cpu)", cm, "relaxed prefix bound (cpu)", lbd, "OK" if lbd <= cm + 1e-6 else "BOUND VIOLATED")
    p.close()


rs9, rNs9, _o, _t, lb9, ub9 = _wide_instance(100, 501, 2)
run("m100 wide", 100, 2, rs9, rNs9, lb9, ub9)
r6, rN6, _ = bench.synth(seed=9, m=14, n=3, k=3)
run("m14 k3 ragged", 14, 2, r6, rN6, [0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2], [2, 2, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3])
ra, rNa, _ = bench.synth(seed=14, m=16, n=3, k=3)
run("m16 k3", 16, 2, ra, rNa, [0] * 16, [3] * 16)
rb, rNb, _ = bench.synth(seed=15, m=12, n=3, k=4)
run("m12 k4 tau3", 12, 3, rb, rNb, [0] * 12, [4] * 12)
