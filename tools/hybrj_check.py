"""
Checks the hybrj restatement (theta_amd/csrc/hybrj4.hpp, host build tools/hybrj_check.cpp) against scipy.optimize.fsolve on
the Lagrangian system of Optimizer._solve_n3plus -- candidate by candidate on the reference's own m=6, K=3 table
(tests/golden/solve_n3_m6k3.npz) and on seeded mid-size instances.  CPU only; run where scipy is.
"""
import ctypes as C
import os
import subprocess
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from scipy import optimize

so = os.path.join(ROOT, "build_ab", "libhybrj_check.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-builtin-pow", "-shared", "-fPIC", os.path.join(ROOT, "tools", "hybrj_check.cpp"), "-o", so])
lib = C.CDLL(so)
dp = C.POINTER(C.c_double)
lib.hybrj_check_solve.argtypes = [C.c_int, C.c_int, dp, dp, C.POINTER(C.c_uint8), dp, C.POINTER(C.c_int)]
lib.hybrj_check_M3.argtypes = [dp, dp, dp, C.POINTER(C.c_int)]
lib.hybrj_check_table.argtypes = [C.c_int, C.c_int, C.c_int, dp, dp, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), dp, dp]


def solve_table(C_u8, r, rN, tau=2):
    """theta_solve_batch's per-candidate procedure (n3_ref_solve, the function the device kernel calls) on the host:
    returns ok (0 None / 1 own iterate / 2 fallback), mu (B, 3), nll (B)."""
    C_u8 = np.ascontiguousarray(C_u8, np.uint8)
    B, m = C_u8.shape[0], C_u8.shape[1]
    r = np.ascontiguousarray(r, np.float64)
    rN = np.ascontiguousarray(rN, np.float64)
    ok, mu, nll = np.zeros(B, np.uint8), np.zeros((B, 3)), np.zeros(B)
    lib.hybrj_check_table(B, m, tau, r.ctypes.data_as(dp), rN.ctypes.data_as(dp), C_u8.ctypes.data_as(C.POINTER(C.c_uint8)),
                          ok.ctypes.data_as(C.POINTER(C.c_uint8)), mu.ctypes.data_as(dp), nll.ctypes.data_as(dp))
    return ok, mu, nll


def m3(S, nu):
    """Optimizer.M3 restated (hybrd): mu for column sums S and mixture nu."""
    S = np.ascontiguousarray(S, np.float64)
    nu = np.ascontiguousarray(nu, np.float64)
    mu = np.zeros(3)
    nf = C.c_int()
    info = lib.hybrj_check_M3(S.ctypes.data_as(dp), nu.ctypes.data_as(dp), mu.ctypes.data_as(dp), C.byref(nf))
    return mu, info, nf.value


def mine(c_u8, r, rN, tau=2):
    nu = np.zeros(3)
    nfev = C.c_int()
    c_u8 = np.ascontiguousarray(c_u8, np.uint8)
    info = lib.hybrj_check_solve(c_u8.shape[0], tau, r.ctypes.data_as(dp), rN.ctypes.data_as(dp), c_u8.ctypes.data_as(C.POINTER(C.c_uint8)),
                                 nu.ctypes.data_as(dp), C.byref(nfev))
    return nu, info, nfev.value


def scipy_side(c_u8, r, rN, tau=2):
    m = c_u8.shape[0]
    Cm = np.zeros((m, 3))
    Cm[:, 0] = tau
    Cm[:, 1:] = c_u8
    import theta_oracle as orc
    Cw = orc.weighted_C(Cm, rN)
    Ch = orc.normalize_C(Cw, m, 3)
    n = 3
    numer = [[r[i] * Ch[i][k] for i in range(m)] for k in range(n)]

    def dmu(x, k):
        acc = 0
        for i in range(m):
            acc += numer[k][i] / sum([Ch[i][j] * x[j] for j in range(n)])
        return (-acc) - x[n]

    def eqs(x):
        return [dmu(x, k) for k in range(n)] + [1 - sum(x[:n])]

    def second(x, k, h):
        acc = 0
        for i in range(m):
            acc += (r[i] * Ch[i][k] * Ch[i][h]) / (sum([Ch[i][j] * x[j] for j in range(n)]) ** 2)
        return acc

    def jac(x):
        J = np.zeros((n + 1, n + 1))
        for i in range(n + 1):
            J[n][i] = -1
            J[i][n] = -1
        J[n][n] = 0
        for i in range(n):
            for j in range(n):
                J[i][j] = second(x, i, j)
        return J

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        x, info, ier, msg = optimize.fsolve(eqs, [1.0 / n] * n + [1], fprime=jac, full_output=True)
    return x[:3], ier, info["nfev"]


def in_range(v):
    return not any(x < 0 or x > 1 for x in v)


def main():
    g = np.load(os.path.join(ROOT, "tests", "golden", "solve_n3_m6k3.npz"))
    r, rN = g["r"].astype(float), g["rN"].astype(float)
    rng = np.random.RandomState(1)
    idx = rng.choice(len(g["C"]), int(sys.argv[1]) if len(sys.argv) > 1 else 3000, replace=False)
    same_x = same_cls = same_nfev = 0
    bad = []
    for k in idx:
        c = g["C"][k]
        a, ia, na = mine(c, r, rN)
        b, ib, nb = scipy_side(c, r, rN)
        cls_a, cls_b = in_range(a), in_range(b)
        ok = np.allclose(a, b, rtol=1e-9, atol=1e-12, equal_nan=True)
        same_x += ok
        same_cls += cls_a == cls_b
        same_nfev += na == nb
        if cls_a != cls_b and len(bad) < 8:
            bad.append((int(k), a.tolist(), b.tolist(), ia, ib, na, nb))
    print("m6k3 table sample %d: identical iterate %d, same in-range class %d, same nfev %d" % (len(idx), same_x, same_cls, same_nfev))
    for b in bad:
        print("  class differs:", b)


if __name__ == "__main__":
    main()


def table():
    """Every entry of the reference's table: does 'hybrj iterate in range' predict 'reported with its own optimum' / 'fallback'?"""
    g = np.load(os.path.join(ROOT, "tests", "golden", "solve_n3_m6k3.npz"))
    r, rN = g["r"].astype(float), g["rN"].astype(float)
    Cs = g["C"]
    acc = g["accepted"].astype(bool)
    nll = g["nll"]
    B, m, _ = Cs.shape
    full = np.concatenate([np.full((B, m, 1), 2.0), Cs.astype(float)], axis=2)
    Cw = full * rN[None, :, None]
    S = Cw.sum(1)
    with np.errstate(all="ignore"):
        Ch = Cw / S[:, None, :]
        F = -(r[None, :] * np.log(Ch.sum(2) / 3.0)).sum(1)
        ref_fb = acc & (np.abs(F - nll) <= 1e-9 * np.abs(nll))
    ref_in = acc & ~ref_fb & np.isfinite(nll)
    pred_in = np.zeros(B, bool)
    val_ok = np.zeros(B, bool)
    for k in range(B):
        nu, info, nf = mine(Cs[k], r, rN)
        pred_in[k] = in_range(nu)
        if pred_in[k] and ref_in[k]:
            with np.errstate(all="ignore"):
                v = -(r * np.log(Ch[k] @ nu)).sum()
            val_ok[k] = abs(v - nll[k]) <= 1e-9 * abs(nll[k])
    print("reference: own optimum %d, fallback %d, None %d, NaN %d" % (ref_in.sum(), ref_fb.sum(), (~acc).sum(), (acc & np.isnan(nll)).sum()))
    print("hybrj iterate in range & reference own-optimum: %d (value equal: %d); in range & reference fallback: %d; in range & None/NaN: %d"
          % ((pred_in & ref_in).sum(), val_ok.sum(), (pred_in & ref_fb).sum(), (pred_in & ~ref_in & ~ref_fb).sum()))
    print("hybrj iterate out of range & reference fallback: %d; out of range & own-optimum: %d; out of range & None/NaN: %d"
          % ((~pred_in & ref_fb).sum(), (~pred_in & ref_in).sum(), (~pred_in & ~ref_in & ~ref_fb).sum()))


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "table":
    table()
