"""CPU prototype (numpy) of the next step DESIGN.md section 8 names: BRANCH AND BOUND ABOVE THE PREFIX.

The relaxation behind n3_sieve.hip's sv_prefix_beyond holds for ANY set of fixed rows: with rows 0 .. d-1 of a candidate fixed,
    min over completions of NLL  >=  min_w [ K0 - sum_{i<d} r_i log q_i(w) + Rp log(z_d . w) ]  +  const_d,
    const_d = Rp log(Rtot / Rp) - sum_{l>=d} r_l log(r_l / (Rtot Nn_l)),      Rp = sum_{i<d} r_i
(the freed intervals fitted perfectly; tools/prefix_bound_check.py checks it against sampled completions).  Applied while the
prefix is BUILT -- depth 1, 2, 3 ... -- it prunes whole subtrees; this script counts, depth by depth, how many prefixes survive
against the incumbent (the planted matrix's own optimum + the driver's collection window) on seeded instances of the bench's
generator, WITHOUT the reference's row-graph constraints (a superset of its space: the counts are upper bounds).  What it says
about the plan: how many 2-parameter convex solves an exact arg-min of a space of (K+1)^(2m) matrices needs.

    python tools/bnb_prototype.py [m] [K] [seed] [max nodes per depth]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

WINDOW = float(os.environ.get("BNB_WINDOW", "0.5"))          # theta_amd.search.COLLECT_WINDOW (the reference's own tie margin is 1e-3: Misc.py:36)


def synth_with_truth(seed, m, k):
    """bench.synth's generator, returning the planted matrix too (same draws in the same order)."""
    rng = np.random.RandomState(seed)
    L = rng.randint(2_000_000, 20_000_000, m)
    rN = np.maximum(rng.poisson(L * 0.01), 1)
    C = np.full((m, 3), float(bench.TAU))
    for j in range(1, 3):
        C[:, j] = rng.randint(0, k + 1, m)
    mu = rng.dirichlet(np.ones(3) * 4)
    p = (C * rN[:, None]) @ mu
    p /= p.sum()
    r = rng.multinomial(int(rN.sum() * 1.2), p)
    order = np.argsort(-rN, kind="stable")            # largest intervals first (the reference sorts by size: DataTools.sort_r)
    return r[order].astype(float), rN[order].astype(float), C[order, 1:].astype(int), mu


def bound(X, Y, r, Nn, K0, Rtot, W, iters=40):
    """Relaxed lower bound of every completion, for a batch of prefixes.  X, Y: (B, d) rows; W: (B, 2) warm starts (u1, u2 of the
    parent).  Point w = (w0, u1, u2) on the slice z.w = 1 with z = (z0, z1, z2) = sums of Nn (1, x, y) over the fixed rows."""
    B, d = X.shape
    rd, nd = r[:d], Nn[:d]
    Rp = rd.sum()
    z0 = nd.sum()
    z1, z2 = X @ nd, Y @ nd
    a, b = X - (z1 / z0)[:, None], Y - (z2 / z0)[:, None]       # q_i = 1/z0 + a_i u1 + b_i u2
    u = W.copy()

    def val(u):
        q = 1.0 / z0 + a * u[:, :1] + b * u[:, 1:]
        ok = (q > 0).all(axis=1)
        with np.errstate(invalid="ignore", divide="ignore"):
            f = -(rd * np.log(np.where(q > 0, q, 1.0))).sum(axis=1)
        return np.where(ok, f, np.inf), q

    f, q = val(u)
    bad = ~np.isfinite(f)
    if bad.any():                                               # a warm start outside the child's domain: the slice's centre
        u[bad] = 0.0
        f, q = val(u)
    for _ in range(iters):
        al, be = a / q, b / q
        g1, g2 = -(rd * al).sum(axis=1), -(rd * be).sum(axis=1)
        h11, h12, h22 = (rd * al * al).sum(axis=1), (rd * al * be).sum(axis=1), (rd * be * be).sum(axis=1)
        # (d <= 2 or rows on a line: singular Hessian -- a ridge keeps the step finite, the bound stays valid at ANY point's value
        # only if it is the minimum, so iterate to convergence below)
        ridge = 1e-12 * (h11 + h22) + 1e-30
        det = (h11 + ridge) * (h22 + ridge) - h12 * h12
        d1 = (-(h22 + ridge) * g1 + h12 * g2) / det
        d2 = (-(h11 + ridge) * g2 + h12 * g1) / det
        lam2 = -(g1 * d1 + g2 * d2)
        if (lam2 < 1e-9).all():
            break
        step = np.ones(B)
        for _bt in range(40):
            un = u + step[:, None] * np.stack([d1, d2], axis=1)
            fn, qn = val(un)
            worse = ~(fn <= f + 1e-12 * np.abs(f))
            if not worse.any():
                break
            step = np.where(worse, step * 0.5, step)
        un = u + step[:, None] * np.stack([d1, d2], axis=1)
        fn, qn = val(un)
        take = fn <= f
        u = np.where(take[:, None], un, u)
        f = np.where(take, fn, f)
        q = np.where(take[:, None], qn, q)
    # lower bound of the minimum from the last point: f - lambda^2 (self-concordance, t < 1/2 by then; else no pruning: -inf)
    al, be = a / q, b / q
    g1, g2 = -(rd * al).sum(axis=1), -(rd * be).sum(axis=1)
    h11, h12, h22 = (rd * al * al).sum(axis=1), (rd * al * be).sum(axis=1), (rd * be * be).sum(axis=1)
    ridge = 1e-12 * (h11 + h22) + 1e-30
    det = (h11 + ridge) * (h22 + ridge) - h12 * h12
    lam2 = ((h22 + ridge) * g1 * g1 - 2 * h12 * g1 * g2 + (h11 + ridge) * g2 * g2) / det
    safe = np.where(lam2 < 0.25 * rd.min(), f - lam2, -np.inf)
    with np.errstate(divide="ignore", invalid="ignore"):
        freed = r[d:]
        const = Rp * np.log(Rtot / Rp) - (freed[freed > 0] * np.log(freed[freed > 0] / (Rtot * Nn[d:][freed > 0]))).sum()
    return K0 + safe + Rp * np.log(1.0) + const, u          # (z.w = 1 on the slice: the normalisation term vanishes)


def walk(r, Nn, K0, Rtot, K, thr, cap=3_000_000, Ct=None, verbose=False):
    """Depth by depth: every row tried on every live prefix, a prefix kept while its bound is <= thr.  Returns the complete
    matrices within thr (X, Y: (n, m)), their minima, the number of bound solves, and whether the walk reached depth m."""
    m = len(r)
    rows = np.array([(x, y) for x in range(K + 1) for y in range(K + 1)], dtype=float)
    X = np.zeros((1, 0))
    Y = np.zeros((1, 0))
    W = np.zeros((1, 2))
    solves = 0
    t0 = time.time()
    lbs = keep = None
    for d in range(1, m + 1):
        B = X.shape[0]
        Xc = np.concatenate([np.repeat(X, len(rows), axis=0), np.tile(rows[:, 0], B)[:, None]], axis=1)
        Yc = np.concatenate([np.repeat(Y, len(rows), axis=0), np.tile(rows[:, 1], B)[:, None]], axis=1)
        Wc = np.repeat(W, len(rows), axis=0)
        lbs = np.empty(Xc.shape[0])
        Us = np.empty((Xc.shape[0], 2))
        for s in range(0, Xc.shape[0], 200_000):
            lbs[s:s + 200_000], Us[s:s + 200_000] = bound(Xc[s:s + 200_000], Yc[s:s + 200_000], r, Nn, K0, Rtot, Wc[s:s + 200_000])
        solves += Xc.shape[0]
        keep = lbs <= thr
        if d == m:                                        # a complete matrix needs both tumour columns non-zero
            keep &= (Xc.sum(axis=1) > 0) & (Yc.sum(axis=1) > 0)
        X, Y, W = Xc[keep], Yc[keep], Us[keep]
        if Ct is not None:
            truth_alive = bool(((X == Ct[:d, 0]).all(axis=1) & (Y == Ct[:d, 1]).all(axis=1)).any())
            assert truth_alive, "the bound cut the incumbent: not a lower bound"
        if verbose:
            print("depth %2d: %9d prefixes bounded, %8d survive (%.3g of the (K+1)^(2d) = %.3g at this depth); %.1f s"
                  % (d, Xc.shape[0], X.shape[0], X.shape[0] / float(K + 1) ** (2 * d), float(K + 1) ** (2 * d), time.time() - t0))
        if X.shape[0] > cap:
            if verbose:
                print("more than %d live prefixes: stopping (the instance does not determine the matrix well enough at this depth)" % cap)
            return X, Y, lbs[keep], solves, False
    return X, Y, lbs[keep], solves, True


def main():
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 7
    cap = int(float(sys.argv[4])) if len(sys.argv) > 4 else 3_000_000
    r, rN, Ct, mu = synth_with_truth(seed, m, K)
    Nn = rN / rN.sum()
    Rtot = r.sum()
    K0 = -(r * np.log(Nn)).sum()
    # incumbent: the planted matrix's own optimum (depth m: the bound IS the candidate's minimum, const = 0)
    inc, _ = bound(Ct[None, :, 0].astype(float), Ct[None, :, 1].astype(float), r, Nn, K0, Rtot, np.zeros((1, 2)))
    # ... polished by coordinate descent (one interval's row at a time, until no row improves it): a search keeps its running
    # minimum up to date, and the planted matrix is several units above the best one -- against its value the "window" would be
    # that much wider
    Cb = Ct.astype(float).copy()
    best = float(inc[0])
    improved = True
    while improved:
        improved = False
        for i in range(m):
            cand = np.repeat(Cb[None], (K + 1) ** 2, axis=0)
            cand[:, i, 0] = np.repeat(np.arange(K + 1), K + 1)
            cand[:, i, 1] = np.tile(np.arange(K + 1), K + 1)
            v, _ = bound(cand[:, :, 0], cand[:, :, 1], r, Nn, K0, Rtot, np.zeros((len(cand), 2)))
            v = np.where((cand[:, :, 0].sum(axis=1) > 0) & (cand[:, :, 1].sum(axis=1) > 0), v, np.inf)
            j = int(np.argmin(v))
            if v[j] < best - 1e-9:
                best, Cb, improved = float(v[j]), cand[j].copy(), True
    print("incumbent: planted %.4f -> polished %.4f (%d rows changed)" % (inc[0], best, int((Cb != Ct).any(axis=1).sum())))
    Ct = Cb.astype(int)
    inc = np.array([best])
    thr = best + WINDOW
    print("instance: m=%d K=%d seed=%d  Rtot=%.3g  planted NLL %.4f  threshold %.4f  space (K+1)^(2m) = %.3g matrices"
          % (m, K, seed, Rtot, inc[0], thr, float(K + 1) ** (2 * m)))
    X, Y, vals, solves, done = walk(r, Nn, K0, Rtot, K, thr, cap, Ct, True)
    if done:
        print("complete matrices within the window: %d (best %.4f, planted %.4f); %d bound solves in total against %.3g matrices"
              % (X.shape[0], vals.min(), inc[0], solves, float(K + 1) ** (2 * m)))


if __name__ == "__main__":
    main()
