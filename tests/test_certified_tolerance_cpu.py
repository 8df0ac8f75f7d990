"""The certificate behind bench.py's leg full_solve_f64_tight_certified (CPU, numpy).

A candidate's restricted likelihood in the sieve kernel (theta_amd/csrc/n3_sieve.hip, sv_step) is
    f(u) = - sum_i R_i log q_i(u),   q_i(u) = 1 + a_i u1 + b_i u2,
the reference's L3 (Optimizer.py:273-330) on the slice where its normalisation is constant.  f / Rmin is a sum of -log terms with
coefficients >= 1, hence standard self-concordant: with lambda the Newton decrement and t = lambda / sqrt(Rmin) < 1, a FULL Newton
step ends at t' <= (t / (1 - t))^2.  In the kernel's unit l2 = lambda^2 / sum R that reads
    l2' <= l2^2 (sum R / Rmin) / (1 - t)^4 <= 1.53 l2^2 sum R / Rmin          for t <= 0.1,
which is what bench.certified_conv_l2 inverts.  Checked here on random problems of the kernel's shape, at points whose decrement
spans the range the leg uses."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench


def _problem(rng):
    T = int(rng.integers(8, 30))
    rmin = float(rng.integers(50, 30000))
    R = np.floor(rmin * (1.0 + 40.0 * rng.random(T) ** 2))
    R[int(rng.integers(T))] = rmin
    x = rng.integers(0, 7, T).astype(float)
    y = rng.integers(0, 7, T).astype(float)
    if np.linalg.matrix_rank(np.stack([np.ones(T), x, y])) < 3:
        return None
    s1, s2 = float((R * x).sum() / R.sum()) + 0.3 * rng.random(), float((R * y).sum() / R.sum()) + 0.3 * rng.random()
    return R, x - s1, y - s2


def _eval(R, a, b, u):
    q = 1.0 + a * u[0] + b * u[1]
    if not (q > 0).all():
        return None
    al, be = a / q, b / q
    g = -np.array([(R * al).sum(), (R * be).sum()])
    H = np.array([[(R * al * al).sum(), (R * al * be).sum()], [(R * al * be).sum(), (R * be * be).sum()]])
    d = np.linalg.solve(H, -g)
    return float(-(g @ d)) / R.sum(), d          # (lambda^2 / sum R, the Newton step)


def test_a_full_newton_step_ends_below_the_self_concordance_bound():
    rng = np.random.default_rng(20260930)
    checked = 0
    worst = 0.0
    while checked < 4000:
        pr = _problem(rng)
        if pr is None:
            continue
        R, a, b = pr
        ror = R.sum() / R.min()
        u = np.zeros(2)
        ok = True
        for _ in range(60):                        # the minimiser, by damped Newton from the centre of the slice
            ev = _eval(R, a, b, u)
            if ev is None:
                ok = False
                break
            l2, d = ev
            if l2 < 1e-30:
                break
            step = 1.0 if l2 * ror < 0.25 else 1.0 / (1.0 + np.sqrt(l2 * ror))
            while _eval(R, a, b, u + step * d) is None:
                step *= 0.5
            u = u + step * d
        if not ok:
            continue
        ustar = u
        for _ in range(8):
            # a point at a chosen distance from the minimiser: decrements from the leg's threshold range up to t = 0.1
            scale = 10.0 ** rng.uniform(-5.5, -1.0)
            p = ustar + scale * rng.standard_normal(2) / np.sqrt(R.sum())
            ev = _eval(R, a, b, p)
            if ev is None:
                continue
            l2, d = ev
            if not (l2 * ror <= 0.01) or l2 < 1e-13:
                continue
            ev2 = _eval(R, a, b, p + d)
            assert ev2 is not None, "a full step from t <= 0.1 stays in the domain"
            bound = 1.53 * ror * l2 * l2
            assert ev2[0] <= bound * (1 + 1e-6) + 1e-27, (l2, ev2[0], bound, ror)
            worst = max(worst, ev2[0] / bound)
            checked += 1
    assert 0.0 < worst <= 1.0 + 1e-6


def test_the_legs_threshold_certifies_the_tight_tolerance():
    r, rN, order = bench.synth()
    r = np.asarray(r, dtype=np.float64)
    conv = bench.certified_conv_l2(r)
    ror = r.sum() / r.min()
    assert conv * ror <= 0.01 + 1e-15                       # t <= 0.1 at every evaluation that passes
    assert 1.53 * ror * conv * conv <= 1e-12 * (1 + 1e-12)  # ... and the step from it ends below the tight tolerance
    assert 1e-9 < conv < 1e-4                               # between the tight and the coarse tolerance of the other legs
    # a problem whose smallest weight is tiny: the t <= 0.1 side binds
    assert bench.certified_conv_l2(np.array([1.0, 1e9])) == 0.01 / (1e9 + 1.0)


# ---- round 6: the tolerance on MU as a certificate (n3_sieve.hip: sv_mu_limit; option "n3_mu_tol") ------------------------------
def mu_of(u, s1, s2, tau):
    """nu -> mu as the kernels do it (M3's closed form, Optimizer.py:318-330): u0 = (1 - s1 u1 - s2 u2) / tau, mu = (u0, u1, u2) / sum"""
    u0 = (1.0 - s1 * u[0] - s2 * u[1]) / tau
    v = np.array([u0, u[0], u[1]])
    return v / v.sum()


def mu_bound(l2, R, H, u, s1, s2, tau):
    """The certificate's bound on |mu(u+) - mu(u*)|_inf, u+ = u + Newton step, from what the evaluation at u has at hand: with
    t = lambda / sqrt(Rmin) <= 0.1 the step ends at lambda+^2 <= lambda^4 / (Rmin (1 - t)^4), the minimiser lies within
    lambda+ / (1 - t+) of u+ in the norm of H(u+) >= (1 - t)^2 H(u), whose smaller eigenvalue is 2 det / (tr + sqrt(tr^2 - 4 det)) >= det / tr;
    d mu_j = (du_j - mu_j (k . du)) / U, k = (1 - s1 / tau, 1 - s2 / tau), U = u0 + u1 + u2, so
    |d mu|_inf <= |du|_2 (1.5 + max(1, |mu1| + |mu2|) |k|_2) / U.  (n3_sieve.hip: sv_mu_limit is this bound solved for l2.)"""
    Rtot, Rmin = R.sum(), R.min()
    tr, det = np.trace(H), np.linalg.det(H)
    sig = det / tr                  # (<= the smaller eigenvalue 2 det / (tr + sqrt(tr^2 - 4 det)): what sv_mu_limit takes)
    U = (1.0 - s1 * u[0] - s2 * u[1]) / tau + u[0] + u[1]
    M = max(1.0, (abs(u[0]) + abs(u[1])) / U)
    J = (1.5 + M * np.hypot(1.0 - s1 / tau, 1.0 - s2 / tau)) / U
    t = np.sqrt(l2 * Rtot / Rmin)
    tp = (t / (1.0 - t)) ** 2
    return 1.01 * l2 * Rtot * J / ((1.0 - t) ** 3 * (1.0 - tp) * np.sqrt(Rmin * sig))


def test_the_mu_certificate_bounds_the_distance_to_the_optimum():
    """What "n3_mu_tol" builds on: the bound above really bounds the distance in mu between the point a candidate is left at (one
    full Newton step beyond the evaluation) and its optimum -- on random problems of the kernel's shape, at points inside the
    simplex whose decrement spans the leg's range."""
    rng = np.random.default_rng(20260931)
    checked, worst = 0, 0.0
    tau = 2.0
    while checked < 3000:
        T = int(rng.integers(8, 30))
        rmin = float(rng.integers(50, 30000))
        R = np.floor(rmin * (1.0 + 40.0 * rng.random(T) ** 2))
        R[int(rng.integers(T))] = rmin
        x = rng.integers(0, 7, T).astype(float)
        y = rng.integers(0, 7, T).astype(float)
        if np.linalg.matrix_rank(np.stack([np.ones(T), x, y])) < 3:
            continue
        w = rng.integers(1, 50, T).astype(float)              # normal counts behind the column sums
        s1, s2 = float((w * x).sum() / w.sum()), float((w * y).sum() / w.sum())
        if not (s1 > 0 and s2 > 0):
            continue
        a, b = x - s1, y - s2

        def ev(u):
            q = 1.0 + a * u[0] + b * u[1]
            if not (q > 0).all():
                return None
            al, be = a / q, b / q
            g = -np.array([(R * al).sum(), (R * be).sum()])
            H = np.array([[(R * al * al).sum(), (R * al * be).sum()], [(R * al * be).sum(), (R * be * be).sum()]])
            d = np.linalg.solve(H, -g)
            return float(-(g @ d)) / R.sum(), d, H
        u = np.array([(1.0 / 3.0) / s1, (1.0 / 3.0) / s2])
        ok = True
        for _ in range(80):
            e = ev(u)
            if e is None:
                ok = False
                break
            l2, d, _H = e
            if l2 < 1e-31:
                break
            step = 1.0 if l2 * R.sum() / R.min() < 0.25 else 1.0 / (1.0 + np.sqrt(l2 * R.sum() / R.min()))
            while ev(u + step * d) is None:
                step *= 0.5
            u = u + step * d
        if not ok:
            continue
        ustar = u
        nu = np.array([1 - s1 * u[0] - s2 * u[1], s1 * u[0], s2 * u[1]])
        if nu.min() < -0.03:                                   # (the certificate is for optima in -- or just outside -- the simplex: the ones the reference reports)
            continue
        for _ in range(6):
            p = ustar + 10.0 ** rng.uniform(-5.0, -1.5) * rng.standard_normal(2) / np.sqrt(R.sum())
            e = ev(p)
            if e is None:
                continue
            l2, d, H = e
            if not (l2 * R.sum() / R.min() <= 0.01) or l2 < 1e-13:
                continue
            bound = mu_bound(l2, R, H, p, s1, s2, tau)
            dist = np.abs(mu_of(p + d, s1, s2, tau) - mu_of(ustar, s1, s2, tau)).max()
            assert dist <= bound + 1e-15, (dist, bound, l2)
            worst = max(worst, dist / bound)
            checked += 1
    assert 0.0 < worst <= 1.0


# ---- round 6: the cubic correction of the tight modes' shared step (n3_sieve.hip: sv_parent_third, sv_child_eval_third) ------------
def _shared_step(R, x, y, N, w, dtype, cubic):
    """A child's shared step as the kernel takes it: sums T, W (and the third-order V) over the candidate's terms at the ROUND's point
    w = (w0, u1, u2) in z = (1, x, y) coordinates, restricted to the child's slice z_bar . w = const (z_bar = (1, s1, s2), the
    N-weighted column sums), Newton's step d = H^-1 G plus -- `cubic` -- H^-1 c with c_j = V[D, D]_j - s_j V[D, D]_0 for
    D = (-s . d, d); returned in the child's own coordinates u = w[1:] / (z_bar . w).  `dtype`: the arithmetic of the sums and of the solve."""
    f = dtype
    R, x, y = R.astype(f), x.astype(f), y.astype(f)
    s1, s2 = f((N * x).sum() / N.sum()), f((N * y).sum() / N.sum())
    w0, u1, u2 = (f(v) for v in w)
    Z = np.stack([np.ones_like(x), x, y], axis=1)
    q = w0 + x * u1 + y * u2
    T = (R / q) @ Z
    W = (Z * (R / q ** 2)[:, None]).T @ Z
    G = np.array([T[1] - s1 * T[0], T[2] - s2 * T[0]], dtype=f)
    A1, A2 = W[0, 1] - s1 * W[0, 0], W[0, 2] - s2 * W[0, 0]
    H11 = -s1 * A1 + (-s1 * W[0, 1] + W[1, 1])
    H12 = -s2 * A1 + (-s1 * W[0, 2] + W[1, 2])
    H22 = -s2 * A2 + (-s2 * W[0, 2] + W[2, 2])
    det = H11 * H22 - H12 * H12
    d = np.array([(H22 * G[0] - H12 * G[1]) / det, (H11 * G[1] - H12 * G[0]) / det], dtype=f)
    if cubic:
        V = np.einsum("i,ia,ib,ic->abc", R / q ** 3, Z, Z, Z).astype(f)
        D = np.array([-s1 * d[0] - s2 * d[1], d[0], d[1]], dtype=f)
        o = np.einsum("abc,b,c->a", V, D, D)
        c1, c2 = o[1] - s1 * o[0], o[2] - s2 * o[0]
        k = np.array([(H22 * c1 - H12 * c2) / det, (H11 * c2 - H12 * c1) / det], dtype=f)
        if abs(k).sum() < 0.5 * abs(d).sum():
            d = d + k
    zw = w0 + s1 * u1 + s2 * u2
    return (np.array([u1, u2], dtype=np.float64) + d.astype(np.float64)) / float(zw), float(s1), float(s2)


def test_the_cubic_correction_of_the_shared_step_cubes_the_decrement():
    """What the tight modes' 2.07 evaluations per candidate rest on: from a point shared by a round's children (here: the optimum of a
    NEIGHBOUR, a candidate that differs in its last rows), Newton's step leaves the first private evaluation a decrement ~l2^2, the step
    with the cubic correction ~l2^3 -- and single precision in the shared sums and the 2x2 solve costs nothing against either (the step is
    a starting point; the evaluation that certifies and values the child is FP64)."""
    rng = np.random.default_rng(20260930)
    ratios, f32_loss, newton, cubicl = [], [], [], []
    while len(ratios) < 400:
        T = int(rng.integers(14, 26))
        R = np.floor(float(rng.integers(2000, 30000)) * (1.0 + 8.0 * rng.random(T) ** 2))
        N = np.floor(R * (0.6 + 0.8 * rng.random(T)))
        x = rng.integers(0, 7, T).astype(float)
        y = rng.integers(0, 7, T).astype(float)
        if np.linalg.matrix_rank(np.stack([np.ones(T), x, y])) < 3:
            continue

        def solve(xx, yy):
            s1, s2 = (N * xx).sum() / N.sum(), (N * yy).sum() / N.sum()
            if not (s1 > 0 and s2 > 0):
                return None
            a, b = xx - s1, yy - s2
            u = np.array([(1.0 / 3.0) / s1, (1.0 / 3.0) / s2])
            for _ in range(100):
                e = _eval(R, a, b, u)
                if e is None:
                    return None
                l2, d = e
                if l2 < 1e-30:
                    return u, s1, s2
                st = 1.0 if l2 * R.sum() / R.min() < 0.25 else 1.0 / (1.0 + np.sqrt(l2 * R.sum() / R.min()))
                while _eval(R, a, b, u + st * d) is None:
                    st *= 0.5
                u = u + st * d
            return None
        # the neighbour: the last three rows drawn again; its optimum, as a point w with z_bar . w = 1, is the round's shared point
        xn, yn = x.copy(), y.copy()
        xn[-3:] = rng.integers(0, 7, 3)
        yn[-3:] = rng.integers(0, 7, 3)
        nb = solve(xn, yn)
        me = solve(x, y)
        if nb is None or me is None:
            continue
        un, s1n, s2n = nb
        w = np.array([1.0 - s1n * un[0] - s2n * un[1], un[0], un[1]])
        if ((w[0] + x * w[1] + y * w[2]) <= 0).any():
            continue
        out = {}
        for name, dt, cub in (("newton", np.float64, False), ("cubic", np.float64, True), ("cubic32", np.float32, True)):
            u, s1, s2 = _shared_step(R, x, y, N, w, dt, cub)
            e = _eval(R, x - s1, y - s2, u)
            out[name] = None if e is None else e[0]
        e0 = _eval(R, x - me[1], y - me[2], w[1:] / (w[0] + me[1] * w[1] + me[2] * w[2]))
        if e0 is None or None in out.values() or not (1e-6 < e0[0] < 3e-3):
            continue
        ratios.append(out["cubic"] / out["newton"])
        f32_loss.append(out["cubic32"] / max(out["cubic"], 1e-300))
        newton.append(out["newton"])
        cubicl.append(out["cubic32"])
    ratios, newton, cubicl = np.array(ratios), np.array(newton), np.array(cubicl)
    assert np.median(ratios) < 0.1 and (ratios < 1.0).mean() > 0.97, (np.median(ratios), (ratios < 1.0).mean())
    # at the bench's certified threshold (4.4e-8) the corrected step passes where Newton's often does not
    assert (cubicl < 4.4e-8).mean() > (newton < 4.4e-8).mean() + 0.1, ((cubicl < 4.4e-8).mean(), (newton < 4.4e-8).mean())
    assert (cubicl < 4.4e-8).mean() > 0.9
    # ... and single precision leaves the decrement where FP64 leaves it, up to a floor far below the threshold
    assert np.median(f32_loss) < 1.5 and (cubicl < np.maximum(4.0 * np.array(ratios) * newton, 1e-11)).mean() > 0.95
