"""
Seeded random search instances and the comparison rule of the parity campaigns (shared by tools/parity_campaign.py,
tests/golden/make_golden_campaign.py and the -m gpu tests).  An instance is everything do_optimization_single takes.

Shapes: "toy" (n=2: m 4-13, k 2-5, tau 1-3, max_normal 0.5-1; n=3: m 4-7, k 2-3; ragged bounds with lb in {0,1}, so the
n=3 spaces hold matrices with an all-zero tumour column) and "mid" (m 10-18, k 3-5, bounds tight around a planted truth,
30 % of the n=3 instances with ONE tumour population -- the shape interval selection + the bounds heuristics produce) and
"low" (n=3 only: like "mid" with m 8-14, k 3-6, LOW coverage -- a few to a few hundred reads per interval, flat likelihoods --
and ONE tumour population in half of the instances: the spaces where near-dependent columns and unconverged solver runs live).
"""
import numpy as np


def _sort_r(rN, r):
    import theta_oracle as orc
    return orc.sort_r([int(x) for x in rN], [int(x) for x in r])


def instance_mid(seed, n):
    rng = np.random.RandomState(seed)
    m, k = int(rng.randint(10, 19)), int(rng.randint(3, 6))
    tau = 2
    L = rng.randint(2_000_000, 20_000_000, m)
    rN = np.maximum(rng.poisson(L * rng.choice([0.001, 0.004, 0.01])), 5)
    C = np.full((m, n), float(tau))
    for j in range(1, n):
        C[:, j] = rng.randint(0, k + 1, m)
    if n == 3 and rng.rand() < 0.3:
        C[:, 2] = C[:, 1]                       # a sample with ONE tumour population analysed with n=3
    mu = rng.dirichlet(np.ones(n) * 3)
    p = (C * rN[:, None]) @ mu
    p = p / p.sum()
    r = np.maximum(rng.multinomial(int(rN.sum() * rng.uniform(0.8, 1.5)), p), 1)
    rs, rNs, order = _sort_r(rN, r)
    cs = np.maximum(C[:, 1:].max(axis=1), 0)[order]
    cmin = C[:, 1:].min(axis=1)[order]
    free = rng.rand(m) < (0.85 if n == 2 else 0.3)
    lb = [int(max(0, a - (1 if f else 0))) for a, f in zip(cmin, free)]
    ub = [int(min(k, b + (1 if f else 0))) for b, f in zip(cs, free)]
    return dict(seed=seed, n=n, m=m, k=k, tau=tau, mx=1.0, r=rs, rN=rNs, order=order, lb=lb, ub=ub, shape="mid")


def instance_toy(seed, n):
    rng = np.random.RandomState(seed)
    if n == 2:
        m, k = int(rng.randint(4, 14)), int(rng.randint(2, 6))
    else:
        m, k = int(rng.randint(4, 8)), int(rng.randint(2, 4))
    tau = int(rng.choice([1, 2, 2, 2, 3])) if n == 2 else 2
    mx = float(rng.choice([1.0, 1.0, 0.5, 0.7])) if n == 2 else 1.0
    L = rng.randint(2_000_000, 20_000_000, m)
    rN = np.maximum(rng.poisson(L * rng.choice([0.0005, 0.004, 0.01])), 5)
    C = np.full((m, n), float(tau))
    for j in range(1, n):
        C[:, j] = rng.randint(0, k + 1, m)
    mu = rng.dirichlet(np.ones(n) * 3)
    p = (C * rN[:, None]) @ mu
    p = p / p.sum()
    r = np.maximum(rng.multinomial(int(rN.sum() * rng.uniform(0.8, 1.5)), p), 1)
    rs, rNs, order = _sort_r(rN, r)
    lb = [int(x) for x in rng.randint(0, 2, m)]
    ub = [int(x) for x in rng.randint(max(1, k - 1), k + 1, m)]
    return dict(seed=seed, n=n, m=m, k=k, tau=tau, mx=mx, r=rs, rN=rNs, order=order, lb=lb, ub=ub, shape="toy")


def instance_low(seed, n):
    rng = np.random.RandomState(seed)
    m, k = int(rng.randint(8, 15)), int(rng.randint(3, 7))
    tau = 2
    rN = np.maximum(rng.poisson(rng.choice([8, 40, 150, 400]) * rng.uniform(0.3, 1.7, m)), 3)
    C = np.full((m, n), float(tau))
    for j in range(1, n):
        C[:, j] = rng.randint(0, k + 1, m)
    if n == 3 and rng.rand() < 0.5:
        C[:, 2] = C[:, 1]
    mu = rng.dirichlet(np.ones(n) * 3)
    p = (C * rN[:, None]) @ mu
    p = p / p.sum()
    r = np.maximum(rng.multinomial(int(rN.sum() * rng.uniform(0.8, 1.5)), p), 1)
    rs, rNs, order = _sort_r(rN, r)
    cs = np.maximum(C[:, 1:].max(axis=1), 0)[order]
    cmin = C[:, 1:].min(axis=1)[order]
    free = rng.rand(m) < 0.35
    lb = [int(max(0, a - (1 if f else 0))) for a, f in zip(cmin, free)]
    ub = [int(min(k, b + (1 if f else 0))) for b, f in zip(cs, free)]
    return dict(seed=seed, n=n, m=m, k=k, tau=tau, mx=1.0, r=rs, rN=rNs, order=order, lb=lb, ub=ub, shape="low")


def instance_amp(seed, n=3, wide=False):
    """n=3 with the bounds the reference's OWN heuristic derives (calculate_bounds_heuristic, DataTools.py:47-67) from counts that
    hold one or two strongly amplified intervals: ub = max(k, y + 1), y = round(tau ratio), exceeds k there -- an interval at four
    times the normal ratio gets [7, 9] -- while the others keep [0, tau].  Copy numbers above 7: the compact row alphabet
    (csrc/n3_core.hpp)."""
    from theta_amd import DataTools as DT
    import contextlib
    import io
    rng = np.random.RandomState(seed)
    m, k, tau = (int(rng.randint(8, 11)) if wide else int(rng.randint(5, 8))), 3, 2       # (wide: m >= 8, the sieve path)
    rN = np.maximum(rng.poisson(rng.choice([300, 2000, 20000]) * rng.uniform(0.5, 1.5, m)), 20)
    ratio = rng.uniform(0.55, 1.15, m)
    amp = rng.choice(m, int(rng.choice([1, 1, 2])), replace=False)
    ratio[amp] = rng.uniform(5.5, 9.0, len(amp)) * ratio.mean()
    r = np.maximum(rng.poisson(rN * ratio), 1)
    rs, rNs, order = _sort_r(rN, r)
    DT.set_total_read_counts(sum(rs), sum(rNs))
    with contextlib.redirect_stdout(io.StringIO()):
        ub, lb = DT.calculate_bounds_heuristic(0.5, rs, rNs, m, tau, k)
    return dict(seed=seed, n=3, m=m, k=k, tau=tau, mx=1.0, r=rs, rN=rNs, order=order, lb=[int(v) for v in lb], ub=[int(v) for v in ub],
                shape="ampw" if wide else "amp")


def instance(seed, n, shape="toy"):
    if shape in ("amp", "ampw"):
        return instance_amp(seed, n, wide=shape == "ampw")
    return instance_mid(seed, n) if shape == "mid" else instance_low(seed, n) if shape == "low" else instance_toy(seed, n)


def count_candidates(inst):
    """Exact size of the instance's candidate space (the oracle's counting DP; no GPU needed)."""
    import theta_oracle as orc
    if inst["n"] == 2:
        return orc.count_n2(inst["m"], list(inst["lb"]), list(inst["ub"]))
    return orc.count_n3_exact(inst["m"], inst["tau"], list(inst["lb"]), list(inst["ub"]))


def best_to_plain(best):
    """`best` as nested lists: [(C rows, mu, nll)], NaN kept as float('nan')."""
    return [(np.asarray(b[0]).tolist(), [float(x) for x in b[1]], float(b[2])) for b in best]


def compare_best(got, ref, tol=1e-6):
    """
    The parity bar of BASELINE.json on two `best` lists [(C, mu, nll)], COMPLETE lists, entry by entry, in order: same
    number of entries; chosen C identical (bit-exact); NLL within `tol` relative -- or NaN on both sides (the entries the
    reference appends through isClose(NaN), Misc.py:44-46); |d mu| < tol where mu is determined (a matrix whose columns
    [1, x, y] are linearly dependent has a LINE of minimisers, and a NaN entry carries rounding residue).
    Returns "" or the reason of the first disagreement.
    """
    if len(got) != len(ref):
        return "lengths differ: %d vs %d" % (len(got), len(ref))
    for i, (x, y) in enumerate(zip(got, ref)):
        if x[0] != y[0]:
            return "entry %d: C differs" % i
        xn, yn = x[2] != x[2], y[2] != y[2]
        if xn != yn:
            return "entry %d: NaN on one side only (%r vs %r)" % (i, x[2], y[2])
        if xn:
            continue
        if abs(x[2] - y[2]) > tol * abs(y[2]):
            return "entry %d: NLL %r vs %r" % (i, x[2], y[2])
        Cm = np.array(x[0], dtype=float)
        if Cm.shape[1] == 3:
            if Cm[:, 1].sum() == 0 or Cm[:, 2].sum() == 0:
                continue        # all-zero column: mu is a unit vector plus residue; the NLL above pins what matters
            if np.linalg.matrix_rank(np.column_stack([np.ones(len(Cm)), Cm[:, 1], Cm[:, 2]])) < 3:
                continue
        if max(abs(u - v) for u, v in zip(x[1], y[1])) >= tol:
            return "entry %d: mu %r vs %r" % (i, x[1], y[1])
    return ""
