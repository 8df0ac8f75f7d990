"""
The bound behind the sieve's prefix pruning (theta_amd/csrc/n3_sieve.hip: sv_prefix_beyond), on the CPU:
with the rows of a prefix fixed and every other interval l fitted perfectly (its term a free t_l > 0), the t_l minimise out and

    min over the completions  >=  min_w [K0 - sum' R_i ln q_i + R' ln(z'.w)] + R' ln(Rtot / R') - sum_l R_l ln(R_l / (Rtot N_l))

-- the likelihood of the prefix alone plus a constant of the problem.  Checked against the minimum (scipy, several starts) of
EVERY completion of random prefixes of a small seeded instance, in the likelihood the search kernels use (q_i = w0 + x_i u1 + y_i u2,
NLL = K0 - sum R ln q + Rtot ln z.w: Optimizer.py:273-330 with the multiplier eliminated).
"""
import itertools

import numpy as np
from scipy.optimize import minimize


def _instance(seed, m, K):
    rng = np.random.RandomState(seed)
    L = rng.randint(2_000_000, 20_000_000, m)
    rN = np.maximum(rng.poisson(L * 0.01), 1).astype(float)
    Ct = rng.randint(0, K + 1, (m, 2))
    mu = np.array([0.3, 0.45, 0.25])
    p = rN * (2 * mu[0] + Ct @ mu[1:])
    r = rng.multinomial(int(rN.sum() * 1.1), p / p.sum()).astype(float)
    return rng, r, rN


def _min(f):
    return min(minimize(f, s, method="Nelder-Mead", options={"xatol": 1e-11, "fatol": 1e-11, "maxiter": 6000}).fun
               for s in ([.2, .2], [.05, .5], [.5, .05]))


def test_the_relaxed_prefix_bound_lies_below_the_minimum_of_every_completion():
    m, free, K = 9, 2, 2
    rng, r, rN = _instance(5, m, K)
    Nn = rN / rN.sum()
    Rtot = r.sum()
    K0 = -(r * np.log(Nn)).sum()
    D = m - free
    slack = []
    for _trial in range(4):
        pre = rng.randint(0, K + 1, (D, 2))
        pre[0], pre[1] = [1, 2], [2, 0]                      # (both tumour columns non-zero, rows not on one line)
        Rp, Nrem = r[:D].sum(), Nn[D:].sum()
        zred = np.array([1 - Nrem, (Nn[:D] * pre[:, 0]).sum(), (Nn[:D] * pre[:, 1]).sum()])

        def reduced(v):
            w = np.array([1.0, v[0], v[1]])
            q = w[0] + pre[:, 0] * w[1] + pre[:, 1] * w[2]
            if (q <= 0).any() or zred @ w <= 0:
                return 1e300
            return K0 - (r[:D] * np.log(q)).sum() + Rp * np.log(zred @ w)
        bound = _min(reduced) + Rp * np.log(Rtot / Rp) - sum(r[l] * np.log(r[l] / (Rtot * Nn[l])) for l in range(D, m) if r[l] > 0)
        lowest = np.inf
        for rows in itertools.product(range(K + 1), repeat=2 * free):
            C = np.vstack([pre, np.array(rows).reshape(free, 2)]).astype(float)
            z = np.array([1.0, (Nn * C[:, 0]).sum(), (Nn * C[:, 1]).sum()])

            def nll(v):
                w = np.array([1.0, v[0], v[1]])
                q = w[0] + C[:, 0] * w[1] + C[:, 1] * w[2]
                if (q <= 0).any() or z @ w <= 0:
                    return 1e300
                return K0 - (r * np.log(q)).sum() + Rtot * np.log(z @ w)
            lowest = min(lowest, _min(nll))
        assert bound <= lowest + 1e-6 * abs(lowest), (bound, lowest)
        slack.append(lowest - bound)
    assert max(slack) > 1.0          # (a bound, not the minimum itself: the freed intervals do cost something)
