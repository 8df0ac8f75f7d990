"""
Test infrastructure for the witness tests (tests/test_gpu_round5.py): exact numpy evaluation of a candidate's likelihood at a given
mixture -- value and Newton decrement on the simplex -- in the reference's own variables (Optimizer.py:167-182, 236-244, 273-311).
Nothing here is on the product path.
"""
import numpy as np


def nu_from_mu(C, rN, mu, tau):
    """Inverse of Optimizer.M3's closed form (Optimizer.py:318-330): mu_j ~ nu_j / S_j with S_j = sum_i rN_i C_ij.
    C: (B, m, 2) tumour columns; mu: (B, 3).  Returns nu (B, 3), Chat (B, m, 3)."""
    C = np.asarray(C, np.float64)
    rN = np.asarray(rN, np.float64)
    B, m, _ = C.shape
    W = np.empty((B, m, 3))
    W[:, :, 0] = tau * rN[None, :]
    W[:, :, 1:] = C * rN[None, :, None]
    S = W.sum(axis=1)                                   # (B, 3)
    with np.errstate(divide="ignore", invalid="ignore"):
        Chat = W / S[:, None, :]
    nu = np.asarray(mu, np.float64) * S
    nu = nu / nu.sum(axis=1, keepdims=True)
    return nu, Chat


def value_and_decrement(C, r, rN, mu, tau=2):
    """NLL(mu) = -sum r_i ln p_i (Optimizer.L3's value) and lambda^2 / sum r at that point, lambda the Newton decrement of the NLL
    restricted to the simplex sum nu = 1 (affine invariant: the same number in the kernel's (u1, u2) chart).  Vectorised over B."""
    r = np.asarray(r, np.float64)
    nu, Chat = nu_from_mu(C, rN, mu, tau)
    p = np.einsum("bij,bj->bi", Chat, nu)
    with np.errstate(divide="ignore", invalid="ignore"):
        nll = -(r[None, :] * np.log(p)).sum(axis=1)
        t = r[None, :] / p
        g = -np.einsum("bi,bij->bj", t, Chat)                     # d NLL / d nu_j
        H = np.einsum("bi,bij,bik->bjk", t / p, Chat, Chat)
    Bm = np.array([[-1.0, 1.0, 0.0], [-1.0, 0.0, 1.0]])          # tangent basis of the simplex
    gt = np.einsum("tj,bj->bt", Bm, g)
    Ht = np.einsum("tj,bjk,sk->bts", Bm, H, Bm)
    det = Ht[:, 0, 0] * Ht[:, 1, 1] - Ht[:, 0, 1] ** 2
    with np.errstate(divide="ignore", invalid="ignore"):
        d0 = (Ht[:, 1, 1] * gt[:, 0] - Ht[:, 0, 1] * gt[:, 1]) / det
        d1 = (Ht[:, 0, 0] * gt[:, 1] - Ht[:, 0, 1] * gt[:, 0]) / det
    lam2 = gt[:, 0] * d0 + gt[:, 1] * d1
    return nll, lam2 / r.sum()
