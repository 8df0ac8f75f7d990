"""
`L2` / `L3`: drop-ins for the fork's vectorised scorers (python/CalcAllC.py:44-76), computed by the
`theta_score_batch` kernel.  Same signatures, same ValueError on shape mismatch, same 'X' markers,
and -- like the reference -- `L2` rescales its argument C in place (CalcAllC.py:54-55).
`L2_many` / `L3_many` score a batch of matrices in one launch.
"""
import numpy as np

from . import _lib


def _vals(v, ok):
    return [float(x) if o else 'X' for x, o in zip(v, ok)]


def L2_many(mus, Cs, m, r, ctx=None):
    Cs = np.asarray(Cs, dtype=np.float64)
    if m != Cs.shape[1]:
        raise ValueError('m not equal to first dimension of C')
    B = Cs.shape[0]
    mu2 = np.zeros((B, 2))
    mu2[:, 0] = mus
    mu2[:, 1] = 1 - np.asarray(mus, dtype=np.float64)
    nll, vals, valid = (ctx or _lib.default_context()).score_batch(2, Cs[:, :, :2], mu2, r)
    return [(float(nll[b]), _vals(vals[b], valid[b])) for b in range(B)]


def L3_many(mus, Cs, m, r, n, ctx=None):
    Cs = np.asarray(Cs, dtype=np.float64)
    if m != Cs.shape[1]:
        raise ValueError('m not equal to first dimension of C')
    if n != Cs.shape[2]:
        raise ValueError('n not equal to second dimension of C')
    nll, vals, valid = (ctx or _lib.default_context()).score_batch(n, Cs, np.asarray(mus, dtype=np.float64), r)
    return [(float(nll[b]), _vals(vals[b], valid[b])) for b in range(Cs.shape[0])]


def L2(mu, C, m, r):
    """CalcAllC.py:44-61."""
    if m != C.shape[0]:
        raise ValueError('m not equal to first dimension of C')
    out = L2_many([mu], np.asarray(C, dtype=np.float64)[None, :, :], m, r)[0]
    C[:, 0] = C[:, 0] * mu            # the reference's in-place scaling (quirk Q7)
    C[:, 1] = C[:, 1] * (1 - mu)
    return out


def L3(mu, C, m, r, n):
    """CalcAllC.py:63-76."""
    if m != C.shape[0]:
        raise ValueError('m not equal to first dimension of C')
    if n != C.shape[1]:
        raise ValueError('n not equal to second dimension of C')
    return L3_many([mu], np.asarray(C, dtype=np.float64)[None, :, :], m, r, n)[0]
