"""tools/bnb_prototype.py (the plan of DESIGN.md section 8: the relaxed prefix bound applied depth by depth) against brute force on
spaces small enough to enumerate: the walk must return EXACTLY the complete matrices whose minimum lies within the threshold -- no
matrix cut by a bound that is not one -- while bounding a small fraction of the space."""
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bnb_prototype as bnb


def _case(seed, m, K, window):
    r, rN, Ct, mu = bnb.synth_with_truth(seed, m, K)
    Nn = rN / rN.sum()
    Rtot = r.sum()
    K0 = -(r * np.log(Nn)).sum()
    rows = np.array(list(itertools.product(range(K + 1), repeat=2)), dtype=float)
    idx = np.array(list(itertools.product(range(len(rows)), repeat=m)))
    X, Y = rows[idx, 0], rows[idx, 1]
    vals = np.empty(len(X))
    for s in range(0, len(X), 100_000):
        vals[s:s + 100_000], _ = bnb.bound(X[s:s + 100_000], Y[s:s + 100_000], r, Nn, K0, Rtot, np.zeros((len(X[s:s + 100_000]), 2)))
    ok = (X.sum(axis=1) > 0) & (Y.sum(axis=1) > 0)
    vals = np.where(ok, vals, np.inf)
    thr = vals.min() + window
    want = set(map(tuple, np.concatenate([X, Y], axis=1)[vals <= thr - 1e-6].astype(int)))        # (1e-6: the walk's f - lambda^2 slack at depth m)
    gx, gy, gv, solves, done = bnb.walk(r, Nn, K0, Rtot, K, thr)
    assert done
    got = set(map(tuple, np.concatenate([gx, gy], axis=1).astype(int)))
    return want, got, solves, len(X), vals.min(), gv


def test_the_walk_returns_what_brute_force_returns():
    for seed, m, K, window in ((3, 5, 2, 0.5), (4, 6, 2, 5.0), (5, 4, 3, 50.0)):
        want, got, solves, total, vmin, gv = _case(seed, m, K, window)
        assert want <= got, (seed, m, K, sorted(want - got)[:3])                  # nothing within the threshold is cut
        assert len(got) <= len(want) + 8                                          # ... and what is kept beyond it sits AT the threshold (slack 1e-6)
        assert abs(gv.min() - vmin) <= 1e-6 * max(1.0, abs(vmin)) * 1e-3
        assert len(want) >= 1
        if m >= 6:
            assert solves < total                                                 # (a toy: pruning starts at depth 4; the large cases are in profiles/r4/NOTES.md)
