"""
Drop-in for the reference's `Optimizer` class (python/Optimizer.py:41-165): `Optimizer(r, rN, m, n,
tau, lower_bound=0, upper_bound=1).solve(C)` -> `(mu, NLL, vals)` or `None`, computed by the
`theta_solve_batch` kernel (per-interval sums in the reference's order; brenth iteration for n=2).
`solve_many` is the batched form the GPU actually wants.
"""
import numpy

from . import _lib


class Optimizer:
    def __init__(self, r, rN, m, n, tau, lower_bound=0, upper_bound=1, ctx=None):
        if lower_bound != 0:
            raise ValueError("the reference never passes a lower bound (RunTHetA.py:135,182); only 0 is supported")
        self.r = [int(x) for x in r]
        self.rN = [int(x) for x in rN]
        self.m, self.n, self.tau = m, n, tau
        self.lB, self.uB = lower_bound, upper_bound
        self._ctx = ctx or _lib.default_context()

    def solve_many(self, Cs):
        """Cs: (B, m, n) float or int matrices with column 0 == tau.  Returns a list like solve()."""
        Cs = numpy.asarray(Cs)
        if Cs.ndim != 3 or Cs.shape[1] != self.m or Cs.shape[2] != self.n:
            raise ValueError("expected candidates of shape (B, %d, %d)" % (self.m, self.n))
        tum = Cs[:, :, 1:]
        if (tum < 0).any() or (tum > 255).any() or (tum != numpy.floor(tum)).any():
            raise ValueError("copy numbers must be integers in [0, 255]")
        u8 = tum.astype(numpy.uint8)
        if self.n == 2:
            u8 = u8[:, :, 0]
        ok, mu, nll, vals = self._ctx.solve_batch(self.n, self.tau, self.r, self.rN, u8, self.uB, want_vals=True)
        out = []
        for b in range(len(ok)):
            if not ok[b]:
                out.append(None)
            elif self.n == 2:
                out.append(((float(mu[b, 0]), float(mu[b, 1])), float(nll[b]), [float(v) for v in vals[b]]))
            else:
                out.append((mu[b].copy(), float(nll[b]), [float(v) for v in vals[b]]))
        return out

    def solve(self, C):
        """Optimizer.py:68-88."""
        return self.solve_many(numpy.asarray(C)[None, :, :])[0]
