"""
Drop-in for the reference's `Enumerator` class (python/Enumerator.py:38-298), backed by the
materialised generator kernel `theta_enumerate`: same constructor, `generate_next_C()` returning a
fresh float64 (m, n) matrix with column 0 == tau or `False` at the end, `_C_to_array()`.
"""
import numpy

from . import _lib

_CHUNK = 8192


class Enumerator:
    def __init__(self, n, m, k, tau, lower_bound=None, upper_bound=None, multi_event=False, ctx=None):
        self.m = m
        self.n = n - 1                      # Enumerator.py:52: number of generated columns
        self.tau = tau
        self.allow_multi_event = True       # Enumerator.py:55: the ctor argument is dead in the reference
        if lower_bound is None or upper_bound is None:
            raise TypeError("bounds are required (the reference fails on max(None), Enumerator.py:58)")
        # Enumerator.py:57,90-113: the reference adjusts the caller's lists IN PLACE
        for i in range(1, len(lower_bound)):
            if lower_bound[i] < lower_bound[i - 1]:
                lower_bound[i] = lower_bound[i - 1]
        for i in reversed(range(len(upper_bound) - 1)):
            if upper_bound[i] > upper_bound[i + 1]:
                upper_bound[i] = upper_bound[i + 1]
        self.lower_bound, self.upper_bound = lower_bound, upper_bound
        self.k = max(self.upper_bound)
        self.iter = list(self.lower_bound) if n == 2 else [0] * m
        ctx = ctx or _lib.default_context()
        ones = numpy.ones(m, dtype=numpy.int64)
        self._problem = _lib.Problem(ctx, n, m, tau, ones, ones, [int(v) for v in lower_bound],
                                     [int(v) for v in upper_bound], 1.0)
        self.count = self._problem.count
        self._next = 0
        self._buf = None
        self._buf_base = 0

    def _row(self, rank):
        if self._buf is None or not (self._buf_base <= rank < self._buf_base + len(self._buf)):
            cnt = min(_CHUNK, self.count - rank)
            self._buf = self._problem.enumerate(rank, cnt)
            self._buf_base = rank
        return self._buf[rank - self._buf_base]

    def generate_next_C(self):
        """Enumerator.py:74-87."""
        if self._next >= self.count:
            return False
        c = self._row(self._next)
        self._next += 1
        C = numpy.zeros((self.m, self.n + 1))
        C[:, 0] = self.tau
        if self.n == 1:
            C[:, 1] = c
            self.iter = [int(v) for v in c]
        else:
            C[:, 1:] = c
        return C

    def get_graph(self):
        """Enumerator.py:166-170, 272-298 (n=3; the reference's time estimate reads it): the row alphabet -- every (a, b) in
        [0..k]^2 with (tau - a)(tau - b) >= 0, a fastest -- and for each row the rows that may follow it (the same row, or one
        with a larger component)."""
        if self.n != 2:
            raise AttributeError("'Enumerator' object has no attribute 'rows'")      # (the reference builds the graph for n=3 only)
        rows = [[a, b] for b in range(self.k + 1) for a in range(self.k + 1) if (self.tau - a) * (self.tau - b) >= 0]
        edges = [[j for j, w in enumerate(rows) if (v == w or w[0] > v[0] or w[1] > v[1])] for v in rows]
        return rows, edges

    def _C_to_array(self):
        """Enumerator.py:154-160 (for n=3 this is the [tau,0,0] matrix whatever the bounds, quirk Q1)."""
        C = numpy.zeros((self.m, self.n + 1))
        C[:, 0] = self.tau
        C[:, 1] = self.iter
        return C
