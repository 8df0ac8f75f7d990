"""
A stand-in for the HIP library's Python surface (theta_amd._lib.Context / Problem), built on the CPU oracle, for testing the
HOST side of the search -- theta_amd/search.py, Problem.search's walk over the pieces of a range, the CLI -- without a GPU.
Test infrastructure only (it lives under tests/ and is never imported by the package): small spaces, every candidate through
the oracle's port of the reference solver.

What it models of the device contract (include/theta_hip.h):
  * theta_search: the accepted candidates of [begin, end) within `window` of min(hint, smallest NLL of the range), in rank order;
    n=3: only candidates the reference values at their OWN optimum are finalists; those it values at its nu = 1/3 fallback,
    and those it returns None for, are "suspects" (rejected, listed with a lower bound when that is within the window) --
    the host asks theta_solve_batch about them (search.fallback_records); matrices with an all-zero tumour column are
    neither: they come back through theta_search_degenerate.
  * theta_solve_batch: ok = 0 / 1 / 2 (None / own optimum / nu = 1/3 fallback), mu, NLL, vals in the reference's arithmetic.
"""
import warnings

import numpy as np

import theta_oracle as orc
from theta_amd import _lib

STATS0 = {"evaluated": 0, "accepted": 0, "degenerate": 0, "iterations": 0, "terms": 0, "list_overflow": 0, "flops": 0.0,
          "flops_f32": 0.0, "dismissed": 0, "survivors": 0, "fallback_candidates": 0, "kernel_ms": 0.0, "setup_ms": 0.0,
          "redo_flops": 0, "redo_flops_f32": 0, "redo_kernel_ms": 0.0, "kernel_launches": 0, "pruned": 0,
          "phase_cycles": [0] * 6, "best_nll": float("inf"), "rejected_bound": float("inf"), "rejected_rank": 0}


def solve_n3_classified(Cm, r, rN):
    """(outcome, soln): outcome 0 = None, 1 = fsolve's own iterate, 2 = the nu = 1/3 fallback (fmin_bfgs was consulted)."""
    called = []
    real = orc.optimize.fmin_bfgs

    def spy(*a, **k):
        called.append(1)
        return real(*a, **k)
    orc.optimize.fmin_bfgs = spy
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            s = orc.solve_n3(Cm, r, rN)
    finally:
        orc.optimize.fmin_bfgs = real
    return (0 if s is None else (2 if called else 1)), s


class StandinContext:
    def __init__(self):
        self.last_solve_fallback = None
        self.solve_calls = 0

    def solve_batch(self, n, tau, r, rN, C_u8, max_normal=1.0, want_vals=True):
        C_u8 = np.asarray(C_u8, dtype=np.uint8)
        B, m = C_u8.shape[0], C_u8.shape[1]
        r = [int(x) for x in r]
        rN = [int(x) for x in rN]
        ok = np.zeros(B, np.uint8)
        mu, nll, vals = np.zeros((B, n)), np.zeros(B), np.zeros((B, m))
        self.solve_calls += 1
        for b in range(B):
            if n == 2:
                Cm = orc.col_to_matrix_n2([int(v) for v in C_u8[b]], tau)
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    s = orc.solve_n2(Cm, r, rN, max_normal)
                out = 0 if s is None else 1
            else:
                Cm = orc.rows_to_matrix_n3([tuple(int(v) for v in row) for row in C_u8[b]], tau)
                out, s = solve_n3_classified(Cm, r, rN)
            ok[b] = out
            if s is not None:
                mu[b], nll[b], vals[b] = np.asarray(s[0], float), float(s[1]), np.asarray(s[2], float)
        self.last_solve_fallback = ok == 2
        return ok.astype(bool), mu, nll, (vals if want_vals else None)

    def boundary_min(self, tau, r, rN, C_u8):
        return np.full(len(C_u8), np.inf)                 # (the certificate is not what these tests are about)

    def score_batch(self, n, Cw, mu, r):
        Cw = np.asarray(Cw, float)
        B, m = Cw.shape[0], Cw.shape[1]
        r = np.asarray(r, float)
        nll, vals, valid = np.zeros(B), np.zeros((B, m)), np.zeros((B, m), bool)
        for b in range(B):
            rb = r[b] if r.ndim == 2 else r
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                if n == 2:
                    out = orc.calc_L2(float(mu[b][0]), Cw[b].copy(), m, rb)
                else:
                    out = orc.calc_L3(np.asarray(mu[b], float), Cw[b].copy(), m, rb, n)
            nll[b] = out[0]
            valid[b] = [not isinstance(x, str) for x in out[1]]
            vals[b] = [0.0 if isinstance(x, str) else float(x) for x in out[1]]
        return nll, vals, valid


class StandinProblem(_lib.Problem):
    """The whole space is enumerated and solved by the oracle at construction: keep it to a few thousand candidates."""
    LIMIT = 20000

    def __init__(self, ctx, n, m, tau, r, rN, lb, ub, max_normal=1.0):
        self.ctx, self.n, self.m, self.tau = ctx, int(n), int(m), int(tau)
        self.r, self.rN, self.max_normal = [int(x) for x in r], [int(x) for x in rN], float(max_normal)
        lb, ub = [int(v) for v in lb], [int(v) for v in ub]
        if n == 2:
            self.cands = [np.array(c, np.uint8) for c in orc.enumerate_n2(m, tau, lb, ub)]
        else:
            if m > 128 or max(ub) > 7:
                raise _lib.ThetaError(_lib.ERR_ARG, "stand-in: beyond the library")
            self.cands = []
            for rows in orc.enumerate_n3(m, tau, lb, ub):
                self.cands.append(np.array(rows, np.uint8))
                assert len(self.cands) <= self.LIMIT, "stand-in device: space too large"
        self.count = len(self.cands)
        self._table = {}
        self._hint = float("inf")
        self._dev_hint = float("inf")
        self.last_suspects = ([], np.zeros(0), None)
        self.last_degenerate = ([], None)
        self.suspects_dropped = 0
        self.suspect_reruns = 0
        self.search_calls = []
        self.options = {}
        self.mix_calls, self.mix_listed = [], []
        self._bounds = (lb, ub)
        self._h = None

    def close(self):
        pass

    def set_option(self, name, value):
        self.options[name] = value

    # ---- the mixture-space search (theta_mix_search), modelled on the enumerated space: what the HOST side of a sharded whole-space
    # search sees -- proposals, a superset of the matrices within a threshold dealt out over the ranks, the statistics it reports
    standin_mix = False            # (a test that wants the mixture-space path over this stand-in sets it on its instance: search._search_local asks)

    def mix_search(self, threshold, leaf_rel=2e-4, cap=1 << 16, propose=False, lines=False, lines_only=False, dive=False):
        import zlib
        g, G = int(self.options.get("mix_shard_rank", 0)), int(self.options.get("mix_shard_world", 1))
        st = {"boxes_tested": 1000, "levels": 10, "max_boxes": 10, "leaves": 5, "listed": 0, "matrices": 0, "lines": 7 if lines else 0, "line_leaves": 0,
              "syncs": 2, "kernel_ms": 0.0, "wall_ms": 0.0, "min_bound": float("inf"), "min_bound_lines": float("inf")}
        self.mix_calls.append(("dive" if dive else "propose" if propose else "list", float(threshold), (g, G)))
        if not (dive or propose) and getattr(self, "mix_fail_rank", None) == g:
            raise _lib.ThetaError(_lib.ERR_CAPACITY, "stand-in: this rank's share of the boxes is too much for it")
        if dive or propose:
            # a few matrices of the space, not the best ones: the driver must get to the minimum from a poor start as well
            out = [self.cands[k] for k in range(0, self.count, max(1, self.count // 5))][:cap]
            return np.array(out, np.uint8).reshape(len(out), self.m, 2), st
        out = []
        for k in range(self.count):
            o, _mu, nll = self._entry(k)
            if o and nll == nll and nll <= threshold and zlib.crc32(self.cands[k].tobytes()) % G == g:
                out.append(self.cands[k])
        self.mix_listed.append([c.tobytes() for c in out])
        st["listed"] = st["matrices"] = len(out)
        return np.array(out, np.uint8).reshape(len(out), self.m, 2), st

    def _entry(self, k):
        """(outcome, mu, nll) of candidate k in the reference's arithmetic."""
        if k not in self._table:
            c = self.cands[k]
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                if self.n == 2:
                    s = orc.solve_n2(orc.col_to_matrix_n2([int(v) for v in c], self.tau), self.r, self.rN, self.max_normal)
                    out = 0 if s is None else 1
                else:
                    out, s = solve_n3_classified(orc.rows_to_matrix_n3([tuple(int(v) for v in row) for row in c], self.tau), self.r, self.rN)
            self._table[k] = (out, None, float("nan")) if s is None else (out, np.asarray(s[0], float), float(s[1]))
        return self._table[k]

    def _is_degenerate(self, k):
        return self.n == 3 and bool((self.cands[k].sum(axis=0) == 0).any())

    def hint(self, nll_upper_bound):
        self._hint = min(self._hint, float(nll_upper_bound))
        self._dev_hint = self._hint

    def _piece(self, b, e, window, cap, hint):
        if hint < float("inf"):
            self._dev_hint = hint
        res = self._search_once(b, e, window, cap)
        if self.n != 3:
            return res, ([], np.zeros(0), None), 0, ([], None)
        return res, self._sus, 0, self._deg

    def _search_once(self, begin, end, window, cap):
        if self.count == 0:
            raise _lib.NoCandidates(_lib.ERR_NO_CANDIDATES, "no valid copy number profiles within the bounds")
        self.search_calls.append((begin, end, self._dev_hint))
        hint, self._dev_hint = self._dev_hint, float("inf")           # one-shot, like theta_problem_hint
        fin, rej, deg = [], [], []
        for k in range(begin, end):
            if self._is_degenerate(k):
                deg.append(k)
                continue
            out, mu, nll = self._entry(k)
            if out == 1 and nll == nll:
                fin.append((k, mu, nll))
            else:
                # rejected by the device: optimum outside the simplex (or a NaN likelihood).  Its lower bound of anything the
                # reference could report: the fallback value less a margin, or -- no value at all -- "cannot tell"
                rej.append((k, (nll - 0.25) if out == 2 and nll == nll else -np.inf))
        best = min([hint] + [t[2] for t in fin])
        fin = [t for t in fin if t[2] <= best + window]
        rej = [t for t in rej if t[1] <= best + window]
        shape = (lambda q: (q, self.m)) if self.n == 2 else (lambda q: (q, self.m, 2))
        st = dict(STATS0)
        st.update(evaluated=end - begin, accepted=len(fin), degenerate=len(deg), best_nll=best, phase_cycles=[0] * 6)
        res = {"nll": np.array([t[2] for t in fin]), "mu": np.array([t[1] for t in fin]).reshape(len(fin), self.n),
               "rank": [t[0] for t in fin], "C": np.array([self.cands[t[0]] for t in fin], np.uint8).reshape(shape(len(fin))),
               "stats": st}
        self._sus = ([t[0] for t in rej], np.array([t[1] for t in rej]),
                     np.array([self.cands[t[0]] for t in rej], np.uint8).reshape(len(rej), self.m, 2) if self.n == 3 else None)
        self._deg = (deg, np.array([self.cands[k] for k in deg], np.uint8).reshape(len(deg), self.m, 2) if self.n == 3 else None)
        return res

    def _probe(self, begin, end):
        running, self._hint = self._hint, float("inf")
        return running

    def values(self, begin, count):
        """theta_search_values: per-candidate (nll, mu) of the fused solve, NaN where the candidate is rejected."""
        nll, mu = np.full(count, np.nan), np.full((count, self.n), np.nan)
        for i in range(count):
            out, m_, v = self._entry(begin + i)
            if out == 1:
                nll[i], mu[i] = v, m_
        return nll, mu, dict(STATS0)

    def enumerate(self, begin, count):
        shape = (count, self.m) if self.n == 2 else (count, self.m, 2)
        return np.array(self.cands[begin:begin + count], np.uint8).reshape(shape)


def worker_context(rank):
    """`search.WORKER_INIT = "standin_device:worker_context"`: what a rank of do_optimization(..., max_processes) runs on in the
    CPU tests -- this process's Problem class becomes the stand-in, the context is the stand-in's."""
    warnings.simplefilter("ignore")
    _lib.Problem = StandinProblem
    return StandinContext()
