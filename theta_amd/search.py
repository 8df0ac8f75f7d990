"""
The search driver: drop-ins for `do_optimization_single` / `do_optimization` / `find_mins`
(python/RunTHetA.py:107-220) on top of the fused HIP search.

What runs where
  GPU  theta_search       every candidate of the rank range: enumerate + solve + NLL + running minimum;
                          returns the few candidates within COLLECT_WINDOW of the minimum
  GPU  theta_solve_batch  those finalists again, in the reference's own summation order (mu, NLL, p*)
  host this file          the reference's *sequential* tie rule (isClose with margin 1e-3 against the
                          first running minimum, RunTHetA.py:198-206) replayed in enumeration order,
                          un-sorting of rows (DataTools.py:132-159), and -- with several GPUs -- one
                          small RCCL exchange of the per-shard finalists.

How the reference's per-candidate outcome is followed:
  * n=3 candidates are reported the way Optimizer._solve_n3plus reports them: their own optimum where the reference's
    fsolve run (MINPACK hybrj, restated in csrc/hybrj4.hpp) ends inside [0,1]^3, else the nu = (1/3,1/3,1/3) fallback
    (`fallback_records`), or None where its BFGS line search walks out of the domain;
  * rank-deficient candidates -- rows (x_i, y_i) on one line: equal tumour columns, x + y = const, an all-zero column
    (whose arithmetic is NaN in the reference from normalize_C on) -- are not solved by the search kernels at all: the
    reference's hybrj runs on an exactly singular Jacobian there and what it reports is not the candidate's optimum (an
    unconverged nu in [0,1]^3 that does not sum to one, hence a mu with a negative entry and a NaN likelihood, or a finite
    one below the true minimum).  They take part with what the reference makes of them (`degenerate_records`:
    theta_solve_batch, the reference's procedure restated).  A NaN likelihood counts as "close" to anything
    (Misc.py:44-46), so the reference appends such entries to `best` wherever they stand after the last replacement of the
    minimum; so does the replay here.
  * about one FULL-RANK candidate in a million gets a NaN likelihood from the reference as well (its hybrj stops unconverged at a
    nu in [0,1]^3 that does not sum to one; which candidates, only the iteration itself tells).  Spaces of up to NAN_SWEEP_MAX
    matrices are therefore swept: every candidate through the restated procedure on the GPU, the NaN ones -- and whatever it
    reports within the window of the search's minimum -- listed with the rank-deficient ones (option "n3_nan_sweep",
    csrc/api.hip: nan_sweep): the listed outcome is THE outcome, so a swept space's `best` is the replay over the procedure's own
    outcomes by construction.  Beyond that size `last_report.nan_sweep` is False.
"""
import os
import sys

import numpy as np

from . import _lib

TIE_MARGIN = 10e-4        # Misc.py:36
# n=3 spaces of up to this many matrices are swept for the candidates the reference reports with a NaN likelihood (csrc/api.hip:
# nan_sweep; 2e8 - 5e8 candidates/s, so 2^33 take about half a minute -- the reference itself walks 4e2 candidates/s per process)
NAN_SWEEP_MAX = int(float(os.environ.get("THETA_NAN_SWEEP_MAX", 2 ** 33)))
COLLECT_WINDOW = 0.5      # how far above the minimum the GPU reports candidates (>> TIE_MARGIN)
# n=3 spaces of at least this many matrices are searched by BRANCH AND BOUND above the prefix (theta_bnb + theta_search on the
# surviving rank ranges) instead of rank by rank: 2^40 candidates are two seconds of the linear walk, BASELINE config 3 (4e27) is 3e8 years
BNB_MIN_CANDIDATES = int(float(os.environ.get("THETA_BNB_MIN_CANDIDATES", 2 ** 40)))
# ... and first of all by branch and bound over the MIXTURE space (theta_mix_search: an octree over mu, every box bounding all matrices at
# once): a fraction of a second for BASELINE configs 3 and 4.  The row-tree walk (theta_bnb) is the fallback where that one gives up
MIX_LEAF_REL = float(os.environ.get("THETA_MIX_LEAF_REL", 0))      # 0: from the data -- a fraction of the radius of the region of mixtures within the window, sqrt(2 window / sum r)
USE_MIX = os.environ.get("THETA_USE_MIX", "1") != "0"
MIX_DIVE = os.environ.get("THETA_MIX_DIVE", "1") != "0"        # the attainable NLL the search starts from: a beam search down the tree first (round 6)
MIX_DIVE_BEAM = int(os.environ.get("THETA_MIX_DIVE_BEAM", 0))           # boxes per level of the dive (0: 512, 1024 for more than 64 intervals)
MIX_DIVE_LEAF = float(os.environ.get("THETA_MIX_DIVE_LEAF", 1e-3))   # ... down to boxes of this relative size (the best row of every interval at their centres is proposed)
MIX_FIRST_BOXES = int(float(os.environ.get("THETA_MIX_FIRST_BOXES", 6e6)))   # boxes the first thresholded walk may test before the incumbent is improved instead
MIX_FIRST_MS = float(os.environ.get("THETA_MIX_FIRST_MS", 400.0))           # ... or run for this long
MIX_MAX_MS_SMALL = float(os.environ.get("THETA_MIX_MAX_MS_SMALL", 20000.0))   # (and the clock of a walkable space's last walk)
MIX_MAX_MS_LARGE = float(os.environ.get("THETA_MIX_MAX_MS_LARGE", 0.0))     # the clock of a space NO walk finishes: 0 = none, it is searched to the end
MIX_WALKABLE = int(float(os.environ.get("THETA_MIX_WALKABLE", 2 ** 46)))     # spaces up to this size fall back to the walks when the mixture-space search meets a flat likelihood ...
MIX_MAX_BOXES_SMALL = int(float(os.environ.get("THETA_MIX_MAX_BOXES_SMALL", 6e7)))   # ... i.e. more boxes than this within the threshold
MIX_LINES = os.environ.get("THETA_MIX_LINES", "1") != "0"      # the rank-deficient matrices too: one more tree per line of the alphabet's grid (round 6)
GET_VALUES_MAX = int(float(os.environ.get("THETA_GET_VALUES_MAX", 2 ** 32)))    # candidates a --GET_VALUES dump may hold (a line each)
BNB_BEAM = int(os.environ.get("THETA_BNB_BEAM", 1024))           # nodes per level of the dive that finds the first attainable NLL
BNB_MAX_NODES = int(float(os.environ.get("THETA_BNB_MAX_NODES", 2 ** 27)))     # node budget of the row-tree walk (a minute of the GPU): beyond, the call gives up
BNB_LINE_NODES = int(float(os.environ.get("THETA_BNB_LINE_NODES", 2 ** 22)))   # node budget of the walk that also follows collinear prefixes

pre = "theta"             # prefix of the --GET_VALUES dump (the reference keeps it in a module global, RunTHetA.py:307-308)


def isClose(v1, v2, margin=TIE_MARGIN):
    """Misc.py:36-47 (NaN counts as close)."""
    for a, b in zip(v1, v2):
        if abs(a - b) > margin:
            return False
    return True


def inRange(v1, minVal=0, maxVal=1):
    """Misc.py:49-57 (NaN counts as in range)."""
    for v in v1:
        if v < minVal or v > maxVal:
            return False
    return True


def reverse_sort_C(C, sorted_index):
    """DataTools.py:132-146."""
    out = np.zeros(C.shape)
    out[np.asarray(sorted_index, dtype=np.int64), :] = C
    return out


def reverse_sort_list(vec, sorted_index):
    """DataTools.py:148-159."""
    out = [0] * len(sorted_index)
    for i, dst in enumerate(sorted_index):
        out[dst] = vec[i]
    return out


def find_mins(best):
    """RunTHetA.py:107-122: merge per-worker lists; only each list's first entry is compared."""
    lowest = float('inf')
    out = []
    for solns in best:
        if len(solns) == 0:
            continue
        nll = solns[0][2]
        if isClose([lowest], [nll]):
            out += solns
        elif nll < lowest:
            lowest = nll
            out = solns
    return out


class SearchReport(object):
    """What the last search did (counters from the kernel, flags from the host replay)."""

    def __init__(self):
        self.stats = {}
        self.candidates = 0
        self.finalists = 0
        self.tie_ambiguous = False
        self.parity_uncertain = False
        self.suspects = 0              # rejected candidates whose unconstrained optimum is within the window
        self.suspect_bound = float("inf")  # smallest NLL any of them can take on the simplex boundary
        self.certificate_complete = True   # False if the device suspect list overflowed (poor sub-range without a hint)
        self.fallback_finalists = 0        # n=3: rejected candidates that joined the finalists with the reference's nu = 1/3 value
        self.degenerate = 0                # n=3: rank-deficient candidates (reported like the reference does)
        self.dropped_not_ok = 0            # finalists of the fused kernel the reference-order re-solve returned None for
        self.suspect_reruns = 0            # pieces of the range searched again because their suspect list overflowed
        self.window = COLLECT_WINDOW       # how far above the minimum the device collected finalists (narrowed if the tie list overflowed)
        self.nan_sweep = None              # n=3: True = every candidate also went through the reference's own procedure and the ones it
                                           # reports with a NaN likelihood joined the replay; False = the space was too large for that
                                           # (NAN_SWEEP_MAX): `best` then lacks the NaN tuples the reference appends for about one full-rank
                                           # matrix in a million (the finite entries and the chosen C are not affected)
        self.nan_sweep_from = 0            # ... from this rank on (> 0: only the tail behind the last replacement of the minimum was swept -- all that
                                           # can hold a NaN tuple the reference keeps)
        self.seconds = 0.0
        self.libm_pow_matches = None       # n=3: does this host's libm square like the restatement in the kernels (csrc/refpow.hpp)?  False: the
                                           # reference RUN ON THIS HOST would report other values for rank-deficient candidates (its outcome there
                                           # hangs on the last bit of libm's pow(x, 2)); everything else is unaffected
        self.gpus = 1                      # ranks the search was sharded over (do_optimization with max_processes > 1)
        self.mix = None                    # n=3 spaces of BNB_MIN_CANDIDATES matrices or more: what the mixture-space branch and bound did (mix_records)
        self.bnb = None                    # n=3 spaces of BNB_MIN_CANDIDATES matrices or more: what the branch and bound did (bnb_plan) -- incumbent of the
                                           # dive, threshold, nodes per depth, surviving rank ranges and the matrices in them, seconds;
                                           # rank_deficient_complete False = collinear prefixes were bounded like the others (their matrices, which the
                                           # reference values off their optimum, are only in `best` where their optimum is within the window)


last_report = SearchReport()


def _full_matrix(c_u8, n, tau):
    m = c_u8.shape[0]
    C = np.zeros((m, n))
    C[:, 0] = tau
    if n == 2:
        C[:, 1] = c_u8
    else:
        C[:, 1:] = c_u8
    return C


def collect_finalists(problem, ctx, r, rN, max_normal, begin, end, window=COLLECT_WINDOW, report=None, ranges=None):
    """
    GPU part: fused search over [begin, end) -- or over the rank ranges `ranges` a branch and bound left -- then the exact-order
    re-solve of the finalists.
    Returns (records, stats); a record is dict(rank, c (uint8), mu (n floats), nll, vals (m floats)).
    """
    # A flat likelihood (a few reads per interval) can put more candidates within the window of the minimum than the device tie
    # list (or the list of rejected candidates near it) holds: the window only has to cover the reference's tie margin (1e-3, twice: its reference point is the FIRST
    # minimum of the final cluster) -- so go again with a narrower one before giving up.
    for attempt, wnd in enumerate((window, window / 10.0, window / 50.0)):
        try:
            res = problem.search(begin, end, window=wnd) if ranges is None else problem.search_ranges(ranges, window=wnd)
            break
        except _lib.ListOverflow:
            # (ties, suspects or -- a piece of 2^20 candidates that still holds more rank-deficient ones than the list, with the
            # sweep's near-minimum entries among them -- the degenerate list: all three shrink with the window)
            if attempt == 2 or wnd / 10.0 < 4 * TIE_MARGIN:
                raise
    if report is not None:
        report.window = wnd
    k = len(res["rank"])
    recs = []
    dropped = 0
    if k:
        ok, mu, nll, vals = ctx.solve_batch(problem.n, problem.tau, r, rN, res["C"], max_normal, want_vals=True)
        for i in range(k):
            if not ok[i]:
                dropped += 1   # borderline: admissible in the fused arithmetic, None in the reference-order arithmetic
                continue
            recs.append({"rank": res["rank"][i], "c": res["C"][i], "mu": mu[i].copy(), "nll": float(nll[i]),
                         "vals": vals[i].copy()})
    if report is not None:
        report.dropped_not_ok = dropped
        report.suspect_reruns = problem.suspect_reruns
    return recs, res["stats"]


def degenerate_records(problem, ctx, r, rN, max_normal, report=None, recs=None, window=COLLECT_WINDOW):
    """
    n=3 rank-deficient candidates of the searched range (rows on one line; all-zero tumour columns among them), valued the way
    the reference values them (theta_solve_batch: hybrj on the singular / NaN system, M3's hybrd call, L3's sums -- a finite
    number or NaN).  The finite ones are ordinary entries of the replay whatever their value (the reference may report one BELOW
    the candidate's true minimum), the NaN ones interact with the running minimum wherever they stand.
    Returned: every NaN one, and the finite ones within `window` of the smallest value known -- theirs and `recs`' --: what lies
    beyond cannot enter the replay (replay_records cuts at the first gap above the minimum, which the window covers), and a space of
    1e10 matrices holds millions of them (round 6: a Python record each was gigabytes of host memory).
    """
    ranks, Cs = problem.last_degenerate
    if not len(ranks):
        return []
    ok, mu, nll, _vals = ctx.solve_batch(3, problem.tau, r, rN, Cs, max_normal, want_vals=False)
    ok = np.asarray(ok) > 0
    nll = np.asarray(nll, np.float64)
    finite = ok & (nll == nll)
    known = [t["nll"] for t in (recs or []) if t["nll"] == t["nll"]]
    if finite.any():
        known.append(float(nll[finite].min()))
    low = min(known, default=float("inf"))
    keep = np.nonzero(ok & ((nll != nll) | (nll <= low + window)))[0]
    out = []
    if len(keep):
        Ck = np.ascontiguousarray(np.asarray(Cs)[keep])
        ok2, mu2, nll2, vals2 = ctx.solve_batch(3, problem.tau, r, rN, Ck, max_normal, want_vals=True)
        out = [{"rank": ranks[int(i)], "c": Ck[j], "mu": mu2[j].copy(), "nll": float(nll2[j]), "vals": vals2[j].copy(), "kind": "degenerate"}
               for j, i in enumerate(keep) if ok2[j]]
    if report is not None:
        report.degenerate = int(ok.sum())
        report.degenerate_kept = len(out)
    return out


def fallback_records(problem, ctx, r, rN, max_normal, recs, window=COLLECT_WINDOW, report=None):
    """
    n=3: what the reference reports for candidates whose optimum lies OUTSIDE the simplex.  Its fsolve root is then out of
    range, fmin_bfgs -- handed dL3_hat, which points uphill (Optimizer.py:246-265) -- returns its start, and
    nu = (1/3, 1/3, 1/3) is accepted (Optimizer.py:150-160): such a candidate IS in the reference's running minimum, with
    the NLL of that point (4 467 of the 21 050 entries of the reference's own m=6, K=3 table).  The fused kernel rejects
    these candidates but lists the ones whose lower bound comes within the window ("suspects"); theta_solve_batch -- the
    reference-order arithmetic -- evaluates the fallback point for them, and those within the window of the minimum join
    the finalists.  Returns the extra records.
    """
    ranks, lbound, Cs = problem.last_suspects
    if not len(ranks):
        return []
    ok, mu, nll, vals = ctx.solve_batch(3, problem.tau, r, rN, Cs, max_normal, want_vals=True)
    lowest = min([t["nll"] for t in recs if t["nll"] == t["nll"]] + [float(v) for v, o in zip(nll, ok) if o and v == v],
                 default=float("inf"))
    have = set(t["rank"] for t in recs)
    out = []
    for i in range(len(ranks)):
        if ok[i] and nll[i] <= lowest + window and ranks[i] not in have:
            out.append({"rank": ranks[i], "c": Cs[i], "mu": mu[i].copy(), "nll": float(nll[i]), "vals": vals[i].copy(), "kind": "fallback"})
    if report is not None:
        report.fallback_finalists = len(out)
    return out


def replay_records(recs, first_duplicate, report=None, q1_first=None):
    """
    The reference's running-minimum rule (RunTHetA.py:194-206) replayed in enumeration order over the finalists; returns the records
    of `best`, in order.  `recs` must hold every candidate within COLLECT_WINDOW of the minimum.
    first_duplicate: n=2 evaluates the rank-0 matrix twice (quirk Q1, RunTHetA.py:188,208).
    q1_first: optional record of the n=3 [tau,0,0] matrix the reference evaluates first.
    """
    if not recs and q1_first is None:
        return []
    recs = sorted(recs, key=lambda t: t["rank"])
    # cut the finalists at the first gap wider than the tie margin: nothing above it can interact
    # with the running minimum ("tie replay").  Records with a NaN likelihood always stay: isClose(NaN)
    # is True (Misc.py:44-46), the reference appends them to whatever list it holds at that moment.
    finite = [t["nll"] for t in recs if t["nll"] == t["nll"]]
    if finite:
        vals_sorted = sorted(finite)
        cut = vals_sorted[-1]
        found_gap = False
        for a, b in zip(vals_sorted, vals_sorted[1:]):
            if b - a > TIE_MARGIN:
                cut = a
                found_gap = True
                break
        if not found_gap and vals_sorted[-1] - vals_sorted[0] > getattr(report, "window", COLLECT_WINDOW) - 2 * TIE_MARGIN and report is not None:
            report.tie_ambiguous = True
        recs = [t for t in recs if not (t["nll"] > cut)]
    seq = []
    if q1_first is not None:
        seq.append(q1_first)
    for t in recs:
        if first_duplicate and t["rank"] == 0:
            seq.append(t)
        seq.append(t)
    best = []
    lowest = float("inf")
    for t in seq:
        L = t["nll"]
        if isClose([L], [lowest]):
            best.append(t)
        elif L < lowest:
            best = [t]
            lowest = L
    return best


def replay_ties(recs, n, tau, sorted_index, first_duplicate, report=None, q1_first=None):
    """replay_records, with every entry in the reference's output form: (C in the ORIGINAL interval order, mu, NLL, vals)."""
    out = []
    for t in replay_records(recs, first_duplicate, report, q1_first):
        C = reverse_sort_C(_full_matrix(t["c"], n, tau), sorted_index)
        vals = reverse_sort_list([float(v) for v in t["vals"]], sorted_index)
        mu = (float(t["mu"][0]), float(t["mu"][1])) if n == 2 else np.array(t["mu"], dtype=np.float64)
        out.append((C, mu, float(t["nll"]), vals))
    return out


def _q1_record(ctx, n, m, tau, r, rN, max_normal=1.0):
    """
    Quirk Q1 for n=3: the reference first evaluates [tau,0,0]*m whatever the bounds (Enumerator.py:154-160,
    RunTHetA.py:188).  Both tumour columns are zero, its solver runs on NaNs, and M3 / L3 end at the
    uniform-by-normal-count model (p_i = rN_i / sum rN) with a mu that is rounding residue -- theta_solve_batch
    reproduces both.  None if the reference's solve returns None for it.
    """
    c0 = np.zeros((1, m, 2), np.uint8)
    ok, mu, nll, vals = ctx.solve_batch(3, tau, [int(x) for x in r], [int(x) for x in rN], c0, max_normal, want_vals=True)
    if not ok[0]:
        return None
    return {"rank": -1, "c": c0[0], "mu": mu[0].copy(), "nll": float(nll[0]), "vals": vals[0].copy()}


def _dump_values(problem, n, m, q1=None):
    """--GET_VALUES (RunTHetA.py:210-215): '<C column 1 as digits> TAB <mu0> TAB <NLL>' per accepted candidate, in the order
    the reference evaluates them -- quirk Q1 included: its first matrix (RunTHetA.py:188) is the first candidate once more
    for n=2 (that line appears twice) and the [tau,0,0] matrix for n=3 (`q1`: a line of m zeros with M3's residue as mu0)."""
    if problem.count > GET_VALUES_MAX:
        # one line per candidate: the reference would write (and evaluate) them all -- 1e27 lines for a space the branch and bound
        # above searched in a second.  Said instead of done; the search's results stand.
        print("WARNING: --GET_VALUES writes one line per candidate matrix; this search holds %.3g of them (limit %d: THETA_GET_VALUES_MAX). "
              "No .likelihoods file is written; the search itself is complete." % (float(problem.count), GET_VALUES_MAX))
        return
    with open(pre + ".likelihoods", "w") as f:
        if q1 is not None:
            f.write("0" * m + "\t" + str(float(q1["mu"][0])) + "\t" + str(float(q1["nll"])) + "\n")
        step = 1 << 16
        for b in range(0, problem.count, step):
            cnt = min(step, problem.count - b)
            C = problem.enumerate(b, cnt)
            if n == 2:
                nll, mu, _ = problem.values(b, cnt)
                rep = nll == nll
            else:
                # n=3: what the reference dumps for a matrix is the outcome of its solver calls (own optimum / nu = 1/3
                # fallback / nothing), which theta_solve_batch reproduces -- not the fused kernel's optimum
                ok, mu, nll, _v = problem.ctx.solve_batch(3, problem.tau, problem.r, problem.rN, C, problem.max_normal, want_vals=False)
                rep = ok
            col = C if n == 2 else C[:, :, 0]
            for i in range(cnt):
                if rep[i]:        # (a NaN likelihood of a reported n=3 tuple is written as 'nan', like the reference does)
                    line = "".join(str(int(v)) for v in col[i]) + "\t" + str(float(mu[i, 0])) + "\t" + str(float(nll[i])) + "\n"
                    f.write(line * 2 if (n == 2 and b + i == 0) else line)


def adjusted_bounds(lower_bounds, upper_bounds):
    """Enumerator._check_bound_order (Enumerator.py:90-113) on copies: lb made non-decreasing front to back, ub back to front."""
    lb, ub = [int(v) for v in lower_bounds], [int(v) for v in upper_bounds]
    for i in range(1, len(lb)):
        if lb[i] < lb[i - 1]:
            lb[i] = lb[i - 1]
    for i in reversed(range(len(ub) - 1)):
        if ub[i] > ub[i + 1]:
            ub[i] = ub[i + 1]
    return lb, ub


def in_space_n3(C, lb, ub, tau):
    """Is the matrix with tumour columns C (m, 2) one Enumerator._generate_next_C_3 yields (Enumerator.py:172-242)?  Rows valid
    ((tau - a)(tau - b) >= 0, :262-264) and within the (order-adjusted) bounds (:241); the symmetry rule (a <= b in the first row
    and while every earlier row had a == b, :181-183, 199-202); every edge a repeat or an increase of some component
    (:258-260); the ratio window of mu1 / mu2 never empty (:225-239, 212) -- exact: ratios compared by cross-multiplication."""
    m = len(C)
    sw = True
    lo = hi = None                                       # ratio bounds as (numerator, denominator > 0); None: unbounded
    for i in range(m):
        a, b = int(C[i][0]), int(C[i][1])
        if (tau - a) * (tau - b) < 0 or not (lb[i] <= a <= ub[i] and lb[i] <= b <= ub[i]):
            return False
        if sw and a > b:
            return False
        if i > 0:
            pa, pb = int(C[i - 1][0]), int(C[i - 1][1])
            if not ((a, b) == (pa, pb) or a > pa or b > pb):
                return False
            dx, dy = a - pa, b - pb
            if dx != 0 and dy != 0:
                ratio = (dy, -dx) if dx < 0 else (-dy, dx)             # dy / (-dx) with a positive denominator
                if dx > 0:
                    if lo is None or ratio[0] * lo[1] > lo[0] * ratio[1]:
                        lo = ratio
                else:
                    if hi is None or ratio[0] * hi[1] < hi[0] * ratio[1]:
                        hi = ratio
                if lo is not None and hi is not None and lo[0] * hi[1] > hi[0] * lo[1]:
                    return False
        sw = sw and a == b
    return True


def in_space_n3_batch(Ms, lb, ub, tau):
    """in_space_n3 for a batch of matrices (B, m, 2) at once -- the same five rules in numpy (Enumerator.py:172-264).  The ratio
    window needs only its FINAL state: the lower end only rises and the upper end only falls along the rows, so it is empty at
    some row exactly if it is empty at the last.  Ratios are quotients of integers below 16 in magnitude: correctly rounded
    division maps equal ratios to equal doubles and distinct ones at least 1/225 apart."""
    Ms = np.asarray(Ms, np.int64)
    if Ms.ndim != 3 or Ms.shape[0] == 0:
        return np.zeros(0, bool)
    lbv, ubv = np.asarray(lb, np.int64)[None, :], np.asarray(ub, np.int64)[None, :]
    a, b = Ms[:, :, 0], Ms[:, :, 1]
    ok = np.all(((tau - a) * (tau - b) >= 0) & (a >= lbv) & (a <= ubv) & (b >= lbv) & (b <= ubv), axis=1)
    # symmetry: the first row with a != b has a < b
    ne = a != b
    first = np.argmax(ne, axis=1)
    idx = np.arange(Ms.shape[0])
    ok &= ~ne.any(axis=1) | (a[idx, first] < b[idx, first])
    if Ms.shape[1] > 1:
        dx, dy = np.diff(a, axis=1), np.diff(b, axis=1)
        ok &= np.all(((dx == 0) & (dy == 0)) | (dx > 0) | (dy > 0), axis=1)
        both = (dx != 0) & (dy != 0)
        with np.errstate(divide="ignore", invalid="ignore"):
            v = -dy.astype(np.float64) / dx.astype(np.float64)
        lo = np.where(both & (dx > 0), v, -np.inf).max(axis=1)
        hi = np.where(both & (dx < 0), v, np.inf).min(axis=1)
        ok &= lo <= hi
    return ok


HEURISTIC_BUDGET_S = float(os.environ.get("THETA_HEURISTIC_BUDGET_S", 10.0))    # safety cap (seconds) of the local search on top of the grid; what bounds it is the number of rounds
HEURISTIC_ROUNDS_LARGE = int(os.environ.get("THETA_HEURISTIC_ROUNDS_LARGE", 1))  # ... of a problem with more than 4096 (interval, row) pairs (the proposal passes of mix_records carry on from there)


def canonical_columns(Ms):
    """matrices (B, m, 2) with the tumour columns in the order the reference's symmetry rule wants (Enumerator.py:181-183, 199-202):
    the first row with a != b has a < b -- the mirror image of a matrix is the same model with mu1 and mu2 exchanged"""
    Ms = np.asarray(Ms).copy()
    if Ms.ndim != 3 or len(Ms) == 0:
        return Ms
    ne = Ms[:, :, 0] != Ms[:, :, 1]
    first = np.argmax(ne, axis=1)
    idx = np.arange(len(Ms))
    swap = ne.any(axis=1) & (Ms[idx, first, 0] > Ms[idx, first, 1])
    Ms[swap] = Ms[swap][:, :, ::-1]
    return Ms


def heuristic_incumbent(ctx, m, tau, lower_bounds, upper_bounds, r, rN, max_normal, rounds=40, budget_s=None):
    """
    An NLL the reference really reports for SOME matrix of an n=3 space too large to walk -- the starting threshold of the branch
    and bound (mix_records, bnb_plan).  No reference counterpart (RunTHetA.py:173-220 visits every matrix).  For a fixed mixture mu the
    likelihood -sum r_i ln(c_i.mu) + Rtot ln(sum rN_h c_h.mu) is maximised one interval at a time in closed form; with the
    intervals sorted by their read-depth ratio (sort_r) the rows so chosen rise with c.mu, which is exactly what the
    reference's row graph and ratio window ask of consecutive rows -- so the assignment for a mixture usually IS a matrix of the space
    (checked: in_space_n3_batch).  A grid of mixtures (all of it at once, in numpy) gives a few hundred matrices; theta_solve_batch
    values them the way the reference would; the best goes through alternating (re-solve mu, re-assign rows) and a steepest-descent
    local search over single-row changes, every trial again a matrix of the space valued by theta_solve_batch -- for `rounds`
    rounds at most (m = 200: 12 800 trials per round, one round; `budget_s` is a safety cap on the clock).  Returns (nll, C (m, 2) uint8) or (inf, None).
    """
    import time
    t0 = time.time()
    budget_s = HEURISTIC_BUDGET_S if budget_s is None else budget_s
    lb, ub = adjusted_bounds(lower_bounds, upper_bounds)
    r = np.asarray(r, np.float64)
    rN = np.asarray(rN, np.float64)
    ri, rNi = [int(x) for x in r], [int(x) for x in rN]
    K = max(ub)
    rows = np.array([(a, b) for b in range(K + 1) for a in range(K + 1) if (tau - a) * (tau - b) >= 0], np.int64)
    lbv, ubv = np.asarray(lb, np.int64), np.asarray(ub, np.int64)
    # allowed[i, j]: row j within the bounds of interval i
    allowed = ((rows[None, :, 0] >= lbv[:, None]) & (rows[None, :, 0] <= ubv[:, None]) &
               (rows[None, :, 1] >= lbv[:, None]) & (rows[None, :, 1] <= ubv[:, None]))
    if not allowed.any(axis=1).all():
        return float("inf"), None
    Rtot = r.sum()

    def assign(mus, C=None, sweeps=4):
        """coordinate ascent on the rows for fixed mixtures mus (G, 3), all G at once; C (G, m): start (row indices) or None =
        the row whose c.mu is nearest the interval's ratio.  Returns (G, m) row indices."""
        f = tau * mus[:, 0:1] + rows[None, :, 0] * mus[:, 1:2] + rows[None, :, 1] * mus[:, 2:3]        # (G, Q): c.mu per alphabet row
        G = f.shape[0]
        gi = np.arange(G)
        with np.errstate(divide="ignore", invalid="ignore"):
            lf = np.where(f > 0, np.log(f), -np.inf)
        if C is None:
            ratio = (r / rN) * (rN.sum() / Rtot)
            scale = np.median(f, axis=1)
            C = np.empty((G, m), np.int64)
            for i in range(m):
                d = np.abs(f - (ratio[i] * scale)[:, None])
                C[:, i] = np.argmin(np.where(allowed[i][None, :], d, np.inf), axis=1)
        C = C.copy()
        Z = (rN[None, :] * f[gi[:, None], C]).sum(axis=1)
        for _ in range(sweeps):
            changed = False
            for i in range(m):
                Zi = Z - rN[i] * f[gi, C[:, i]]
                with np.errstate(divide="ignore", invalid="ignore"):
                    score = r[i] * lf - Rtot * np.log(Zi[:, None] + rN[i] * f)
                score = np.where(allowed[i][None, :] & (f > 0), score, -np.inf)
                j = np.argmax(score, axis=1)
                if not changed and np.any(j != C[:, i]):
                    changed = True
                C[:, i] = j
                Z = Zi + rN[i] * f[gi, j]
            if not changed:
                break
        return C

    def canon(Ms):
        """matrices (B, m, 2) with the tumour columns in the order the reference's symmetry rule wants"""
        Ms = Ms.copy()
        ne = Ms[:, :, 0] != Ms[:, :, 1]
        first = np.argmax(ne, axis=1)
        idx = np.arange(len(Ms))
        swap = ne.any(axis=1) & (Ms[idx, first, 0] > Ms[idx, first, 1])
        Ms[swap] = Ms[swap][:, :, ::-1]
        return Ms

    def value(Ms):
        """(matrices of the space among Ms, without repeats; their NLL by the reference's procedure, inf where it reports none; mu)"""
        if len(Ms) == 0:
            return np.zeros((0, m, 2), np.uint8), np.zeros(0), np.zeros((0, 3))
        Ms = Ms[in_space_n3_batch(Ms, lb, ub, tau)]
        if len(Ms) == 0:
            return np.zeros((0, m, 2), np.uint8), np.zeros(0), np.zeros((0, 3))
        arr = np.unique(np.ascontiguousarray(Ms.astype(np.uint8)).reshape(len(Ms), -1), axis=0).reshape(-1, m, 2)
        ok, mu, nll, _v = ctx.solve_batch(3, tau, ri, rNi, np.ascontiguousarray(arr), max_normal, want_vals=False)
        nll = np.where((ok > 0) & (nll == nll), nll, np.inf)
        return arr, nll, mu

    mu0, sp = np.meshgrid(np.linspace(0.05, 0.9, 18), np.linspace(0.05, 0.95, 19), indexing="ij")
    mu0, sp = mu0.ravel(), sp.ravel()
    grid = np.stack([mu0, (1 - mu0) * sp, (1 - mu0) * (1 - sp)], axis=1)
    mats, nll, mus = value(canon(rows[assign(grid)]))
    if not len(mats) or not np.isfinite(nll).any():
        return float("inf"), None
    best = int(np.argmin(nll))
    bM, bv, bmu = mats[best].astype(np.int64), float(nll[best]), mus[best]
    index = -np.ones((K + 1, K + 1), np.int64)
    index[rows[:, 0], rows[:, 1]] = np.arange(len(rows))
    ii, jj = np.nonzero(allowed)                                  # every (interval, row within its bounds)
    if len(ii) > 4096:
        # a round of single-row changes this large (m = 200, k = 7: 12 800 trials, 60 ms) buys one row at a time: the proposal
        # passes of mix_records, which start from whatever this returns, move all rows at once -- ONE round, not a dozen
        rounds = min(rounds, HEURISTIC_ROUNDS_LARGE)
    for _round in range(rounds):
        # (the number of ROUNDS bounds the local search: the same input gives the same incumbent, hence the same octree threshold,
        # whatever the host's load -- round-5 advice; the clock is a safety cap only)
        if time.time() - t0 > budget_s:
            break
        parts = []
        # alternate: rows for the mixture the reference reports for the best matrix so far
        if np.all(np.isfinite(bmu)) and bmu.min() >= 0:
            start = index[bM[:, 0], bM[:, 1]]
            parts.append(canon(rows[assign(np.stack([bmu, bmu[[0, 2, 1]]]), np.stack([start, start]))]))
        # steepest descent: every single-row change
        T = np.repeat(bM[None, :, :], len(ii), axis=0)
        T[np.arange(len(ii)), ii] = rows[jj]
        parts.append(T[np.any(rows[jj] != bM[ii], axis=1)])
        mats, nll, mus = value(np.concatenate(parts))
        if not len(mats):
            break
        k = int(np.argmin(nll))
        if not nll[k] < bv - 1e-9 * abs(bv):
            break
        bM, bv, bmu = mats[k].astype(np.int64), float(nll[k]), mus[k]
    return bv, np.asarray(bM, np.uint8)


def mix_records(problem, ctx, r, rN, max_normal, bounds, report=None, exchange=None, window=COLLECT_WINDOW):
    """
    The records of an n=3 space too large to walk, by branch and bound over the mixture space (theta_mix_search, csrc/bnb.hip):
    what RunTHetA.py:173-220 would have kept of the whole space -- every matrix the reference reports within `window` of the
    minimum, at its own optimum or at its nu = 1/3 fallback -- as replay records in enumeration order.
      1. an attainable NLL: heuristic_incumbent (matrices built from a grid of mixtures, valued by theta_solve_batch);
      2. theta_mix_search against that NLL + window: a superset of the matrices whose objective can be that low for SOME
         mixture mu >= 0 -- any matrix the reference reports within the window is one, since what it reports is the objective at
         a mixture;
      3. the reference's own rules (in_space_n3: symmetry, ratio window) and its own procedure (theta_solve_batch) on each.
    `rank` of a record is its position in the enumeration order among the listed matrices (theta_mix_search returns them in
    that order), not its rank in the space.  RANK-DEFICIENT matrices (rows on one line), which the reference may report BELOW
    their optimum -- at a mixture with negative entries --, are bounded by the trees of the lines (round 6: theta_mix_search with
    THETA_MIX_LINES, one quadtree over (alpha, beta) per line of the alphabet's grid) and those of one repeated row are valued
    outright: every FINITE outcome within the window is a record, and report.mix["rank_deficient_bound"] bounds the rest.  Not
    found: NaN outcomes (some rank-deficient matrices; one full-rank matrix in a million) unless a tree happens to list the matrix.
    Returns (records, stats-like dict); fills report.mix.  Raises ThetaError(ERR_CAPACITY) when too many boxes or matrices lie
    within the window (a flat likelihood): the caller falls back to the row-tree walk.
    """
    import time
    t0 = time.time()
    lb, ub = adjusted_bounds(bounds[0], bounds[1])
    info = {"passes": []}
    # leaves: boxes about as wide as the region of mixtures whose objective is within the window of a matrix's optimum (relative
    # radius sqrt(2 window / sum r): the tangent bound is then off by a fraction of the window, and few leaves list each matrix)
    leaf_final = MIX_LEAF_REL if MIX_LEAF_REL > 0 else min(5e-3, max(2e-5, 0.7 * float(np.sqrt(2.0 * max(window, 0.05) / max(float(np.sum(r)), 1.0)))))
    info["leaf_rel"] = leaf_final

    def value(props):
        """the smallest NLL the reference reports for a proposal that is a matrix of the space (None: for none of them)"""
        props = canonical_columns(props)
        pk = np.asarray(props)[in_space_n3_batch(props, lb, ub, problem.tau)] if len(props) else []
        if not len(pk):
            return None, 0
        okp, _mu, nllp, _v = ctx.solve_batch(3, problem.tau, r, rN, np.ascontiguousarray(np.asarray(pk, np.uint8)), max_normal, want_vals=False)
        fin = [float(v) for v, o in zip(nllp, okp) if o and v == v]
        return (min(fin) if fin else None), len(pk)

    def share(x):
        return float(exchange(x)) if exchange is not None else x

    # 1. an attainable NLL.  A DIVE first (round 6): no threshold, every level of the tree keeps its few hundred boxes of smallest bound
    # down to the leaf size -- a few thousand boxes, two host synchronisations --, and the matrices that fit the centres of its best
    # leaves are valued by the reference's procedure.  (How good that value is decides what the search COSTS, never what it finds.)
    inc = float("inf")
    if MIX_DIVE:
        td = time.time()
        # (the beam: 512 boxes per level; 1024 for more than 64 intervals -- on config 5's shape, m = 200, the narrower beam loses the
        # minimum's basin by 15-40 units for most alignments of the tree, profiles/r6/dive_probe_roots.txt)
        problem.set_option("mix_beam", MIX_DIVE_BEAM if MIX_DIVE_BEAM else (1024 if problem.m > 64 else 512))
        props, std = problem.mix_search(float("inf"), leaf_rel=max(leaf_final, MIX_DIVE_LEAF), cap=256, dive=True)
        found, n_in = value(props)
        info["dive"] = {"proposals": len(props), "in_space": n_in, "best": found, "boxes": std["boxes_tested"], "ms": std["wall_ms"],
                        "seconds": time.time() - td}
        if found is not None:
            inc = found
    inc = share(inc)
    heuristic_done = False

    def heuristic():
        th = time.time()
        hv, _hC = heuristic_incumbent(ctx, problem.m, problem.tau, bounds[0], bounds[1], r, rN, max_normal)
        info["heuristic_nll"] = hv if hv < float("inf") else None
        info["heuristic_seconds"] = time.time() - th
        return hv

    if not inc < float("inf"):
        # (the dive found no matrix of the space: the grid of mixtures of round 5)
        inc = share(min(inc, heuristic()))
        heuristic_done = True
        if not inc < float("inf"):
            raise _lib.ThetaError(_lib.ERR_CAPACITY, "mixture-space search: no attainable NLL to start from")

    def final(thr, guard, guard_ms):
        # the whole alphabet's tree AND one tree per line of the alphabet's grid (the rank-deficient matrices, which the reference may
        # report at a mixture with negative entries -- below their own minimum over mu >= 0), in one walk
        problem.set_option("mix_max_boxes", guard)
        problem.set_option("mix_max_ms", guard_ms)
        try:
            return problem.mix_search(thr, leaf_rel=leaf_final, cap=1 << 18, lines=MIX_LINES)
        finally:
            problem.set_option("mix_max_boxes", 0)
            problem.set_option("mix_max_ms", 0)

    # 2. the search against that NLL + window.  With a good value it is a fraction of a second; should the value be poor (the dive
    # went down the wrong basin) the walk is stopped after MIX_FIRST_BOXES boxes and the value improved first: coarse passes whose
    # best boxes propose matrices, each lowering the threshold of the next -- round 5's ladder.
    mats = st = None
    thr = inc + window + 4 * TIE_MARGIN
    exceeded = False
    try:
        mats, st = final(thr, MIX_FIRST_BOXES, MIX_FIRST_MS)
    except _lib.ThetaError as e:
        if e.code != _lib.ERR_CAPACITY:
            raise
        exceeded = True
        info["first_walk"] = {"gave_up": str(e)[:160]}
    if share(-1.0 if exceeded else 0.0) < 0.0:
        mats = st = None
        # (the dive's value is attainable: the ladder below starts from it; the host heuristic -- 20 ms at m = 50, 110 at m = 200 -- only
        # where the dive found nothing, above)
        inc = share(inc)
        walkable = problem.count <= MIX_WALKABLE
        # (a space the linear walk can finish is not worth more of the clock than the walk itself would take: 2e10 matrices a second;
        # a likelihood that flat -- a few reads per interval -- is the walk's.  A space no walk finishes is searched to the end.)
        budget_ms = min(MIX_MAX_MS_SMALL, max(300.0, 1e3 * problem.count / 2e10)) if walkable else MIX_MAX_MS_LARGE
        t_ladder = time.time()

        def left_ms():
            """what is left of the budget for ALL the passes and the last walk together (0: no budget)"""
            if not budget_ms:
                return 0
            left = budget_ms - 1e3 * (time.time() - t_ladder)
            if left <= 1.0:
                raise _lib.ThetaError(_lib.ERR_CAPACITY, "mixture-space search: %.0f ms spent on a likelihood too flat to prune by" % budget_ms)
            return left
        for leaf in (3e-2, 1e-2, 3e-3, 1e-3, 5e-4):
            if leaf <= 2.0 * leaf_final:
                break
            found = None
            try:
                problem.set_option("mix_max_ms", left_ms())
                try:
                    props, stp = problem.mix_search(inc + window + 4 * TIE_MARGIN, leaf_rel=leaf, cap=256, propose=True)
                finally:
                    problem.set_option("mix_max_ms", 0)
                found, n_in = value(props)
                if found is not None:
                    inc = min(inc, found)
                info["passes"].append({"leaf": leaf, "leaves": stp["leaves"], "proposals": len(props), "in_space": n_in, "best": found,
                                       "min_bound": stp["min_bound"], "ms": stp["wall_ms"]})
            except _lib.ThetaError as e:
                if e.code != _lib.ERR_CAPACITY:
                    raise
                info["passes"].append({"leaf": leaf, "gave_up": True})
            inc = share(inc)
        thr = inc + window + 4 * TIE_MARGIN
        # (several ranks: whether this walk gives up -- a clock, a capacity -- depends on the rank's own share of the boxes; the ranks
        # leave TOGETHER, or the one that falls back to the walks would pair its collectives with the wrong ones of the others)
        err = None
        try:
            mats, st = final(thr, MIX_MAX_BOXES_SMALL if walkable else 0, left_ms())
        except _lib.ThetaError as e:
            if e.code != _lib.ERR_CAPACITY:
                raise
            err = e
        if share(-1.0 if err is not None else 0.0) < 0.0:
            raise err if err is not None else _lib.ThetaError(_lib.ERR_CAPACITY, "mixture-space search: another rank's share of the boxes was too much for it")
    if MIX_LINES:
        # ... and the matrices of one repeated row (rank 1: the same value at every mixture), which no tree bounds
        const = problem.constant_matrices()
        if len(const):
            mats = _in_enumeration_order(np.concatenate([np.asarray(mats, np.uint8).reshape(-1, problem.m, 2), const]))
    keep = [int(i) for i in np.nonzero(in_space_n3_batch(mats, lb, ub, problem.tau))[0]] if len(mats) else []
    recs = []
    low = float("inf")
    n_def = 0
    if keep:
        arr = np.ascontiguousarray(mats[keep])
        ok, mu, nll, vals = ctx.solve_batch(3, problem.tau, r, rN, arr, max_normal, want_vals=True)
        fin = [float(v) for v, o in zip(nll, ok) if o and v == v]
        low = min(fin) if fin else float("inf")
        deficient = _rank_deficient(arr)
        for j, i in enumerate(keep):
            if ok[j] and (nll[j] != nll[j] or nll[j] <= low + window):
                kind = "degenerate" if deficient[j] else ("fallback" if ok[j] == 2 else "own")
                n_def += 1 if deficient[j] else 0
                recs.append({"rank": i, "c": arr[j], "mu": mu[j].copy(), "nll": float(nll[j]), "vals": vals[j].copy(), "kind": kind})
    mbl = st.get("min_bound_lines", float("inf"))
    info.update(incumbent=inc, threshold=thr, minimum=low if low < float("inf") else None, listed=len(mats), in_space=len(keep), records=len(recs),
                boxes_tested=st["boxes_tested"], levels=st["levels"], max_boxes=st["max_boxes"], leaves=st["leaves"], kernel_ms=st["kernel_ms"],
                min_bound=st["min_bound"], syncs=st.get("syncs"),
                # rank-deficient matrices (rows on one line): every one the reference reports at a FINITE value <= threshold is among the
                # records (valued by its own procedure); nothing finite can be reported for any other below `rank_deficient_bound`.
                # Their NaN outcomes (and the NaN of one full-rank matrix in a million) are listed only where they happen to be bounded.
                lines=st.get("lines", 0), line_leaves=st.get("line_leaves", 0), rank_deficient_records=n_def,
                rank_deficient_complete=bool(MIX_LINES), nan_complete=False,
                rank_deficient_bound=(min(mbl, thr) if MIX_LINES else None),
                search_ms=st["wall_ms"], seconds=time.time() - t0)
    if report is not None:
        report.mix = info
        report.fallback_finalists = sum(1 for t in recs if t["kind"] == "fallback")
        report.degenerate = n_def
    stats = {"evaluated": 0, "kernel_ms": st["kernel_ms"], "boxes_tested": st["boxes_tested"]}
    return recs, stats


def merge_mix_records(recs):
    """The records of a SHARDED mixture-space search after the exchange: a matrix within reach of boxes of two ranks was listed
    (and valued alike) by both -- once each; and the reference's enumeration order (RunTHetA.py:191-208 walks it) is the
    lexicographic order of the rows, a row by (b, a), which the replay needs as `rank`."""
    if not recs:
        return recs
    keys = [tuple((int(x[1]) << 4) | int(x[0]) for x in np.asarray(t["c"]).reshape(-1, 2)) for t in recs]
    order = sorted(range(len(recs)), key=lambda i: keys[i])
    out = []
    for i in order:
        if out and keys[i] == out[-1][0]:
            continue
        out.append((keys[i], recs[i]))
    merged = []
    for pos, (_k, t) in enumerate(out):
        t = dict(t)
        t["rank"] = pos
        merged.append(t)
    return merged


def _in_enumeration_order(mats):
    """matrices (B, m, 2) without repeats, in the reference's enumeration order: lexicographic in the rows, a row by (b, a)
    (Enumerator.py:248-256 builds the alphabet b-major)."""
    mats = np.asarray(mats, np.uint8)
    if len(mats) == 0:
        return mats
    keys = (mats[:, :, 1].astype(np.int64) << 4) | mats[:, :, 0].astype(np.int64)
    order = np.lexsort(keys.T[::-1])
    mats, keys = mats[order], keys[order]
    first = np.ones(len(mats), bool)
    first[1:] = np.any(keys[1:] != keys[:-1], axis=1)
    return mats[first]


def _rank_deficient(C):
    """(B, m, 2) tumour columns -> True where the points (x_i, y_i) lie on one line (rank([tau, x, y]) < 3; exact, integers)."""
    C = np.asarray(C, np.int64)
    A = np.concatenate([np.ones(C.shape[:2] + (1,), np.int64), C], axis=2)
    G = np.einsum("bij,bik->bjk", A, A)
    det = (G[:, 0, 0] * (G[:, 1, 1] * G[:, 2, 2] - G[:, 1, 2] * G[:, 2, 1])
           - G[:, 0, 1] * (G[:, 1, 0] * G[:, 2, 2] - G[:, 1, 2] * G[:, 2, 0])
           + G[:, 0, 2] * (G[:, 1, 0] * G[:, 2, 1] - G[:, 1, 1] * G[:, 2, 0]))
    return det == 0


def bnb_plan(problem, ctx, r, rN, max_normal, report=None, exchange=None, window=COLLECT_WINDOW, bounds=None):
    """
    Which rank ranges of an n=3 space can hold an entry of `best`?  (RunTHetA.py:173-220 visits every rank; BASELINE configs 3
    and 4 hold 4e27 / 2.6e38.)  Two walks of the row tree on the GPU (theta_bnb, csrc/bnb.hip):
      1. a DIVE -- no pruning, every level keeps its BNB_BEAM smallest relaxed bounds -- whose few ranges are searched for the
         first NLL the reference really reports for some matrix (finalists in reference arithmetic; nu = 1/3 fallbacks count);
      2. the EXACT walk against that NLL + `window`: every node whose relaxed lower bound lies beyond is dropped with its
         subtree; what survives comes back as rank ranges.  Every matrix whose optimum is within `window` of the minimum of
         the space lies in one of them -- finalists and suspects of an exhaustive search alike.
    Rank-deficient matrices (rows on one line) are valued by the reference OFF their optimum; the walk first
    tries to keep every collinear prefix (BNB_LINE_NODES nodes at most: small spaces), so that they all lie in the ranges too;
    where that is infeasible -- 1e21 such prefixes at m = 50 -- they are bounded like anything else and the report says so.
    Returns (ranges, incumbent); fills report.bnb.
    """
    import time
    t0 = time.time()
    info = {"dive": [], "incumbent": None, "threshold": None}
    inc = float("inf")
    if bounds is not None:
        # (a) the assignment heuristic: matrices of the space built from a grid of mixtures, valued by the reference's procedure
        th = time.time()
        hv, hC = heuristic_incumbent(ctx, problem.m, problem.tau, bounds[0], bounds[1], r, rN, max_normal)
        info["heuristic"] = {"nll": hv if hv < float("inf") else None, "seconds": time.time() - th}
        inc = hv
    beam = BNB_BEAM
    for _attempt in range(3):
        rg, st = problem.bnb(float("inf"), beam=beam)
        problem.set_option("n3_nan_sweep", 0)
        res = problem.search_ranges(rg, window=0.0)
        vals = []
        if len(res["rank"]):
            ok, _mu, nll, _v = ctx.solve_batch(3, problem.tau, r, rN, res["C"], max_normal, want_vals=False)
            vals += [float(v) for v, o in zip(nll, ok) if o and v == v]
        sus = problem.last_suspects
        if len(sus[0]):
            ok, _mu, nll, _v = ctx.solve_batch(3, problem.tau, r, rN, sus[2], max_normal, want_vals=False)
            vals += [float(v) for v, o in zip(nll, ok) if o and v == v]
        info["dive"].append({"beam": beam, "ranges": len(rg), "leaves": st["leaves"], "nodes": st["nodes_expanded"], "bnb_ms": st["wall_ms"],
                             "found": min(vals) if vals else None})
        if vals:
            inc = min(inc, min(vals))
        if inc < float("inf"):
            break
        beam *= 8
    if not inc < float("inf"):
        raise _lib.ThetaError(_lib.ERR_NO_CANDIDATES, "branch and bound: the dive found no matrix the reference accepts")
    if exchange is not None:
        inc = float(exchange(inc))
    thr = inc + window + 4 * TIE_MARGIN
    ranges = st = None
    complete_lines = True
    try:
        ranges, st = problem.bnb(thr, follow_collinear=True, max_nodes=BNB_LINE_NODES)
    except _lib.ThetaError as e:
        if e.code != _lib.ERR_OVERFLOW:
            raise
        complete_lines = False
        ranges, st = problem.bnb(thr, max_nodes=BNB_MAX_NODES)
    info.update(incumbent=inc, threshold=thr, ranges=len(ranges), leaves=st["leaves"], nodes=st["nodes_expanded"],
                children_bounded=st["children_bounded"], newton_iterations=st["newton_iterations"], pruned=st["children_pruned"],
                max_frontier=st["max_frontier"], frontier=st["frontier"], emit_depth=st["emit_depth"], kernel_ms=st["kernel_ms"],
                bnb_ms=st["wall_ms"], rank_deficient_complete=complete_lines, plan_seconds=time.time() - t0)
    if report is not None:
        report.bnb = info
    return ranges, inc


def _share_of_ranges(ranges, g, G):
    """Rank g's part of the ranges: equal numbers of matrices, cut at rank boundaries (the ranges are in rank order)."""
    if G <= 1:
        return list(ranges)
    total = sum(e - b for b, e in ranges)
    lo, hi = total * g // G, total * (g + 1) // G
    out, seen = [], 0
    for b, e in ranges:
        n = e - b
        a0, a1 = max(lo - seen, 0), min(hi - seen, n)
        if a1 > a0:
            out.append((b + a0, b + a1))
        seen += n
    return out


def _make_problem(ctx, n, m, tau, lower_bounds, upper_bounds, r, rN, max_normal):
    problem = _lib.Problem(ctx, n, m, tau, [int(x) for x in r], [int(x) for x in rN], [int(v) for v in lower_bounds],
                           [int(v) for v in upper_bounds], max_normal)
    if problem.count == 0:
        raise _lib.NoCandidates(_lib.ERR_NO_CANDIDATES, "no valid copy number profiles within the bounds")
    return problem


def _search_local(n, m, tau, lower_bounds, upper_bounds, r, rN, max_normal, shard=(0, 1), ctx=None, report=None, hint_exchange=None,
                  problem=None):
    """
    Everything one GPU does for its shard: the fused search, the finalists in reference arithmetic, and -- n=3 -- the
    records the reference reports away from a candidate's own optimum (nu = 1/3 fallbacks, all-zero columns).
    hint_exchange: with several shards, a function local_min -> global_min (an all-reduce) applied to the shard's PROBE
    minimum before the search, so that a shard whose own candidates are poor starts from what the best shard can reach
    instead of flooding its suspect list.
    """
    ctx = ctx or _lib.default_context()
    r = [int(x) for x in r]
    rN = [int(x) for x in rN]
    if problem is None:               # (do_optimization builds it first: the size of the space decides how many GPUs pay)
        problem = _make_problem(ctx, n, m, tau, lower_bounds, upper_bounds, r, rN, max_normal)
    g, G = shard
    begin = problem.count * g // G
    end = problem.count * (g + 1) // G
    # (what decides is the size of a rank's share: a shard of a few million ranks of a huge space is walked like any range)
    # (decided from a quantity every rank computes alike -- the shares differ by one across ranks, and a rank on the other side of the
    # line would pair its collectives with the wrong ones of its peers: round-5 advice)
    big = n == 3 and (getattr(problem, "_h", None) is not None or getattr(problem, "standin_mix", False)) and problem.count // G >= BNB_MIN_CANDIDATES
    use_bnb = big and problem.count < 2 ** 128 - 1 and m >= 8
    my_ranges = None
    if big and USE_MIX:
        # branch and bound over the mixture space: the whole space at once.  Several ranks (round 6): the boxes a few cuts below the
        # roots are dealt out by their path (options mix_shard_*: csrc/bnb.hip), every rank walks and lists its own, the attainable
        # NLL is agreed on by all-reduce after every step that can lower it, and the records meet in the exchange -- north_star's
        # sharding, G times the capacity for a flat likelihood (round 5 ran the whole search on every rank and used rank 0's records)
        try:
            if G > 1:
                problem.set_option("mix_shard_world", G)
                problem.set_option("mix_shard_rank", g)
            recs, stats = mix_records(problem, ctx, r, rN, max_normal, (lower_bounds, upper_bounds), report=report, exchange=hint_exchange)
            if report is not None:
                report.nan_sweep = False
                report.mix["shard"] = [g, G]
            if G > 1:
                # (positions in this rank's own list mean nothing to the others: unique keys for the exchange, the enumeration order
                # is restored from the matrices afterwards -- merge_mix_records)
                for t in recs:
                    t["rank"] = t["rank"] * G + g
            return problem, ctx, recs, stats
        except _lib.ThetaError as e:
            if e.code != _lib.ERR_CAPACITY:
                raise
            if report is not None:
                report.mix = {"gave_up": str(e)}
    if use_bnb:
        # every rank plans the whole space (the walk is deterministic and takes a fraction of a second) and searches its share of
        # the surviving ranges; the dive's minimum is agreed on first (hint_exchange: the all-reduce of a sharded search)
        try:
            all_ranges, inc = bnb_plan(problem, ctx, r, rN, max_normal, report=report, exchange=hint_exchange, bounds=(lower_bounds, upper_bounds))
        except _lib.ThetaError as e:
            if e.code not in (_lib.ERR_CAPACITY, _lib.ERR_OVERFLOW, _lib.ERR_ARG):
                raise
            # neither branch and bound gets through (ERR_ARG: a mix-only problem -- more than 64 rows within the bounds -- has no row-tree walk)
            # -- neither branch and bound gets through (a likelihood too flat to prune by: few reads, or many intervals of equal ratio)
            raise _lib.ThetaError(_lib.ERR_OVERFLOW, "%d candidate matrices, and the likelihood is too flat for either branch and bound to "
                                  "prune the space (%s)" % (problem.count, str(e)[:120]))
        my_ranges = _share_of_ranges(all_ranges, g, G)
        problem.hint(inc)
    elif hint_exchange is not None:
        local = problem._probe(begin, end) if end > begin else float("inf")
        shared = hint_exchange(local)
        if shared < float("inf"):
            problem.hint(shared)
    if n == 3:
        sweep = problem.count <= NAN_SWEEP_MAX and not use_bnb
        problem.set_option("n3_nan_sweep", 1 if sweep else 0)
        if report is not None:
            report.nan_sweep = sweep
    def gather(b, e):
        rc, st = collect_finalists(problem, ctx, r, rN, max_normal, b, e, report=report, ranges=my_ranges)
        if n == 3:
            rc = rc + fallback_records(problem, ctx, r, rN, max_normal, rc, report=report)
            # rank-deficient candidates: the listed outcome (the reference's own procedure) is THE outcome -- a None included --,
            # whatever the search kernels made of the same matrix as a finalist or a suspect (the sieve evaluates them like any other)
            listed = set(problem.last_degenerate[0])
            if listed:
                rc = [t for t in rc if t["rank"] not in listed]
            rc = rc + degenerate_records(problem, ctx, r, rN, max_normal, report=report, recs=rc)
        return rc, st
    recs, stats = gather(begin, end)
    if n == 3 and not sweep and G == 1 and NAN_SWEEP_MAX > 0 and not use_bnb:
        # A space too large to sweep whole.  The reference only KEEPS a NaN tuple that stands behind the last replacement of its
        # running minimum (a replacement starts a new list, RunTHetA.py:198-206): the ranks before the first entry of `best`
        # cannot contribute one.  If the tail behind that entry is short enough, it alone is swept -- searched once more with the
        # sweep on -- and `best` is complete after all.
        q1 = _q1_record(ctx, n, m, tau, r, rN, max_normal)
        first = replay_records(recs, False, None, q1)
        tail = max(begin, first[0]["rank"]) if first and first[0]["rank"] >= 0 else begin
        if first and end - tail <= NAN_SWEEP_MAX:
            # (the second search replaces the problem's side lists and the report's per-call figures with the TAIL's: what the first
            # one found before `tail` is kept and merged back below, so that the certificate -- the boundary minimum over every
            # suspect of the range -- and the report still speak of the whole range: round-4 advice)
            sus0, sus_dropped0, window0 = problem.last_suspects, problem.suspects_dropped, getattr(report, "window", COLLECT_WINDOW)
            counts0 = {k: getattr(report, k, 0) for k in ("fallback_finalists", "degenerate", "dropped_not_ok", "suspect_reruns")} if report is not None else {}
            head = [t for t in recs if t["rank"] < tail]
            problem.set_option("n3_nan_sweep", 1)
            known = [t["nll"] for t in first if t["nll"] == t["nll"]]
            if known:
                problem.hint(min(known))
            more, _st = gather(tail, end)
            problem.set_option("n3_nan_sweep", 0)
            recs = head + more
            keep = [i for i, rk in enumerate(sus0[0]) if rk < tail]
            if keep:
                rk1, lb1, C1 = problem.last_suspects
                C0 = np.asarray(sus0[2])[keep]
                problem.last_suspects = ([sus0[0][i] for i in keep] + list(rk1), np.concatenate([np.asarray(sus0[1])[keep], np.asarray(lb1, float)]),
                                         np.concatenate([C0, np.asarray(C1, np.uint8).reshape((-1,) + C0.shape[1:])]))
            problem.suspects_dropped = sus_dropped0 + problem.suspects_dropped
            if report is not None:
                report.nan_sweep = True
                report.nan_sweep_from = tail
                report.window = min(window0, report.window)
                # per-call figures: the head's share of the first search + the tail's (the first search's tail share is superseded)
                head_ranks = set(t["rank"] for t in head)
                report.fallback_finalists = sum(1 for t in head if t.get("kind") == "fallback") + report.fallback_finalists
                report.degenerate = sum(1 for t in head if t.get("kind") == "degenerate") + report.degenerate
                report.dropped_not_ok = counts0.get("dropped_not_ok", 0) + report.dropped_not_ok
                report.suspect_reruns = counts0.get("suspect_reruns", 0) + report.suspect_reruns
    return problem, ctx, recs, stats


def _friendly_exit(e):
    # a search the library cannot hold (n=3: more than 256 intervals, copy numbers above 15, or a range no search finishes -- the
    # reference would enumerate such a space for years): say so instead of a traceback
    print("ERROR: %s. Use fewer intervals (--NUM_INTERVALS) or tighter bounds. Exiting..." % e)
    sys.exit(1)


def _certificate(rep, problem, ctx, tau, r, rN, best):
    # n=3: candidates whose optimum lies outside the simplex take part with the reference's nu = 1/3 fallback value
    # (fallback_records).  Should the reference's solver leave its typical path on one of them (stall inside [0,1]^3),
    # whatever it reports is at least the candidate's minimum over the simplex boundary, computed exactly on the GPU:
    # above the winner => no such accident can change `best`.
    ranks, lbound, Cs = problem.last_suspects
    rep.suspects = len(ranks)
    rep.certificate_complete = problem.suspects_dropped == 0      # (a search that lost suspects raises; kept for readers)
    finite = [b[2] for b in best if b[2] == b[2]]
    if len(ranks) and finite:
        bmin = ctx.boundary_min(tau, [int(x) for x in r], [int(x) for x in rN], Cs)
        rep.suspect_bound = float(bmin.min())
        rep.parity_uncertain = bool(rep.suspect_bound < min(finite) + TIE_MARGIN)


def do_optimization_single(n, m, k, tau, lower_bounds, upper_bounds, r, rN, max_normal, sorted_index, multi_event=False,
                           get_values=False, _problem=None, _ctx=None):
    """
    RunTHetA.py:173-220 -- same arguments, same return value: list of
    (C in the ORIGINAL interval order as float64 (m, n) with column 0 == tau, mu, NLL, vals).
    """
    import time
    t0 = time.time()
    global last_report
    rep = SearchReport()
    try:
        problem, ctx, recs, stats = _search_local(n, m, tau, lower_bounds, upper_bounds, r, rN, max_normal, report=rep, ctx=_ctx,
                                                  problem=_problem)
    except _lib.NoCandidates:
        print("Error: No valid Copy Number Profiles exist for these intervals within the bounds specified. Exiting...")
        sys.exit(1)
    except _lib.ThetaError as e:
        if e.code in (_lib.ERR_OVERFLOW, _lib.ERR_ARG):
            _friendly_exit(e)
        raise
    q1 = _q1_record(ctx, n, m, tau, r, rN, max_normal) if n == 3 else None
    best = replay_ties(recs, n, tau, sorted_index, first_duplicate=(n == 2), report=rep, q1_first=q1)
    if get_values:
        _dump_values(problem, n, m, q1)
    rep.stats = stats
    rep.candidates = problem.count
    rep.finalists = len(recs)
    whole = rep.mix is not None and "gave_up" not in rep.mix
    if n == 3 and best and not whole:
        _certificate(rep, problem, ctx, tau, r, rN, best)
    elif whole:
        # (the mixture-space search has no suspects to certify: what it covers and what it cannot is in report.mix -- every finite outcome
        # within the window, rank-deficient ones included, `rank_deficient_bound`, `nan_complete: False`; round-5 advice)
        rep.certificate_complete = None
    if n == 3 and hasattr(ctx, "_h"):
        rep.libm_pow_matches = _lib.libm_pow_matches()
    rep.seconds = time.time() - t0
    last_report = rep
    return best


# below this many candidates per GPU a further GPU costs more (a process, a context, the counting tables) than it saves: a
# second of start-up against 3e10 - 9e10 candidates a second
MIN_CANDIDATES_PER_GPU = int(float(os.environ.get("THETA_MIN_CANDIDATES_PER_GPU", 2 ** 33)))


def gpus_for(max_processes, count=None):
    """
    How many GPUs a do_optimization(..., max_processes) call shards over: the reference's process count (--NUM_PROCESSES,
    RunTHetA.py:124-141) capped by the GPUs this process sees and by what the space is worth (MIN_CANDIDATES_PER_GPU).
    THETA_NGPU overrides everything (several ranks then share a GPU if there are fewer: tests on a one-GPU box).
    """
    env = os.environ.get("THETA_NGPU")
    if env and int(max_processes) > 1:        # (honoured only where the caller asked for a parallel run: a variable left over from a test
        return max(1, int(env))               # never shards a max_processes = 1 call -- round-4 advice)
    g = max(1, min(int(max_processes), _lib.device_count()))
    if count is not None:
        g = max(1, min(g, int(count // MIN_CANDIDATES_PER_GPU)))
    return g


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


WORKER_INIT = None        # tests only: "module:function" returning a worker's context (a stand-in device), see shard_worker.py


def _spawn_shards(world, args, transport, ndev):
    """Starts ranks 1 .. world-1 as fresh interpreters (python -m theta_amd.shard_worker), one per GPU; returns
    (port, [(Popen, status path)], temp dir)."""
    import pickle
    import subprocess
    import tempfile
    tmp = tempfile.mkdtemp(prefix="theta_shards_")
    port = _free_port()
    procs = []
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # (dmabuf IPC: what RCCL needs between processes on these hosts)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "THETA_NGPU"):
        env.pop(k, None)                                   # the workers' ranks come from the payload, not from a launcher
    for rank in range(1, world):
        job = {"rank": rank, "world": world, "port": port, "transport": transport, "device": rank % max(1, ndev), "args": args,
               "init": WORKER_INIT, "sys_path": [p for p in sys.path if p]}
        pay, st = os.path.join(tmp, "job%d.pickle" % rank), os.path.join(tmp, "status%d.pickle" % rank)
        with open(pay, "wb") as f:
            pickle.dump(job, f)
        procs.append((subprocess.Popen([sys.executable, "-m", "theta_amd.shard_worker", pay, st], env=env, cwd=tmp), st))
    return port, procs, tmp


def _join_shards(procs, tmp, timeout=600.0):
    """Waits for the workers and returns their status records (a worker that vanished without one is reported as such)."""
    import pickle
    import shutil
    import subprocess
    out = []
    for p, st in procs:
        try:
            p.wait(timeout)
        except subprocess.TimeoutExpired:
            p.kill()
        if os.path.exists(st):
            with open(st, "rb") as f:
                out.append(pickle.load(f))
        else:
            out.append({"state": "error", "code": -1, "message": "worker exited with %r and left no status" % (p.returncode,)})
    shutil.rmtree(tmp, ignore_errors=True)
    return out


def do_optimization(n, m, k, tau, lower_bounds, upper_bounds, r, rN, max_normal, sorted_index, max_processes=1,
                    multi_event=False, get_values=False):
    """
    RunTHetA.py:124-171 -- same arguments, same return value.  The reference feeds max_processes - 1 forked workers from a
    queue and merges their lists with find_mins; here `max_processes` is the number of GPUs of this node the candidate ranks
    are sharded over (capped by the GPUs present and by the size of the space, gpus_for): this process is rank 0 on GPU 0 and
    starts one worker process per further GPU (theta_amd.shard_worker), every rank searches its contiguous rank range
    (do_optimization_distributed), and the library's own collectives -- an all-reduce(min) of the probe minima before, ONE
    theta_exchange_finalists after, RCCL over xGMI -- replace the queue and find_mins.  The result is the list
    do_optimization_single returns (the reference's multi-process run lists tied solutions in worker order; the
    single-process order is the one reproduced).
    """
    import time
    t0 = time.time()
    global last_report
    ctx = _lib.default_context() if WORKER_INIT is None else None
    if WORKER_INIT is not None:
        import importlib
        mod, fn = WORKER_INIT.split(":")
        ctx = getattr(importlib.import_module(mod), fn)(0)
    try:
        problem = _make_problem(ctx, n, m, tau, lower_bounds, upper_bounds, r, rN, max_normal)
    except _lib.NoCandidates:
        print("Error: No valid Copy Number Profiles exist for these intervals within the bounds specified. Exiting...")
        sys.exit(1)
    except _lib.ThetaError as e:
        if e.code in (_lib.ERR_OVERFLOW, _lib.ERR_ARG):
            _friendly_exit(e)
        raise
    world = gpus_for(max_processes, problem.count)
    if world <= 1:
        return do_optimization_single(n, m, k, tau, lower_bounds, upper_bounds, r, rN, max_normal, sorted_index, multi_event,
                                      get_values, _problem=problem, _ctx=ctx)
    ndev = _lib.device_count() if WORKER_INIT is None else 0
    transport = os.environ.get("THETA_COMM_TRANSPORT") or ("rccl" if 0 < world <= ndev else "host")     # RCCL refuses two ranks on one device
    args = (n, m, k, tau, [int(v) for v in lower_bounds], [int(v) for v in upper_bounds], [int(x) for x in r], [int(x) for x in rN],
            max_normal, list(sorted_index))
    port, procs, tmp = _spawn_shards(world, args, transport, ndev)
    failure = None
    best = None
    try:
        comm = _lib.Comm(ctx if hasattr(ctx, "_h") else None, rank=0, world=world, addr="127.0.0.1", port=port, transport=transport)
        try:
            best = do_optimization_distributed(*args, comm=comm, ctx=ctx, problem=problem)
        finally:
            comm.close()
    except BaseException as e:          # (SystemExit included: the workers are waited for whatever happened here)
        failure = e
    statuses = _join_shards(procs, tmp, timeout=60.0 if failure is not None else 600.0)
    if failure is not None:
        raise failure
    bad = [s for s in statuses if s["state"] != "ok"]
    if bad:
        raise _lib.ThetaError(bad[0].get("code", -1) if isinstance(bad[0].get("code"), int) else -1,
                              "shard worker failed: %s" % bad[0].get("message"))
    rep = last_report
    rep.gpus = world
    rep.transport = transport
    rep.shard_kernel_ms = [float(rep.stats.get("kernel_ms", 0.0))] + [s.get("kernel_ms", 0.0) for s in statuses]
    if get_values:
        q1 = _q1_record(ctx, n, m, tau, r, rN, max_normal) if n == 3 else None
        _dump_values(problem, n, m, q1)
    rep.seconds = time.time() - t0
    return best


# --------------------------------------------------------------------------------------------------
# several GPUs: one process per GPU, candidate ranks sharded, ONE small exchange at the end
# --------------------------------------------------------------------------------------------------
def do_optimization_distributed(n, m, k, tau, lower_bounds, upper_bounds, r, rN, max_normal, sorted_index, comm, ctx=None, problem=None):
    """
    One process per GPU; `comm` is a theta_amd.Comm (the library's communicator: RCCL over xGMI, or its host transport in
    CPU-side tests).  Rank g searches the candidate ranks [N*g/G, N*(g+1)/G); the only communication is an all-reduce(min)
    of the shards' probe minima before the search (a hint, it changes no result) and theta_exchange_finalists after it
    (replaces find_mins, RunTHetA.py:107-122).  Every rank returns the same `best` as do_optimization_single would.
    """
    global last_report
    rep = SearchReport()
    g, G = comm.rank, comm.world

    shared = []

    def share(local_min):
        shared.append(True)
        return float(comm.allreduce_min([local_min])[0])

    # Every rank goes through the SAME sequence of collectives whatever happens to it: a data-dependent failure of one shard
    # (a device list that stays full, ERR_CAPACITY) must not leave the others waiting in the exchange -- RCCL has no time-out.
    # A rank that fails still takes part in the hint all-reduce (with +inf) and in an all-reduce(max) of a status flag right
    # after the search; then all ranks leave together, each with the failing rank's error code (round-2 advice).
    failure = None
    problem = recs = stats = None
    try:
        problem, ctx, recs, stats = _search_local(n, m, tau, lower_bounds, upper_bounds, r, rN, max_normal, shard=(g, G), ctx=ctx,
                                                  report=rep, hint_exchange=share if G > 1 else None, problem=problem)
    except _lib.NoCandidates:
        print("Error: No valid Copy Number Profiles exist for these intervals within the bounds specified. Exiting...")
        sys.exit(1)                                           # (a property of the problem: the same on every rank)
    except _lib.ThetaError as e:
        if e.code in (_lib.ERR_OVERFLOW, _lib.ERR_ARG):      # (the same on every rank: all of them leave here)
            _friendly_exit(e)
        failure = e
    if G > 1:
        if failure is not None and not shared:
            share(float("inf"))
        worst = int(comm.allreduce_max([float(failure.code) if failure is not None else 0.0])[0])
        if worst:
            if failure is not None:
                raise failure
            raise _lib.ThetaError(worst, "another rank's shard failed with this status; all ranks leave together")
    elif failure is not None:
        raise failure
    # ONE window for the merged list: a shard whose device lists overflowed collected within a narrower window of its minimum
    # (collect_finalists), so the merged list is only dense up to the narrowest one -- agree on it, exchange within it, and
    # let the ambiguity check of the replay see it on every rank (round-3 advice)
    if G > 1:
        rep.window = float(comm.allreduce_min([rep.window])[0])
    merged, gmin = comm.exchange_finalists(n, m, recs, rep.window)
    if rep.mix is not None and "gave_up" not in rep.mix:
        merged = merge_mix_records(merged)
    q1 = _q1_record(ctx, n, m, tau, r, rN, max_normal) if n == 3 else None
    best = replay_ties(merged, n, tau, sorted_index, first_duplicate=(n == 2), report=rep, q1_first=q1)
    rep.stats = stats
    rep.candidates = problem.count
    rep.finalists = len(merged)
    last_report = rep
    return best
