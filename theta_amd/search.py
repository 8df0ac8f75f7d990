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


def collect_finalists(problem, ctx, r, rN, max_normal, begin, end, window=COLLECT_WINDOW, report=None):
    """
    GPU part: fused search over [begin, end) then the exact-order re-solve of the finalists.
    Returns (records, stats); a record is dict(rank, c (uint8), mu (n floats), nll, vals (m floats)).
    """
    # A flat likelihood (a few reads per interval) can put more candidates within the window of the minimum than the device tie
    # list (or the list of rejected candidates near it) holds: the window only has to cover the reference's tie margin (1e-3, twice: its reference point is the FIRST
    # minimum of the final cluster) -- so go again with a narrower one before giving up.
    for attempt, wnd in enumerate((window, window / 10.0, window / 50.0)):
        try:
            res = problem.search(begin, end, window=wnd)
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


def degenerate_records(problem, ctx, r, rN, max_normal, report=None):
    """
    n=3 rank-deficient candidates of the searched range (rows on one line; all-zero tumour columns among them), valued the way
    the reference values them (theta_solve_batch: hybrj on the singular / NaN system, M3's hybrd call, L3's sums -- a finite
    number or NaN).  All of them are returned: the finite ones are ordinary entries of the replay whatever their value (the
    reference may report one BELOW the candidate's true minimum), the NaN ones interact with the running minimum wherever
    they stand.
    """
    ranks, Cs = problem.last_degenerate
    if not len(ranks):
        return []
    ok, mu, nll, vals = ctx.solve_batch(3, problem.tau, r, rN, Cs, max_normal, want_vals=True)
    out = [{"rank": ranks[i], "c": Cs[i], "mu": mu[i].copy(), "nll": float(nll[i]), "vals": vals[i].copy(), "kind": "degenerate"}
           for i in range(len(ranks)) if ok[i]]
    if report is not None:
        report.degenerate = len(out)
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
    if hint_exchange is not None:
        local = problem._probe(begin, end) if end > begin else float("inf")
        shared = hint_exchange(local)
        if shared < float("inf"):
            problem.hint(shared)
    if n == 3:
        sweep = problem.count <= NAN_SWEEP_MAX
        problem.set_option("n3_nan_sweep", 1 if sweep else 0)
        if report is not None:
            report.nan_sweep = sweep
    def gather(b, e):
        rc, st = collect_finalists(problem, ctx, r, rN, max_normal, b, e, report=report)
        if n == 3:
            rc = rc + fallback_records(problem, ctx, r, rN, max_normal, rc, report=report)
            # rank-deficient candidates: the listed outcome (the reference's own procedure) is THE outcome -- a None included --,
            # whatever the search kernels made of the same matrix as a finalist or a suspect (the sieve evaluates them like any other)
            listed = set(problem.last_degenerate[0])
            if listed:
                rc = [t for t in rc if t["rank"] not in listed]
            rc = rc + degenerate_records(problem, ctx, r, rN, max_normal, report=report)
        return rc, st
    recs, stats = gather(begin, end)
    if n == 3 and not sweep and G == 1 and NAN_SWEEP_MAX > 0:
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
    # a search the library cannot hold (n=3: more than 256 intervals, more than 64 distinct rows (a, b) within the bounds, or a range no search finishes -- the
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
    if n == 3 and best:
        _certificate(rep, problem, ctx, tau, r, rN, best)
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
    q1 = _q1_record(ctx, n, m, tau, r, rN, max_normal) if n == 3 else None
    best = replay_ties(merged, n, tau, sorted_index, first_duplicate=(n == 2), report=rep, q1_first=q1)
    rep.stats = stats
    rep.candidates = problem.count
    rep.finalists = len(merged)
    last_report = rep
    return best
