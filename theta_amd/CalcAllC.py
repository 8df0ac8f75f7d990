"""
`L2` / `L3`: drop-ins for the fork's vectorised scorers (python/CalcAllC.py:44-76), computed by the
`theta_score_batch` kernel.  Same signatures, same ValueError on shape mismatch, same 'X' markers,
and -- like the reference -- `L2` rescales its argument C in place (CalcAllC.py:54-55).
`L2_many` / `L3_many` score a batch of matrices in one launch.
"""
import math

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
    nll, vals, valid = (ctx or _lib.default_context()).score_batch(2, Cs[:, :, :2], mu2, r)   # r: (m,) or (B, m)
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


# --------------------------------------------------------------------------------------------------
# extension of the searched C to all intervals (python/CalcAllC.py:78-328), batched on the GPU scorer
# --------------------------------------------------------------------------------------------------
def weighted_C(C, rN):
    """Optimizer.py:176-182."""
    return np.asarray(C, dtype=np.float64) * np.asarray(rN, dtype=np.float64)[:, None]


def calculateX(tumorI, normalI, sumR, sumAll, mu, n, row, h):
    """CalcAllC.py:78-89: real-valued optimum of entry h of one extra row."""
    row = [x * normalI for x in row]
    nR = float(tumorI) / (sumR + tumorI)
    sumRow = sum([row[i] * mu[i] for i in range(n) if i != h])
    return float(nR * (sumAll + sumRow) - sumRow) / ((1 - nR) * mu[h])


def _common(c, mu, r, rN, all_tumor, intervals_used):
    m, n = c.shape
    c_new = np.zeros((m + 1, n))
    c_new[:m, :] = c
    c_new = weighted_C(c_new, list(rN) + [0])
    c_all = np.zeros((len(all_tumor), n))
    for i, val in enumerate(intervals_used):
        c_all[val] = c[i]
    sum_all = sum([c_new[j][k] * mu[k] for j in range(m) for k in range(n)])
    return m, n, c_new, c_all, sum_all, sum(r)


def calc_all_c_2(best, r, rN, all_tumor, all_normal, intervals_used, compat=True):
    """
    CalcAllC.py:92-143.  compat=True reproduces the fork's behaviour exactly: CalcAllC.L2 rescales its
    argument in place (CalcAllC.py:54-55), so the shared (m+1)-row matrix is multiplied by (mu, 1-mu)
    again on EVERY call.  That mutation does not depend on any likelihood, so all the
    literal matrices are built on the host and scored in one kernel launch.  compat=False scores the
    intended, unscaled matrices.
    """
    out = []
    used = set(intervals_used)
    for c, mu, likelihood, vals in best:
        m, n, c_new, c_all, sum_all, sum_r = _common(c, mu, r, rN, all_tumor, intervals_used)
        mats, rs, slots = [], [], []
        for i in range(len(all_tumor)):
            if i in used:
                continue
            if all_normal[i] == 0:
                c_all[i][0] = 2
                c_all[i][1] = -1
                continue
            c_all[i][0] = 2
            x = calculateX(all_tumor[i], all_normal[i], sum_r, sum_all, mu, n, [2, 0], 1) / all_normal[i]
            if x < 0:
                c_all[i][1] = 0
                continue
            bot, top = math.floor(x), math.ceil(x)
            rr = list(r) + [all_tumor[i]]
            c_new[m][0] = 2 * all_normal[i]
            for v in (bot, top):
                c_new[m][1] = v * all_normal[i]
                mats.append(c_new.copy())
                rs.append(rr)
                if compat:                       # what the reference's L2 does to its argument
                    c_new[:, 0] *= mu[0]
                    c_new[:, 1] *= (1 - mu[0])
            slots.append((i, int(bot), int(top)))
        if mats:
            # ONE launch for every unused interval: each matrix carries its own r (theta_score_batch_rows)
            res = L2_many([mu[0]] * len(mats), np.array(mats), m + 1, np.asarray(rs, dtype=np.float64))
            nlls = [x[0] for x in res]
            for k, (i, bot, top) in enumerate(slots):
                c_all[i][1] = bot if nlls[2 * k] < nlls[2 * k + 1] else top
        c_all_w = weighted_C(c_all, all_normal)
        like, v = L2(mu[0], c_all_w, len(all_tumor), np.asarray(all_tumor, dtype=np.float64))
        out.append([(c_all, mu, like, v)])
    return out


def _score_rows_n3(c_new, m, mu, n, jobs):
    """
    L3 of c_new with its last row replaced, for a list of jobs (r_ext, normal, rows): every (a, b) of every job is one matrix
    of ONE launch (theta_score_batch_rows: one r per matrix).  Returns the NLL lists, job by job.
    """
    total = sum(len(rows) for _r, _n, rows in jobs)
    if total == 0:
        return [[] for _ in jobs]
    mats = np.repeat(c_new[None, :, :], total, axis=0)
    rr = np.zeros((total, m + 1))
    k = 0
    for r_ext, normal, rows in jobs:
        for (a, b) in rows:
            mats[k, m, 0] = 2 * normal
            mats[k, m, 1] = a * normal
            mats[k, m, 2] = b * normal
            rr[k] = r_ext
            k += 1
    res = L3_many(np.repeat(np.asarray(mu, dtype=np.float64)[None, :], total, axis=0), mats, m + 1, rr, n)
    out, k = [], 0
    for _r, _n, rows in jobs:
        out.append([x[0] for x in res[k:k + len(rows)]])
        k += len(rows)
    return out


def calc_all_c_3(best, r, rN, all_tumor, all_normal, intervals_used):
    """CalcAllC.py:145-243 (the --NO_MULTI_EVENT variant): floor/ceil with one column at 2, plus the diagonal scan.
    All unused intervals advance together: one launch for their four floor/ceil rows and the first eight diagonal rows, then one
    launch per further block of eight for the intervals whose NLL has not risen yet (CalcAllC.py:223-232)."""
    out = []
    used = set(intervals_used)
    for c, mu, likelihood, vals in best:
        m, n, c_new, c_all, sum_all, sum_r = _common(c, mu, r, rN, all_tumor, intervals_used)
        state = {}
        for i in range(len(all_tumor)):
            if i in used:
                continue
            c_all[i][0] = 2
            if all_normal[i] == 0:
                c_all[i][1] = -1
                c_all[i][2] = -1
                continue
            nrm = all_normal[i]
            x = calculateX(all_tumor[i], nrm, sum_r, sum_all, mu, n, [2, 0, 2], 1) / nrm
            xt, xb = int(max(0, math.ceil(x))), int(max(0, math.floor(x)))
            y = calculateX(all_tumor[i], nrm, sum_r, sum_all, mu, n, [2, 2, 0], 2) / nrm
            yt, yb = int(max(0, math.ceil(y))), int(max(0, math.floor(y)))
            state[i] = {"rows": [(xb, 2), (xt, 2), (2, yb), (2, yt)], "labels": [[xb, 2], [xt, 2], [2, yb], [2, yt]],
                        "r_ext": list(r) + [all_tumor[i]], "nrm": nrm, "cand": [], "prev": float("inf"), "j": 0, "first": True}
        live = list(state.keys())
        while live:
            jobs = []
            for i in live:
                st = state[i]
                blk = [(jj, jj) for jj in range(st["j"], st["j"] + 8)]
                jobs.append((st["r_ext"], st["nrm"], (st["rows"] if st["first"] else []) + blk))
            res = _score_rows_n3(c_new, m, mu, n, jobs)
            nxt = []
            for i, nl in zip(live, res):
                st = state[i]
                if st["first"]:
                    st["cand"] += [(nl[k], st["labels"][k]) for k in range(4)]
                    nl = nl[4:]
                    st["first"] = False
                done = False
                for jj, l in zip(range(st["j"], st["j"] + 8), nl):
                    st["cand"].append((l, [jj, jj]))
                    if l > st["prev"] or l != l:
                        done = True
                        break
                    st["prev"] = l
                st["j"] += 8
                if not done:
                    nxt.append(i)
            live = nxt
        for i, st in state.items():
            st["cand"].sort()
            c_all[i][1], c_all[i][2] = st["cand"][0][1]
        c_all_w = weighted_C(c_all, all_normal)
        like, v = L3(mu, c_all_w, len(all_tumor), np.asarray(all_tumor, dtype=np.float64), n)
        out.append([(c_all, mu, like, v)])
    return out


def calc_all_c_3_multi_event(best, r, rN, all_tumor, all_normal, intervals_used):
    """CalcAllC.py:245-328 (default for n=3): for every x in 0..ceil(x*) the best of floor/ceil(y*(x)).
    The candidate rows of ALL unused intervals are scored in one launch."""
    out = []
    used = set(intervals_used)
    for c, mu, likelihood, vals in best:
        m, n, c_new, c_all, sum_all, sum_r = _common(c, mu, r, rN, all_tumor, intervals_used)
        jobs, owners = [], []
        for i in range(len(all_tumor)):
            if i in used:
                continue
            c_all[i][0] = 2
            if all_normal[i] == 0:
                c_all[i][1] = -1
                c_all[i][2] = -1
                continue
            nrm = all_normal[i]
            maxX = math.ceil(calculateX(all_tumor[i], nrm, sum_r, sum_all, mu, n, [2, 0, 0], 1) / nrm)
            if maxX < 0:
                maxX = 0
            rows = []
            for x in range(int(maxX) + 1):
                y = calculateX(all_tumor[i], nrm, sum_r, sum_all, mu, n, [2, x, 0], 2) / nrm
                bot, top = int(max(0, math.floor(y))), int(max(0, math.ceil(y)))
                if x < 2:
                    bot, top = min(bot, 2), min(top, 2)
                elif x > 2:
                    bot, top = max(2, bot), max(2, top)
                rows += [(x, bot), (x, top)]
            jobs.append((list(r) + [all_tumor[i]], nrm, rows))
            owners.append(i)
        for i, (_r, _n, rows), nl in zip(owners, jobs, _score_rows_n3(c_new, m, mu, n, jobs)):
            lmin, row_min = float("inf"), None
            for (x, yv), l in zip(rows, nl):       # strict '<' in evaluation order, like the reference
                if l < lmin:
                    lmin, row_min = l, (x, yv)
            c_all[i][1], c_all[i][2] = row_min
        c_all_w = weighted_C(c_all, all_normal)
        like, v = L3(mu, c_all_w, len(all_tumor), np.asarray(all_tumor, dtype=np.float64), n)
        out.append([(c_all, mu, like, v)])
    return out
