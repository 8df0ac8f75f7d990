"""
Host-side feeders of the search: bounds heuristics, ratio sort / un-sort, sample check.
Mirrors the parts of the reference's python/DataTools.py that sit on the RunTHetA path
(DataTools.py:42-199); clustering helpers (:201-337) are out of scope.
"""
import numpy as np

_totals = {"r": None, "rN": None}


def set_total_read_counts(r, rN):
    """DataTools.py:42-45 (the reference keeps the totals in module globals)."""
    _totals["r"] = float(r)
    _totals["rN"] = float(rN)


def _ratios(r, rN):
    sr, sn = _totals["r"], _totals["rN"]
    return [(float(t) / sr) / (float(n_) / sn) for t, n_ in zip(r, rN)]


def _round_half_away(x):
    """Python 2's round(): the reference's bounds use it (DataTools.py:64,87)."""
    return float(np.floor(abs(x) + 0.5) * (1 if x >= 0 else -1))


def calculate_bounds_heuristic(x, r, rN, m, tau, k):
    """DataTools.py:47-67.  Returns (upper_bounds, lower_bounds)."""
    print("Calculating bounds using bound heuristic...")
    ratios = _ratios(r, rN)
    mean = (1.0 / m) * sum(ratios)
    std_dev = ((1.0 / (m - 1)) * sum([(mean - q) ** 2 for q in ratios])) ** 0.5
    cut = mean + x * std_dev
    lower, upper = [0] * m, [tau] * m
    for i, q in enumerate(ratios):
        if q > cut:
            y = _round_half_away(tau * q)
            lower[i] = max(tau, y - 1)
            upper[i] = max(k, y + 1)
    return upper, lower


def calculate_bounds_normal_heuristic(normal_bound_heuristic, heuristic_lb, heuristic_ub, r, rN, m, k):
    """DataTools.py:69-93."""
    print("Calculating bounds using normal bound heuristic...")
    ratios = _ratios(r, rN)
    nb = normal_bound_heuristic
    upper, lower = [nb] * m, [nb] * m
    for j, q in enumerate(ratios):
        if q < heuristic_lb:
            lower[j], upper[j] = 0, nb
        elif q > heuristic_ub:
            if q > 2:
                y = _round_half_away(nb * q)
                lower[j], upper[j] = y - 1, max(k, y + 1)
            else:
                lower[j], upper[j] = nb, k
    return upper, lower


def sort_r(rN, r):
    """DataTools.py:95-118: stable ascending sort by the normalised tumour/normal ratio."""
    sr, sn = _totals["r"], _totals["rN"]
    ratio = [(t * 1.0 / n_) * (sn / sr) for n_, t in zip(rN, r)]
    order = [i for _, i in sorted(((ratio[i], i) for i in range(len(ratio))), key=lambda kv: kv[0])]
    return [r[i] for i in order], [rN[i] for i in order], order


def sort_by_sorted_index(vec, sorted_index):
    """DataTools.py:120-130."""
    return [vec[i] for i in sorted_index]


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


def determine_frac_copy_num(rN, r, lengths, dev):
    """DataTools.py:162-199: fraction of the genome whose ratio deviates from 1 by more than dev."""
    sr, sn = sum(r), sum(rN)
    lo, hi = 1.0 - dev, 1.0 + dev
    hit = 0
    for t, n_, ln in zip(r, rN, lengths):
        if n_ == 0:
            continue
        q = (t * 1.0 / n_) * (1.0 * sn / sr)
        if q > hi or q < lo:
            hit += ln
    return float(hit) / float(sum(lengths))
