"""
Automatic interval selection for the search (the reference's default path):
python/SelectIntervals.py:50-160, 208-223.  Cluster ("meta interval") variants are out of scope.
"""
import math
import sys

MIN_LENGTH_N2 = 1000000   # SelectIntervals.py:45
MIN_LENGTH_N3 = 5000000   # SelectIntervals.py:46


def _columns(lines):
    return [[ln[j] for ln in lines] for j in range(len(lines[0]))]


def filter_intervals_n2(lengths, tumor_counts, norm_counts, m, k, lower, upper):
    """SelectIntervals.py:208-219: long enough, covered, and not amplified beyond (k+1)/2."""
    tot_t, tot_n = float(sum(tumor_counts)), float(sum(norm_counts))
    keep = [i for i in range(m) if tumor_counts[i] > 0 and norm_counts[i] > 0 and lengths[i] >= MIN_LENGTH_N2]
    return [i for i in keep if ((tumor_counts[i] / tot_t) / (norm_counts[i] / tot_n)) < float(k + 1) / 2]


def select_intervals_n2(lengths, tumor_counts, norm_counts, m, k, force, num_intervals, lower=None, upper=None):
    """SelectIntervals.py:127-160: the num_intervals longest admissible intervals, in input order."""
    idx = filter_intervals_n2(lengths, tumor_counts, norm_counts, m, k, lower, upper)
    total = float(sum(lengths))
    if lower is None or upper is None:
        lines = [[i, lengths[i], tumor_counts[i], norm_counts[i]] for i in idx]
    else:
        lines = [[i, lengths[i], tumor_counts[i], norm_counts[i], lower[i], upper[i]] for i in idx]
    lines.sort(key=lambda x: x[1])          # stable, like the reference
    lim = min(num_intervals, len(idx))
    top = lines[-lim:] if lim else []
    if sum(t[1] for t in top) < 0.1 * total:
        msg = ("WARNING: This sample isn't a good candidate for THetA analysis. The longest  %d intervals chosen for "
               "analysis represent <10%% of the combined length of all provided intervals." % lim)
        if not force:
            print(msg + " Run with --FORCE flag to ignore this warning. Exiting...")
            sys.exit(1)
        print(msg)
    top.sort(key=lambda x: x[0])
    print("\tSelected", len(top), "intervals for analysis.")
    return _columns(top)


def select_intervals_n3(lengths, tumor_counts, norm_counts, m, upper_bounds, lower_bounds, copy, tau, force,
                        num_intervals):
    """SelectIntervals.py:50-125: intervals for n=3 from the n=2 result, with tightened bounds."""
    if tau != 2:
        print("ERROR: For automatic interval selection with 3 subpopulations, the default copy number (--TAU) must be 2. "
              "To run with other values, bounds must be provided in the input file.")
        sys.exit(1)
    used = [x != "X" for x in upper_bounds]
    real = [i for i in range(m) if used[i]]
    lengths = [v for i, v in enumerate(lengths) if used[i]]
    tumor_counts = [v for i, v in enumerate(tumor_counts) if used[i]]
    norm_counts = [v for i, v in enumerate(norm_counts) if used[i]]
    ub = [int(v) for i, v in enumerate(upper_bounds) if used[i]]
    lb = [int(v) for i, v in enumerate(lower_bounds) if used[i]]
    copy = [int(v) for i, v in enumerate(copy) if used[i]]
    b = int(math.ceil(num_intervals * .75))
    c = int(num_intervals - b)
    lines = [[real[i], lengths[i], tumor_counts[i], norm_counts[i], ub[i], lb[i], copy[i]]
             for i in range(len(real)) if lengths[i] >= MIN_LENGTH_N3]
    lines.sort(key=lambda x: -x[1])
    chosen = []
    for i, ln in enumerate(lines):
        if c > 0 and ln[6] == 2 and ln[4] == 2:
            chosen.append(i)
            c -= 1
        elif b > 0 and ln[6] in [0, 1, 3]:
            chosen.append(i)
            b -= 1
    for i, ln in enumerate(lines):
        if c > 0 and ln[6] == 2 and ln[4] > 2:
            chosen.append(i)
            c -= 1
    if c > 0 or b > 0:
        msg = ("WARNING: This sample isn't a good candidate for THetA analysis with 3 subpopulations: There aren't a "
               "sufficient number of intervals that fit the criteria for interval selection.")
        if not force:
            print(msg + " Run with --FORCE flag to ignore this warning. Exiting...")
            sys.exit(1)
        print(msg)
    top = [lines[i] for i in chosen]
    for ln in top:                      # new bounds from the n=2 copy number (SelectIntervals.py:104-118)
        cn = ln[6]
        if cn == 0:
            pass
        elif cn == 1:
            ln[5] = 1
        elif cn == 2:
            ln[5] = 1
            ln[4] = min(3, ln[4])
        else:
            ln[4] = 3
    top.sort(key=lambda x: x[0])
    print("\tSelected", len(chosen), "intervals for analysis.")
    return _columns(top)
