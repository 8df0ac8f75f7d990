"""
Choice between the n=2 and n=3 results (python/ModelSelection.py:43-149): BIC-like penalised NLL,
then the reference's extra criteria on the n=3 solution; the winner is copied to <prefix>.BEST.results.
Pure host logic on result files (row f4 of SURVEY.md section 8).
"""
import math
import os
import shutil

from .FileIO import read_interval_file


def _min_nll(path):
    best = float("inf")
    with open(path) as f:
        for line in f:
            if not line.startswith("#"):
                best = min(best, float(line.strip().split("\t")[0]))
    return best


def load_results(path):
    """FileIO.py:801-833: (NLL, C rows incl. the normal column as strings, mu) per solution."""
    out = []
    with open(path) as f:
        for line in f:
            if line.startswith("#"):
                continue
            nll, mu, C = line.strip().split("\t")[:3]
            rows = [["2"] + row.split(",") for row in C.split(":")]
            out.append((float(nll), rows, [float(x) for x in mu.split(",")]))
    return out


def get_frac_breakdown(C, lengths):
    """ModelSelection.py:151-186: fractions of the genome that are normal / clonal / sub-clonal."""
    tot = norm = clonal = sub = 0
    for i, row in enumerate(C):
        ln = lengths[i]
        tot += ln
        t = [str(x) for x in row[1:]]
        if all(v == "X" for v in t):
            continue
        if all(v == "2" for v in t):
            norm += ln
        elif all(v == t[0] for v in t):
            clonal += ln
        else:
            sub += ln
    return float(norm) / tot, float(clonal) / tot, float(sub) / tot


def additional_criteria(n2Result, n3Result, inputFile, min_pop=0.05, min_clonal=0.0, max_ratio=5, min_ratio=0.05):
    """ModelSelection.py:109-149."""
    lengths = read_interval_file(inputFile)[0]
    valid = False
    for nll, C, mu in load_results(n3Result):
        _, clonal, sub = get_frac_breakdown(C, lengths)
        small_enough, big_enough = False, True
        if clonal > 0:
            ratio = float(sub) / float(clonal)
            small_enough = ratio < max_ratio
            if ratio < min_ratio:
                big_enough = False
        if all(v > min_pop for v in mu[1:]) and clonal > min_clonal and small_enough and big_enough:
            valid = True
    return (3, n3Result) if valid else (2, n2Result)


def ModelSelection(inputFile, n2Result, n3Result):
    """ModelSelection.py:43-107."""
    tumor = normal = nint = 0
    with open(inputFile) as f:
        for line in f:
            if line.startswith("#"):
                continue
            t, n_ = line.strip().split("\t")[4:6]
            if int(n_) > 0:
                tumor += int(t)
                normal += int(n_)
                nint += 1
    p2 = 2 * _min_nll(n2Result) + (nint + 1) * math.log(tumor + normal)
    p3 = 2 * _min_nll(n3Result) + (nint + 1) * 2 * math.log(tumor + normal)
    num, res = 2, n2Result
    if p3 <= p2:
        num, res = additional_criteria(n2Result, n3Result, inputFile)
    target = res.replace(".n" + str(num) + ".results", ".BEST.results")
    print("Selected n=" + str(num) + " solution.  Writing to", target)
    shutil.copyfile(res, target)
    if os.path.isfile(res + ".pdf"):
        shutil.copyfile(res + ".pdf", target + ".pdf")
    return num, target
