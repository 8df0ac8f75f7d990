#!/usr/bin/env python3
"""
Golden outputs of the reference's command line over a MATRIX OF FLAGS -- data only.

Every case runs the reference (converted 2->3 outside the repo, see make_golden.py / make_golden_cli.py) on the seeded
14-interval synthetic file tests/golden/cli/syn14.intervals with one flag combination, small enough for the CPU oracle to
follow (a few hundred n=2 matrices, a few thousand n=3 matrices), and stores the text of every output file it wrote
(.withBounds, .results, .likelihoods, .BEST.results) in tests/golden/cli_matrix.json:
    {case: {"args": [...], "rc": exit code, "files": {suffix: text}}}
tests/test_host_cli_cpu.py replays the same command lines over the stand-in device and compares.
"""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden  # noqa
import make_golden_cli  # noqa

SYN = os.path.join(HERE, "cli", "syn14.intervals")
N2RES = os.path.join(HERE, "cli", "syn14s.n2.results")        # (the n=2 result the two-stage golden starts its n=3 stage from)
N2BOUNDS = os.path.join(HERE, "cli", "syn14s.n2.withBounds")

def write_odd_intervals(path):
    """20 intervals with the rows interval selection has rules for: two shorter than 1 Mb, one shorter than 5 Mb, one without
    tumour reads, one without normal reads, one amplified beyond (k+1)/2, equal lengths (the stable sort decides)."""
    import numpy as np
    rng = np.random.RandomState(2020)
    m = 20
    L = rng.randint(6_000_000, 20_000_000, m)
    L[3], L[11], L[15] = 400_000, 900_000, 3_000_000
    L[6] = L[7] = 9_000_000
    rN = rng.poisson(L * 0.01)
    c = rng.randint(0, 4, m)
    c[9] = 7
    mu = 0.4
    p = rN * (2 * mu + c * (1 - mu))
    p = p / p.sum()
    r = rng.multinomial(int(rN.sum() * 1.1), p)
    r[13] = 0
    rN[17] = 0
    with open(path, "w") as f:
        f.write("#ID\tchrm\tstart\tend\ttumorCount\tnormalCount\n")
        pos = 1
        for i in range(m):
            f.write("%d\t%d\t%d\t%d\t%d\t%d\n" % (i + 1, 1 + i // 7, pos, pos + L[i], r[i], rN[i]))
            pos += L[i] + 1


ODD = os.path.join(HERE, "cli", "odd20.intervals")

CASES = {
    "n2_select9": [SYN, "-n", "2", "-k", "3", "--NUM_INTERVALS", "9"],
    "n2_k4_maxnormal": [SYN, "-n", "2", "-k", "4", "--NUM_INTERVALS", "8", "-m", "0.6"],
    "n2_bound_heuristic": [SYN, "-n", "2", "-k", "3", "--BOUND_HEURISTIC", "0.3"],
    "n2_normal_bound_heuristic": [SYN, "-n", "2", "-k", "3", "--NORMAL_BOUND_HEURISTIC", "2", "--HEURISTIC_LB", "0.8", "--HEURISTIC_UB", "1.2"],
    "n2_no_selection": [SYN, "-n", "2", "-k", "3", "--NO_INTERVAL_SELECTION"],
    "n2_get_values": [SYN, "-n", "2", "-k", "3", "--NUM_INTERVALS", "9", "--GET_VALUES"],
    "n2_bounds_only": [SYN, "-n", "2", "-k", "3", "--BOUNDS_ONLY"],
    "n2_tau3": [SYN, "-n", "2", "-k", "4", "-t", "3", "--NUM_INTERVALS", "8"],
    "n3_from_results": [N2BOUNDS, "-n", "3", "-k", "3", "--NUM_INTERVALS", "7", "--FORCE", "--RESULTS", N2RES],
    "n3_no_multi_event": [N2BOUNDS, "-n", "3", "-k", "3", "--NUM_INTERVALS", "7", "--FORCE", "--RESULTS", N2RES, "--NO_MULTI_EVENT"],
    "n3_get_values": [N2BOUNDS, "-n", "3", "-k", "2", "--NUM_INTERVALS", "6", "--FORCE", "--RESULTS", N2RES, "--GET_VALUES"],
    "n2_min_frac_exit": [SYN, "-n", "2", "-k", "3", "--MIN_FRAC", "0.9"],
    "n2_ratio_dev": [SYN, "-n", "2", "-k", "3", "--NUM_INTERVALS", "9", "--RATIO_DEV", "0.3", "--MIN_FRAC", "0.2"],
    # (no n=2 case on a file WITH bounds columns: the reference reads them as strings and dies in Enumerator.py:136 with a
    #  TypeError -- `self.iter[i] += 1` on a str -- after writing its bounds file; theta_amd converts them and runs)
    "n3_k4_six_intervals": [N2BOUNDS, "-n", "3", "-k", "4", "--NUM_INTERVALS", "6", "--FORCE", "--RESULTS", N2RES],
    "n3_maxnormal_ignored": [N2BOUNDS, "-n", "3", "-k", "3", "--NUM_INTERVALS", "6", "--FORCE", "--RESULTS", N2RES, "-m", "0.5"],
    "odd_n2_select10": [ODD, "-n", "2", "-k", "3", "--NUM_INTERVALS", "10"],
    "odd_n2_select12_k4": [ODD, "-n", "2", "-k", "4", "--NUM_INTERVALS", "12"],
    "odd_two_stage": [ODD, "-k", "3", "--NUM_INTERVALS", "7", "--FORCE"],
    "example_n2_select8": [os.path.join(HERE, "cli", "Example.intervals"), "-n", "2", "-k", "3", "--NUM_INTERVALS", "8"],
    "example_two_stage_select7": [os.path.join(HERE, "cli", "Example.intervals"), "-k", "2", "--NUM_INTERVALS", "7", "--FORCE"],
    "n2_k6_select7": [SYN, "-n", "2", "-k", "6", "--NUM_INTERVALS", "7"],
    "n2_maxnormal_03": [SYN, "-n", "2", "-k", "3", "--NUM_INTERVALS", "9", "-m", "0.3"],
    "n2_two_processes": [SYN, "-n", "2", "-k", "4", "--NUM_INTERVALS", "8", "-m", "0.6", "--NUM_PROCESSES", "2"],
    "two_stage_tau3": [SYN, "-k", "3", "-t", "3", "--NUM_INTERVALS", "8", "--FORCE"],
    "n3_k4_five_intervals": [N2BOUNDS, "-n", "3", "-k", "4", "--NUM_INTERVALS", "5", "--FORCE", "--RESULTS", N2RES],
    "n3_without_results_file": [N2BOUNDS, "-n", "3", "-k", "3", "--NUM_INTERVALS", "7", "--FORCE"],
}


def main():
    make_golden.import_reference()
    write_odd_intervals(ODD)
    out = {}
    for name, args in CASES.items():
        tmp = tempfile.mkdtemp(prefix="theta_clim_")
        launcher = os.path.join(tmp, "_launch.py")
        with open(launcher, "w") as f:
            f.write(make_golden_cli.LAUNCH)
        p = subprocess.run([sys.executable, launcher] + args + ["-d", tmp, "-p", "c"], cwd=tmp, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True)
        files = {}
        for fn in sorted(os.listdir(tmp)):
            if fn.startswith("c.") and (fn.endswith(".results") or fn.endswith(".withBounds") or fn.endswith(".likelihoods")):
                files[fn[2:]] = open(os.path.join(tmp, fn)).read()
        rel = [a.replace(HERE + os.sep, "") for a in args]
        out[name] = {"args": rel, "rc": p.returncode, "files": files,
                     "stdout_tail": [l for l in p.stdout.splitlines() if l.startswith("ERROR") or l.startswith("WARNING")][-3:]}
        print(name, p.returncode, sorted(files), file=sys.stderr)
    with open(os.path.join(HERE, "cli_matrix.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
