"""
Command-line flags and the .intervals / .withBounds / .results file surface of RunTHetA, so that the
GPU search drops into the reference's workflow: python/FileIO.py:45-227 (flags), :386-446 (interval
reader), :484-504 (results reader), :620-664 (results writer), :733-798 (bounds + RunN3 script writers).
SNP / BAF / plotting I/O is out of scope.
"""
import argparse
import os
import sys

N_VALS = [None, 2, 3]     # FileIO.py:42
K_VALS = range(7)         # FileIO.py:43: k = 0 .. 6
MAX_K_EXTENDED = 8        # with --ALLOW_LARGE_K (not a flag of the reference): what the kernels hold (n=3: copy numbers up to 15; more than 64 distinct rows within the bounds: searched over the mixture space only)


def parse_arguments(argv=None, silent=False):
    """FileIO.py:45-227: same flags, same defaults, same 27-tuple."""
    p = argparse.ArgumentParser(prog="RunTHetA")
    p.add_argument("QUERY_FILE", help="Interval file")
    p.add_argument("--TUMOR_FILE", default=None)
    p.add_argument("--NORMAL_FILE", default=None)
    p.add_argument("-n", "--N", type=int, default=None, help="Number of subpopulations")
    p.add_argument("-k", "--MAX_K", type=int, default=3, help="The maximum value allowed for entries in C")
    p.add_argument("-t", "--TAU", type=int, default=2, help="Expected number of copies in normal genome")
    p.add_argument("-d", "--DIR", default="./", help="Directory where result file is written to")
    p.add_argument("-p", "--OUTPUT_PREFIX", default=None)
    p.add_argument("-m", "--MAX_NORMAL", type=float, default=1.0)
    p.add_argument("--NUM_PROCESSES", type=int, default=1)
    p.add_argument("--NUM_INTERVALS", type=int, default=100)
    p.add_argument("--BOUND_HEURISTIC", default=False)
    p.add_argument("--NORMAL_BOUND_HEURISTIC", type=int, default=False)
    p.add_argument("--HEURISTIC_LB", type=float, default=0.9)
    p.add_argument("--HEURISTIC_UB", type=float, default=1.1)
    p.add_argument("--BOUNDS_ONLY", action="store_true", default=False)
    p.add_argument("--NO_MULTI_EVENT", action="store_true", default=False)
    p.add_argument("--RESULTS", default=None)
    p.add_argument("--FORCE", action="store_true", default=False)
    p.add_argument("--GET_VALUES", action="store_true", default=False)
    p.add_argument("--NO_INTERVAL_SELECTION", action="store_true", default=False)
    p.add_argument("--READ_DEPTH_FILE", default=None)
    p.add_argument("--GRAPH_FORMAT", default=".pdf")
    p.add_argument("--BAF", action="store_true", default=False)
    p.add_argument("--RATIO_DEV", type=float, default=0.1)
    p.add_argument("--MIN_FRAC", type=float, default=0.05)
    p.add_argument("--NO_CLUSTERING", action="store_true", default=False)
    p.add_argument("--ALLOW_LARGE_K", action="store_true", default=False,
                   help="(theta_amd only) accept -k above the reference's limit of 6, up to %d" % MAX_K_EXTENDED)
    a = p.parse_args(argv)

    if a.N not in N_VALS:
        raise ValueError("Invalid value entered for n: " + str(a.N) + ". Currently supported values for n: " + str(N_VALS))
    k_vals = range(MAX_K_EXTENDED + 1) if a.ALLOW_LARGE_K else K_VALS          # FileIO.py:136
    if a.MAX_K not in k_vals:
        raise ValueError("Invalid value entered for k: " + str(a.MAX_K) + ". Supported values for k: 0-" + str(max(k_vals)))
    if a.TAU < 0:
        raise ValueError("Invalid value for tau: " + str(a.TAU) + ". Tau must be non-negative")
    if a.MAX_NORMAL < 0 or a.MAX_NORMAL > 1:
        raise ValueError("Invalid value for max_normal: " + str(a.MAX_NORMAL) + ". Max_normal must be between 0 and 1")
    if a.RATIO_DEV < 0:
        raise ValueError("Invalid value for ratio_dev: " + str(a.RATIO_DEV) + ". Ratio_dev must be non-negative.")
    if a.MIN_FRAC < 0 or a.MIN_FRAC > 1:
        raise ValueError("Invalid value for min_frac: " + str(a.MIN_FRAC) + ". Min_frac must be between 0 and 1.")
    prefix = a.OUTPUT_PREFIX
    if prefix is None:
        prefix = os.path.basename(a.QUERY_FILE).split(".")[0]
    num_intervals = a.NUM_INTERVALS
    if a.N == 3 and num_intervals == 100:
        num_intervals = 20                      # FileIO.py:170
    if not silent:
        print("=================================================")
        print("Arguments are:")
        print("\tQuery File:", a.QUERY_FILE)
        if a.N is not None:
            print("n:", a.N)
        print("\tk:", a.MAX_K)
        print("\ttau:", a.TAU)
        print("\tOutput Directory:", a.DIR)
        print("\tOutput Prefix:", prefix)
        if a.N == 2:
            print("\tMax Normal:", a.MAX_NORMAL)
        print("\tSearch engine: theta_amd (HIP, gfx950)")
        print("=================================================")
    return (a.QUERY_FILE, a.RESULTS, a.N, a.MAX_K, a.TAU, a.DIR, prefix, a.MAX_NORMAL, a.BOUND_HEURISTIC,
            a.NORMAL_BOUND_HEURISTIC, a.HEURISTIC_LB, a.HEURISTIC_UB, a.NUM_PROCESSES, a.BOUNDS_ONLY,
            not a.NO_MULTI_EVENT, a.FORCE, a.GET_VALUES, not a.NO_INTERVAL_SELECTION, num_intervals,
            a.READ_DEPTH_FILE, a.GRAPH_FORMAT, a.BAF, a.RATIO_DEV, a.MIN_FRAC, a.TUMOR_FILE, a.NORMAL_FILE,
            a.NO_CLUSTERING)


def read_interval_file(filename):
    """FileIO.py:386-446: ID chrm start end tumorCount normalCount [upperBound [lowerBound]]."""
    lengths, tumor, normal, upper, lower = [], [], [], [], []
    nline = 0
    with open(filename) as f:
        for line in f:
            if line.startswith("#"):
                continue
            cols = line.strip().replace(" ", "\t").split()
            nline += 1
            if len(cols) < 6 or len(cols) > 8:
                sys.stderr.write("Invalid input file format in interval file line #" + str(nline) + ":\n" + str(cols) +
                                 "\nToo few/many columns. Exiting...\n")
                sys.exit(1)
            lengths.append(int(cols[3]) - int(cols[2]))
            tumor.append(int(cols[4]))
            normal.append(int(cols[5]))
            upper.append(cols[6] if len(cols) > 6 else "X")
            lower.append(cols[7] if len(cols) > 7 else "X")
    if nline == 1:
        sys.stderr.write("Number of intervals must be greater than 1. Exiting...\n")
        sys.exit(1)
    if all(x == "X" for x in upper):
        upper = None
    if all(x == "X" for x in lower):
        lower = None
    return [lengths, tumor, normal, len(lengths), upper, lower]


def read_results_file(filename):
    """FileIO.py:484-504: the ':'-separated copy-number column of the first solution."""
    with open(filename) as f:
        lines = f.readlines()
    if lines and lines[0].startswith("#"):
        lines = lines[1:]
    if len(lines) == 0:
        print("ERROR: The result file provided appears to be empty. Exiting...")
        sys.exit(1)
    elif len(lines) > 1:
        print("WARNING: The results file contains more than one solution. THetA will use the first provided solution.")
    return lines[0].strip().split("\t")[2].split(":")


def write_out_result(directory, prefix, results, n):
    """FileIO.py:620-664: '#NLL<TAB>mu<TAB>C<TAB>p*'; C rows ':'-separated, tumour columns ','; -1 -> X."""
    path = os.path.join(directory, prefix + ".n" + str(n) + ".results")
    print("Writing results file to", path)
    with open(path, "w") as f:
        f.write("#NLL\tmu\tC\tp*\n")
        for C, mu, L, vals in results:
            rows = []
            for i in range(C.shape[0]):
                rows.append(",".join("X" if int(C[i][j]) == -1 else str(int(C[i][j])) for j in range(1, C.shape[1])))
            f.write(str(float(L)) + "\t" + ",".join(str(float(x)) for x in mu) + "\t" + ":".join(rows) + "\t" +
                    ",".join(v if isinstance(v, str) else str(float(v)) for v in vals) + "\n")
    return path


def write_out_bounds(directory, prefix, inputFile, upper_bounds, lower_bounds, n, order=None):
    """FileIO.py:733-784: the input file with UpperBound / LowerBound columns (X where not selected)."""
    with open(inputFile) as f:
        lines = f.readlines()
    out = os.path.join(directory, prefix + ".n" + str(n) + ".withBounds")
    print("Writing bounds file to", out)
    if "#" in lines[0]:
        lines = lines[1:]
    pos = {v: i for i, v in enumerate(order)} if order is not None else None
    with open(out, "w") as f:
        f.write("#ID\tchrm\tstart\tend\ttumorCount\tnormalCount\tUpperBound\tLowerBound\n")
        for i, line in enumerate(lines):
            f.write("\t".join(line.strip().split("\t")[:6]).strip())
            if pos is None:
                f.write("\t" + str(int(upper_bounds[i])) + "\t" + str(int(lower_bounds[i])))
            elif i in pos:
                f.write("\t" + str(int(upper_bounds[pos[i]])) + "\t" + str(int(lower_bounds[pos[i]])))
            else:
                f.write("\tX\tX")
            f.write("\n")
    return out


def write_out_N3_script(directory, prefix, inputFile):
    """FileIO.py:786-798: the command that continues with n=3 from the n=2 bounds and results."""
    filename = os.path.join(directory, prefix + ".RunN3.bash")
    print("Writing script to run N=3 to ", filename)
    bounds = os.path.join(directory, prefix + ".n2.withBounds")
    results = os.path.join(directory, prefix + ".n2.results")
    cmd = "python " + " ".join(sys.argv).replace("-n 2", "").replace(inputFile, bounds) + " -n 3" + " --RESULTS " + results
    with open(filename, "w") as f:
        f.write("#!/bin/bash\n")
        f.write(cmd)
    return filename
