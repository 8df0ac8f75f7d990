#!/usr/bin/env python3
"""
Golden OUTPUT FILES of the reference's command line (RunTHetA.py main) -- data only.

Runs the reference (converted 2->3 outside the repo, see make_golden.py) as a subprocess on
  * example/Example.intervals -n 2 -k 3                         (BASELINE config 1)
  * a seeded 14-interval synthetic file, -n 2, default flags
  * the same file through the default two-stage pipeline (no -n, --FORCE, ONE process: n=2, then n=3 on the intervals and
    bounds derived from the FIRST n=2 solution -- 1 369 938 candidate matrices, ~55 minutes of the reference --, then model
    selection); prefix syn14d.  (With --NUM_PROCESSES 8 the reference lists the two tied n=2 solutions in the other order
    -- find_mins concatenates the workers' lists, RunTHetA.py:107-122 -- so its n=3 stage starts from different bounds.
    The GPU driver mirrors the single-process order.)
  * the same file through the two-stage pipeline with -k 3 --NUM_INTERVALS 9 (prefix syn14s; n=3 stage: 7 intervals, 3 576 matrices): small enough for the
    CPU oracle, so that tests/test_host_cli_cpu.py can drive the command line over the stand-in device without a GPU
    (`--only-small` regenerates just this one);
and copies the resulting .withBounds / .results files (and the synthetic inputs) to tests/golden/cli/.
"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden  # noqa  (import_reference converts the sources into /tmp/theta_ref_py3)

LAUNCH = r'''
import sys, time, string, types
time.clock = time.perf_counter
string.join = lambda seq, sep=" ": sep.join(seq)
sys.modules.setdefault("bnpy", types.ModuleType("bnpy"))
import matplotlib; matplotlib.use("Agg")
sys.path.insert(0, "%s")
import RunTHetA
RunTHetA.main()
''' % make_golden.SCRATCH


def write_intervals(path, seed, m, with_bounds=None):
    rng = np.random.RandomState(seed)
    L = rng.randint(2_000_000, 20_000_000, m)
    rN = rng.poisson(L * 0.01)
    c = rng.randint(0, 4, m)
    mu = 0.35
    p = rN * (2 * mu + c * (1 - mu))
    p = p / p.sum()
    r = rng.multinomial(int(rN.sum() * 1.1), p)
    with open(path, "w") as f:
        f.write("#ID\tchrm\tstart\tend\ttumorCount\tnormalCount" + ("\tUpperBound\tLowerBound" if with_bounds else "") + "\n")
        pos = 1
        for i in range(m):
            line = "%d\t1\t%d\t%d\t%d\t%d" % (i + 1, pos, pos + L[i], r[i], rN[i])
            if with_bounds:
                line += "\t%d\t%d" % with_bounds[i]
            f.write(line + "\n")
            pos += L[i] + 1


def run(args, cwd):
    launcher = os.path.join(cwd, "_launch.py")
    with open(launcher, "w") as f:
        f.write(LAUNCH)
    subprocess.run([sys.executable, launcher] + args, cwd=cwd, check=True, stdout=subprocess.DEVNULL)


def main():
    make_golden.import_reference()
    out = os.path.join(HERE, "cli")
    os.makedirs(out, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="theta_cli_")
    # synthetic inputs
    syn = os.path.join(out, "syn14.intervals")
    write_intervals(syn, 77, 14)
    # (bounds given in the file + --NO_INTERVAL_SELECTION reach the reference's Enumerator as strings, which
    #  neither Python 2 nor 3 survives for n=3 -- Enumerator.py:254 -- so there is no n=3 CLI golden)
    run([syn, "-k", "3", "--NUM_INTERVALS", "9", "-d", tmp, "-p", "syn14s", "--FORCE"], tmp)
    if "--only-small" in sys.argv:
        sys.argv += ["--skip-n3", "--skip-example"]
    else:
        run([syn, "-n", "2", "-k", "3", "-d", tmp, "-p", "syn14"], tmp)
    if "--skip-n3" not in sys.argv:
        run([syn, "-k", "3", "-d", tmp, "-p", "syn14d", "--FORCE"], tmp)
    if "--skip-example" not in sys.argv:
        run(["/root/reference/example/Example.intervals", "-n", "2", "-k", "3", "-d", tmp, "-p", "Example"], tmp)
    for f in sorted(os.listdir(tmp)):
        if f.endswith(".results") or f.endswith(".withBounds") or f.endswith(".RunN3.bash"):
            shutil.copy(os.path.join(tmp, f), os.path.join(out, f))
            print("golden:", f)


if __name__ == "__main__":
    main()
