"""
Randomised differential test of the COMMAND LINE against the reference itself, run in the build container (it needs
/root/reference; nothing here ships to the GPU box): seeded random `.intervals` files and flag combinations go through the
reference's RunTHetA (converted 2->3 outside the repo, tests/golden/make_golden.py) and through theta_amd.RunTHetA over the
oracle-backed stand-in device (tests/standin_device.py); exit codes and every output file are compared the way
tests/test_host_cli_cpu.py compares the committed cases.

    python tools/cli_differential.py [--cases 40] [--seed 1]      -> one line per case, divergences in full
"""
import argparse
import contextlib
import io
import os
import subprocess
import sys
import tempfile
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)

import numpy as np


def write_file(path, rng):
    m = int(rng.randint(7, 15))
    L = rng.randint(1_200_000, 20_000_000, m)
    for i in rng.choice(m, int(rng.randint(0, 3)), replace=False):
        L[i] = int(rng.randint(100_000, 4_000_000))                     # short rows (selection rules)
    rN = rng.poisson(L * 0.01)
    k = int(rng.randint(2, 5))
    c = rng.randint(0, k + 1, m)
    mu = float(rng.uniform(0.2, 0.7))
    p = rN * (2 * mu + c * (1 - mu)) + 1e-9
    p = p / p.sum()
    r = rng.multinomial(int(rN.sum() * rng.uniform(0.8, 1.3)), p)
    if rng.rand() < 0.25:
        r[int(rng.randint(0, m))] = 0
    if rng.rand() < 0.15:
        rN[int(rng.randint(0, m))] = 0
    with open(path, "w") as f:
        f.write("#ID\tchrm\tstart\tend\ttumorCount\tnormalCount\n")
        pos = 1
        for i in range(m):
            f.write("%d\t%d\t%d\t%d\t%d\t%d\n" % (i + 1, 1 + i // 5, pos, pos + L[i], r[i], rN[i]))
            pos += L[i] + 1
    return m


def random_args(path, rng):
    args = [path]
    two_stage = rng.rand() < 0.35
    if not two_stage:
        args += ["-n", "2"]
    args += ["-k", str(int(rng.randint(2, 5)) if not two_stage else int(rng.randint(2, 4)))]
    args += ["--NUM_INTERVALS", str(int(rng.randint(5, 9)) if not two_stage else int(rng.randint(5, 8)))]
    if two_stage:
        args += ["--FORCE"]
        if rng.rand() < 0.4:
            args += ["--NO_MULTI_EVENT"]
    else:
        u = rng.rand()
        if u < 0.2:
            args += ["-m", "%.2f" % rng.uniform(0.3, 0.9)]
        elif u < 0.35:
            args += ["--BOUND_HEURISTIC", "%.2f" % rng.uniform(0.2, 0.7)]
        elif u < 0.5:
            args += ["--NORMAL_BOUND_HEURISTIC", str(int(rng.randint(1, 3))), "--HEURISTIC_LB", "0.85", "--HEURISTIC_UB", "1.15"]
        elif u < 0.6:
            args += ["--GET_VALUES"]
    if rng.rand() < 0.2:
        args += ["--MIN_FRAC", "%.2f" % rng.uniform(0.0, 0.6)]
    return args


def collect(d):
    out = {}
    for fn in sorted(os.listdir(d)):
        if fn.startswith("c.") and (fn.endswith(".results") or fn.endswith(".withBounds") or fn.endswith(".likelihoods")):
            out[fn[2:]] = open(os.path.join(d, fn)).read()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    import make_golden
    import make_golden_cli
    make_golden.import_reference()
    warnings.simplefilter("ignore")
    import standin_device as sd
    import test_host_cli_cpu as T
    from theta_amd import _lib, RunTHetA
    ctx = sd.StandinContext()
    cache = {}

    def make(c, n, m, tau, r, rN, lb, ub, mx=1.0):
        p = sd.StandinProblem(c, n, m, tau, r, rN, lb, ub, mx)
        p._table = cache.setdefault((n, m, tau, tuple(map(int, r)), tuple(map(int, rN)), tuple(map(int, lb)), tuple(map(int, ub)), float(mx)), {})
        return p
    _lib.Problem = make
    _lib.default_context = lambda: ctx
    bad = 0
    for case in range(a.cases):
        rng = np.random.RandomState(a.seed * 1000 + case)
        work = tempfile.mkdtemp(prefix="theta_diff_")
        path = os.path.join(work, "in.intervals")
        write_file(path, rng)
        args = random_args(path, rng)
        dref, dmine = os.path.join(work, "ref"), os.path.join(work, "mine")
        os.makedirs(dref)
        os.makedirs(dmine)
        with open(os.path.join(dref, "_l.py"), "w") as f:
            f.write(make_golden_cli.LAUNCH)
        p = subprocess.run([sys.executable, os.path.join(dref, "_l.py")] + args + ["-d", dref, "-p", "c"], cwd=dref, capture_output=True, text=True)
        ref_rc, ref_files = p.returncode, collect(dref)
        crashed = "Traceback" in p.stderr
        rc = 0
        buf = io.StringIO()
        os.chdir(dmine)                           # (like the reference, the --GET_VALUES dump goes to the working directory)
        try:
            with contextlib.redirect_stdout(buf):
                RunTHetA.main(args + ["-d", dmine, "-p", "c"])
        except SystemExit as e:
            rc = e.code if isinstance(e.code, int) else (0 if e.code is None else 1)
        except AssertionError as e:
            print("case %d %s: stand-in refused (%s)" % (case, " ".join(args[1:]), e))
            continue
        except Exception as e:
            rc = "exception %r" % (e,)
        mine_files = collect(dmine)
        why = []
        if crashed:
            why.append("reference raised: " + p.stderr.strip().splitlines()[-1][:120])
        if rc != ref_rc and not crashed:
            why.append("exit code %r vs reference %r" % (rc, ref_rc))
        if sorted(mine_files) != sorted(ref_files) and not crashed:
            why.append("files %s vs reference %s" % (sorted(mine_files), sorted(ref_files)))
        for suffix in sorted(set(mine_files) & set(ref_files)):
            try:
                if suffix.endswith(".withBounds"):
                    assert T._rows(mine_files[suffix]) == T._rows(ref_files[suffix])
                elif suffix.endswith(".results"):
                    a_, b_ = os.path.join(work, "a." + suffix), os.path.join(work, "b." + suffix)
                    open(a_, "w").write(mine_files[suffix])
                    open(b_, "w").write(ref_files[suffix])
                    T._compare_results_nan_aware(a_, b_)
                else:
                    T._compare_likelihoods(mine_files[suffix], ref_files[suffix])
            except AssertionError as e:
                why.append("%s differs: %s" % (suffix, str(e)[:200]))
        status = "ok" if not why else ("REFERENCE CRASH" if crashed and len(why) == 1 else "DIVERGENCE")
        bad += status == "DIVERGENCE"
        print("case %d [%s] %s%s" % (case, status, " ".join(args[1:]), "".join("\n    " + w for w in why)), flush=True)
        if status == "DIVERGENCE":
            print("    kept: " + work)
    print("divergences: %d of %d" % (bad, a.cases))


if __name__ == "__main__":
    main()
