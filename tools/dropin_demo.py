"""
INTEGRATION.md section 2(a), demonstrated in the build container (needs /root/reference; nothing here ships): the REFERENCE's
own command line -- its argument parser, interval selection, bounds, calc_all_c_*, model selection and file writers, converted
2->3 outside the repo (tests/golden/make_golden.py) -- with the two driver functions replaced by theta_amd's, exactly the
two-line change the document describes.  No GPU here, so the library's Python surface is the oracle-backed stand-in
(tests/standin_device.py); on a GPU box the same two lines bind the HIP library.  The files the patched reference writes are
compared with the files the unpatched reference wrote (tests/golden/cli/syn14s.*, tests/golden/cli_matrix.json).

    python tools/dropin_demo.py [--operators]       -> one line per command, "identical" or the difference
With --operators it is section 2(b) instead: the reference's own driver loop keeps running, over theta_amd's Enumerator,
Optimizer and CalcAllC.L2 / L3 classes.
"""
import contextlib
import io
import json
import os
import sys
import tempfile
import time
import string
import types
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
warnings.simplefilter("ignore")

import make_golden

make_golden.import_reference()
# the shims the converted reference needs (tests/golden/make_golden_cli.py: LAUNCH)
time.clock = time.perf_counter
string.join = lambda seq, sep=" ": sep.join(seq)
sys.modules.setdefault("bnpy", types.ModuleType("bnpy"))
import matplotlib

matplotlib.use("Agg")
sys.path.insert(0, make_golden.SCRATCH)
import RunTHetA as REF                      # the reference's module

import standin_device as sd
import test_host_cli_cpu as T
from theta_amd import _lib, search as S

ctx = sd.StandinContext()
cache = {}


def make(c, n, m, tau, r, rN, lb, ub, mx=1.0):
    p = sd.StandinProblem(c, n, m, tau, r, rN, lb, ub, mx)
    p._table = cache.setdefault((n, m, tau, tuple(map(int, r)), tuple(map(int, rN)), tuple(map(int, lb)), tuple(map(int, ub)), float(mx)), {})
    return p


_lib.Problem = make
_lib.default_context = lambda: ctx

MODE = "operators" if "--operators" in sys.argv else "driver"
if MODE == "driver":
    # ---- the two lines of INTEGRATION.md 2(a) --------------------------------------------------------------------------
    REF.do_optimization_single = S.do_optimization_single
    REF.do_optimization = S.do_optimization
else:
    # ---- the reference's OWN driver loop (RunTHetA.py:173-220) over this repository's operator classes
    import CalcAllC as REF_CALC
    import TimeEstimate as REF_TIME
    from theta_amd import CalcAllC as MY_CALC
    from theta_amd.Enumerator import Enumerator as MyEnumerator
    from theta_amd.Optimizer import Optimizer as MyOptimizer
    REF.Enumerator = REF_TIME.Enumerator = MyEnumerator
    REF.Optimizer = REF_TIME.Optimizer = MyOptimizer
    REF_CALC.L2, REF_CALC.L3 = MY_CALC.L2, MY_CALC.L3

GOLD = os.path.join(ROOT, "tests", "golden")
matrix = json.load(open(os.path.join(GOLD, "cli_matrix.json")))
cases = {k: v for k, v in matrix.items() if v["rc"] == 0}
S.pre = "c"          # (driver mode: the replaced driver writes the --GET_VALUES dump itself; the reference sets its own global, RunTHetA.py:307)
bad = 0
for name in sorted(cases):
    gold = cases[name]
    d = tempfile.mkdtemp(prefix="theta_dropin_")
    argv = [os.path.join(GOLD, a) if a.startswith("cli" + os.sep) else a for a in gold["args"]] + ["-p", "c", "-d", d]
    sys.argv = ["RunTHetA.py"] + argv
    os.chdir(d)                               # (the reference writes its --GET_VALUES dump to the working directory)
    rc = 0
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            REF.main()
    except SystemExit as e:
        rc = e.code if isinstance(e.code, int) else (0 if e.code is None else 1)
    why = []
    if rc != gold["rc"]:
        why.append("exit code %r" % rc)
    for suffix, text in gold["files"].items():
        path = os.path.join(d, "c." + suffix)
        if not os.path.exists(path):
            why.append("missing " + suffix)
            continue
        mine = open(path).read()
        try:
            if suffix.endswith(".withBounds"):
                assert T._rows(mine) == T._rows(text)
            elif suffix == "likelihoods":
                T._compare_likelihoods(mine, text)
            else:
                ref_path = os.path.join(d, "ref." + suffix)
                open(ref_path, "w").write(text)
                T._compare_results_nan_aware(path, ref_path)
        except AssertionError as e:
            why.append("%s differs %s" % (suffix, str(e)[:150]))
    bad += bool(why)
    print("%-28s %s" % (name, "identical (%s)" % ", ".join(sorted(gold["files"])) if not why else "DIFFERENT: " + "; ".join(why)), flush=True)
print("[%s replaced] %d of %d commands of the patched reference reproduce the unpatched reference's files" % (MODE, len(cases) - bad, len(cases)))
