#!/usr/bin/env python3
"""
Generates the golden fixtures under tests/golden/ by IMPORTING THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference).  The reference is Python 2; it is
converted mechanically with lib2to3 into a scratch directory OUTSIDE the repo (/tmp/theta_ref_py3)
and imported from there with three shims (time.clock, string.join, a stub `bnpy`).  Only DATA --
inputs and the reference's outputs -- is written into the repo; no reference source is copied.

    python tests/golden/make_golden.py [--skip-example]

Fixtures written:
  kat_calcallc.json     the reference's own L2/L3 known-answer pickles restated as JSON
  enum_order.json       enumeration order (n=2, n=3) on small instances + counts
  solve_n2.json         per-candidate Optimizer.solve tables, n=2, incl. degenerate cases
  solve_n3_small.json   per-candidate tables for small exhaustive n=3 instances
  solve_n3_m6k3.npz     the 21 050-candidate n=3 instance (accept flag, mu, NLL per candidate)
  best_synth.json       do_optimization_single `best` on seeded synthetic inputs
  example_n2.json       config 1 (example/Example.intervals -n 2 -k 3): search inputs + best
"""
import hashlib
import json
import os
import pickle
import shutil
import subprocess
import sys
import time
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/python"
SCRATCH = "/tmp/theta_ref_py3"

sys.path.insert(0, os.path.join(REPO, "oracle"))


def import_reference():
    if not os.path.isdir(SCRATCH):
        os.makedirs(SCRATCH)
        for f in os.listdir(REF):
            if f.endswith(".py"):
                shutil.copy(os.path.join(REF, f), os.path.join(SCRATCH, f))
        subprocess.run(["chmod", "-R", "u+w", SCRATCH], check=True)
        subprocess.run([sys.executable, "-m", "lib2to3", "-w", "-n", SCRATCH],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    import string
    time.clock = time.perf_counter
    string.join = lambda seq, sep=" ": sep.join(seq)
    sys.modules.setdefault("bnpy", types.ModuleType("bnpy"))
    import matplotlib
    matplotlib.use("Agg")
    sys.path.insert(0, SCRATCH)
    warnings.simplefilter("ignore")
    import RunTHetA  # noqa
    return RunTHetA


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=None, separators=(",", ":"))
    print("wrote", name)


def fl(x):
    """float -> JSON-safe (nan/inf as strings)."""
    x = float(x)
    if x != x:
        return "nan"
    if x in (float("inf"), float("-inf")):
        return "inf" if x > 0 else "-inf"
    return x


def synth(m, n, k, seed, tau=2):
    from theta_oracle import synth_counts, sort_r
    r, rN, L, C, mu = synth_counts(m, n, k, seed, tau)
    rs, rNs, order = sort_r(rN, r)
    return rs, rNs, order


def main():
    skip_example = "--skip-example" in sys.argv
    R = import_reference()
    from Enumerator import Enumerator
    from Optimizer import Optimizer
    import CalcAllC
    import TimeEstimate

    # ---- 1. KAT pickles -------------------------------------------------------------------
    def pk(name):
        with open(os.path.join(REF, name), "rb") as f:
            return pickle.load(f, encoding="latin1")
    l2a, l3a = pk("L2args.pkl"), pk("L3args.pkl")
    kat = {}
    for tag, a in (("L2", l2a), ("L3", l3a)):
        ent = {"mu": np.asarray(a[0]).tolist() if tag == "L3" else float(a[0]),
               "C": np.asarray(a[1]).tolist(), "m": int(a[2]), "r": np.asarray(a[3]).tolist()}
        if tag == "L3":
            ent["n"] = int(a[4])
        for br in ("branch", "master"):
            o = pk("%soutputs_%s.pkl" % (tag, br))
            ent[br] = {"nll": float(o[0]), "vals": [v if isinstance(v, str) else float(v) for v in o[1]]}
        kat[tag] = ent
    # Q10 golden (all-zero masked row -> NaN) computed by the reference's CalcAllC.L3
    c_nan = np.array([[2., 1, 1], [0, 0, 0], [2, 3, 2], [4, 2, 2]])
    c_ok = np.array([[2., 1, 1], [0, 3, 1], [2, 3, 2], [4, 2, 2]])
    rr = np.array([10, 5, 20, 30])
    kat["Q10"] = {"mu": [.2, .5, .3], "r": rr.tolist(),
                  "C_nan": c_nan.tolist(), "nll_nan": fl(CalcAllC.L3([.2, .5, .3], c_nan.copy(), 4, rr, 3)[0]),
                  "C_ok": c_ok.tolist(), "nll_ok": fl(CalcAllC.L3([.2, .5, .3], c_ok.copy(), 4, rr, 3)[0])}
    dump("kat_calcallc.json", kat)

    # ---- 2. enumeration order + counts ------------------------------------------------------
    def enum_all(n, m, k, tau, lb, ub):
        e = Enumerator(n, m, k, tau, list(lb), list(ub), True)
        out = []
        C = e.generate_next_C()
        while C is not False:
            out.append(C[:, 1:].astype(int).tolist())
            C = e.generate_next_C()
        return out, e

    enum_cases = []
    specs = [
        (2, 3, 2, [0, 0, 0], [2, 2, 2]),
        (2, 3, 3, [0, 1, 0], [3, 2, 3]),
        (2, 8, 4, [0, 0, 1, 1, 0, 2, 2, 1], [2, 3, 3, 2, 4, 4, 3, 4]),
        (2, 6, 5, [0, 0, 0, 2, 2, 3], [1, 2, 5, 5, 5, 5]),
        (3, 2, 2, [0, 0], [2, 2]),
        (3, 3, 2, [0, 0, 1], [2, 2, 2]),
        (3, 4, 3, [0, 0, 1, 1], [1, 2, 3, 3]),
        (3, 3, 4, [0, 1, 2], [2, 4, 4]),
    ]
    for n, m, k, lb, ub in specs:
        seq, e = enum_all(n, m, k, 2, lb, ub)
        ent = {"n": n, "m": m, "k": k, "tau": 2, "lb": lb, "ub": ub, "count": len(seq), "seq": seq}
        if n == 2:
            ent["count_ref"] = int(TimeEstimate.count_number_matrices_2(m, list(e.upper_bound), list(e.lower_bound)))
        else:
            ent["count_ref_upper"] = float(TimeEstimate.count_number_matrices_3(m, list(e.upper_bound), list(e.lower_bound), e))
            ent["rows"] = e.rows
        enum_cases.append(ent)
    # larger: hashed
    for n, m, k, lb, ub in [(3, 6, 3, [0] * 6, [3] * 6), (2, 12, 4, [0] * 12, [4] * 12),
                            (3, 5, 4, [0, 0, 1, 1, 2], [2, 3, 4, 4, 4]), (3, 4, 5, [0] * 4, [5] * 4)]:
        seq, e = enum_all(n, m, k, 2, lb, ub)
        h = hashlib.sha256()
        for c in seq:
            h.update(bytes(np.asarray(c, dtype=np.uint8).reshape(-1)))
        enum_cases.append({"n": n, "m": m, "k": k, "tau": 2, "lb": lb, "ub": ub, "count": len(seq),
                           "sha256_u8": h.hexdigest(), "first": seq[:50], "last": seq[-50:]})
    dump("enum_order.json", {"cases": enum_cases})

    # ---- 3. per-candidate solve tables, n=2 -------------------------------------------------
    def solve_table(n, m, k, lb, ub, r, rN, max_normal=1.0, want_vals=False):
        e = Enumerator(n, m, k, 2, list(lb), list(ub), True)
        o = Optimizer(list(r), list(rN), m, n, 2, upper_bound=max_normal)
        rows = []
        C = e.generate_next_C()
        while C is not False:
            s = o.solve(C)
            if s is None:
                rows.append(None)
            else:
                ent = [[fl(x) for x in s[0]], fl(s[1])]
                if want_vals:
                    ent.append([fl(v) for v in s[2]])
                rows.append(ent)
            C = e.generate_next_C()
        return rows

    n2 = {"cases": []}
    for (m, k, seed, mx) in [(10, 3, 1, 1.0), (12, 4, 2, 1.0), (10, 3, 3, 0.5), (7, 5, 4, 1.0)]:
        r, rN, order = synth(m, 2, k, seed)
        tab = solve_table(2, m, k, [0] * m, [k] * m, r, rN, mx, want_vals=(m <= 7))
        n2["cases"].append({"m": m, "k": k, "seed": seed, "max_normal": mx, "r": r, "rN": rN,
                            "lb": [0] * m, "ub": [k] * m, "table": tab})
    # degenerate cases
    r4, rN4 = [100, 200, 300, 400], [150, 200, 250, 300]
    o4 = Optimizer(r4, rN4, 4, 2, 2, upper_bound=1)
    o4h = Optimizer(r4, rN4, 4, 2, 2, upper_bound=0.5)
    deg = []
    for col, opt, mx in [([0, 0, 0, 0], o4, 1), ([0, 0, 0, 1], o4, 1), ([2, 2, 2, 2], o4, 1), ([1, 1, 1, 1], o4, 1),
                         ([0, 1, 2, 3], o4, 1), ([0, 1, 2, 3], o4h, 0.5), ([3, 3, 3, 3], o4, 1), ([0, 0, 2, 2], o4h, 0.5)]:
        C = np.zeros((4, 2)); C[:, 0] = 2; C[:, 1] = col
        s = opt.solve(C)
        deg.append({"col": col, "max_normal": mx,
                    "soln": None if s is None else [[fl(x) for x in s[0]], fl(s[1]), [fl(v) for v in s[2]]]})
    n2["degenerate"] = {"r": r4, "rN": rN4, "cases": deg}
    dump("solve_n2.json", n2)

    # ---- 4. per-candidate tables, n=3 small -------------------------------------------------
    n3 = {"cases": []}
    for (m, k, lb, ub, seed) in [(4, 2, [0] * 4, [2] * 4, 11), (5, 3, [0, 0, 1, 1, 1], [2, 3, 3, 3, 3], 12),
                                 (4, 4, [0, 0, 0, 1], [3, 4, 4, 4], 13)]:
        r, rN, order = synth(m, 3, k, seed)
        tab = solve_table(3, m, k, lb, ub, r, rN, 1.0, want_vals=(m == 4 and k == 2))
        # the Q1 first matrix [tau,0,0]*m as the reference driver evaluates it
        C0 = np.zeros((m, 3)); C0[:, 0] = 2
        s0 = Optimizer(list(r), list(rN), m, 3, 2).solve(C0)
        n3["cases"].append({"m": m, "k": k, "seed": seed, "r": r, "rN": rN, "lb": lb, "ub": ub, "table": tab,
                            "q1_first": None if s0 is None else [[fl(x) for x in s0[0]], fl(s0[1])]})
    dump("solve_n3_small.json", n3)

    # the 21 050-candidate instance
    m, k, seed = 6, 3, 21
    r, rN, order = synth(m, 3, k, seed)
    e = Enumerator(3, m, k, 2, [0] * m, [k] * m, True)
    o = Optimizer(list(r), list(rN), m, 3, 2)
    acc, mus, nlls, cs = [], [], [], []
    C = e.generate_next_C()
    t0 = time.time()
    while C is not False:
        s = o.solve(C)
        cs.append(C[:, 1:].astype(np.uint8))
        if s is None:
            acc.append(0); mus.append([np.nan] * 3); nlls.append(np.nan)
        else:
            acc.append(1); mus.append([float(x) for x in s[0]]); nlls.append(float(s[1]))
        C = e.generate_next_C()
    print("n3 m6k3: %d candidates in %.1fs" % (len(acc), time.time() - t0))
    np.savez_compressed(os.path.join(HERE, "solve_n3_m6k3.npz"), r=np.array(r), rN=np.array(rN),
                        lb=np.zeros(m, int), ub=np.full(m, k), accepted=np.array(acc, np.uint8),
                        mu=np.array(mus), nll=np.array(nlls), C=np.array(cs))

    # ---- 5. `best` on synthetic inputs --------------------------------------------------------
    def best_to_json(best):
        return [{"C": np.asarray(b[0]).tolist(), "mu": [fl(x) for x in b[1]], "nll": fl(b[2]),
                 "vals": [fl(v) for v in b[3]]} for b in best]

    bests = {"cases": []}
    for (n, m, k, lb, ub, seed, mx) in [
            (2, 12, 4, [0] * 12, [4] * 12, 31, 1.0),
            (2, 10, 3, [0] * 10, [3] * 10, 32, 0.5),
            (2, 9, 5, [0, 0, 0, 1, 1, 2, 2, 2, 3], [2, 3, 3, 4, 5, 5, 5, 5, 5], 33, 1.0),
            (3, 5, 3, [0] * 5, [3] * 5, 34, 1.0),
            (3, 6, 2, [0] * 6, [2] * 6, 35, 1.0),
            (3, 5, 4, [0, 0, 1, 1, 2], [2, 3, 4, 4, 4], 36, 1.0)]:
        r, rN, order = synth(m, n, k, seed)
        best = R.do_optimization_single(n, m, k, 2, list(lb), list(ub), list(r), list(rN), mx, list(order), True, False)
        bests["cases"].append({"n": n, "m": m, "k": k, "lb": lb, "ub": ub, "seed": seed, "max_normal": mx,
                               "r": r, "rN": rN, "order": order, "best": best_to_json(best)})
    dump("best_synth.json", bests)

    # ---- 6. config 1: Example.intervals -n 2 -k 3 -----------------------------------------------
    if not skip_example:
        import FileIO, DataTools, SelectIntervals
        fn = "/root/reference/example/Example.intervals"
        lengths, tumorCounts, normCounts, m, ub, lb = FileIO.read_interval_file(fn)
        k, tau, n = 3, 2, 2
        # RunTHetA.run_fixed_N (298-447) pre-steps, default flags, no SNP files
        sum_r, sum_rN = sum(tumorCounts), sum(normCounts)
        DataTools.set_total_read_counts(sum_r, sum_rN)
        order, lengths_s, tumor_s, norm_s = SelectIntervals.select_intervals_n2(lengths, tumorCounts, normCounts, m, k, False, 100)
        m_s = len(order)
        r, rN, sorted_index = DataTools.sort_r(norm_s, tumor_s)
        ubs, lbs = DataTools.calculate_bounds_heuristic(0.5, r, rN, m_s, tau, k)
        t0 = time.time()
        best = R.do_optimization_single(n, m_s, k, tau, list(lbs), list(ubs), list(r), list(rN), 1.0, list(sorted_index), True, False)
        dt = time.time() - t0
        print("example n2 search: %.1fs" % dt)
        dump("example_n2.json", {"m_all": m, "order": [int(x) for x in order], "m": m_s, "k": k, "tau": tau,
                                 "r": [int(x) for x in r], "rN": [int(x) for x in rN],
                                 "sorted_index": [int(x) for x in sorted_index],
                                 "ub": [int(x) for x in ubs], "lb": [int(x) for x in lbs], "max_normal": 1,
                                 "best": best_to_json(best), "ref_search_seconds": dt})


if __name__ == "__main__":
    main()
