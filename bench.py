#!/usr/bin/env python3
"""
bench.py -- candidate C-matrices evaluated per second (BASELINE.json metric) on MI355X.  No torch: numpy + the C ABI.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ... -- the launcher only
     provides RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT; the collectives are the library's own RCCL
     communicator, theta_comm_create)

Workload (config.workload): BASELINE config 4's shape -- synthetic m=50 intervals, n=3, k=6, full bounds [0,6] (2.6e38
admissible matrices, inexhaustible) -- searched as rank ranges of the reference's enumeration order.  One "step" = one
theta_search call over `--batch` consecutive candidates (enumerate + solve + NLL + arg-min on the device).  Each GPU
owns the contiguous shard [N*g/G, N*(g+1)/G) of the rank space and its steps are spread evenly through that shard.  Weak
scaling: per-GPU work is fixed.  With N > 1 the per-shard finalists are merged with ONE RCCL exchange
(theta_exchange_finalists) inside the timed region.

Prints ONE JSON line (rank 0).
  value      candidates PASSED THROUGH THE FULL SOLVE by all GPUs / max-over-ranks wall time of the K timed steps -- SURVEY 8(d)'s
             definition at the reference's precision AND tolerance: every candidate of the range is generated, iterated in FP64
             until the point it is left at has lambda^2 / sum r < 1e-12 -- by the self-concordance certificate of its last Newton
             step -- and valued; none is dismissed by a bound (leg "full_solve_f64_tight_certified": n3_no_dismiss + n3_force_f64 +
             n3_conv_l2 = certified_conv_l2, the sieve kernel's double instantiation).  What a candidate is left at is WITNESSED:
             theta_search_witness + tests/test_gpu_round5.py check decrement, mu (1e-6) and NLL of sampled candidates against the
             oracle; `witness` on the line is the kernel's own record summary for one timed range.  dtype "f64".
             (Round 4's headline, the COARSE tolerance 1e-4 -- mu to ~1e-3 --, is leg "full_solve_f64".)
  roofline   dominant kernel n3_sieve_kernel<6, double> (n3_sieve.hip: burst enumeration + one evaluation shared by the children
             of a last-level node), vector-ALU bound -- candidates are generated on chip, ~0 algorithmic HBM bytes.  `achieved` =
             FLOP executed by the likelihood arithmetic (counted in-kernel from the evaluations actually run) / HIP-event time
             of the sieve + finish kernels; `peak` = the FP64 vector peak (78.6 TFLOP/s; the v_log_f32 of the screened value is
             weighted with the FP32 peak).  Executed FLOP FALL when the algorithm improves (sharing an evaluation between
             siblings removed half of them), so `frac` is a statement about the kernel, not about progress.
             legs (N = 1, after the timed region, on ALL of its rank ranges; each with its own kernel time, FLOP and frac):
               full_solve_f64   the headline, again as a leg record (per-step kernel ms min / median / max, counters)
               full_solve_f64_tight  the same at the tight tolerance (lambda^2 / sum r < 1e-12: every candidate's mu to 1e-6)
               full_solve_f64_tight_certified  the same guarantee met by certificate (the decrement of the point a candidate is LEFT at
                                     is bounded below 1e-12 by the self-concordance bound of its last Newton step)
               full_solve_f32   the same in packed single precision (n3_no_dismiss)
               search           the shipped branch-and-bound: a candidate whose rigorous lower bound lies beyond the window of
                                the running minimum is finished after one shared packed-FP32 evaluation ("searched", not
                                "fully solved": its rate is a rider, never the headline)
             traffic: HBM bytes per launch, measured by two rocprofv3 --pmc passes of this same command on three steps
             (FETCH_SIZE, WRITE_SIZE; gfx950 corrections of MI355X_MICROARCH.md), or null when that is not possible.
  cpu_baseline  the CPU oracle (oracle/theta_oracle.py, a port of the reference's Python) on this host's cores, bounded sample.
"""
import argparse
import json
import multiprocessing as mp
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np

FP64_VECTOR_PEAK_TFLOPS = 78.6     # MI355X FP64 vector peak (AMD spec; = 256 CU x 4 SIMD x 16 FMA lanes x 2 x 2.4 GHz)
FP32_VECTOR_PEAK_TFLOPS = 157.3    # MI355X FP32 vector peak (packed v_pk_fma_f32: two FP32 FMAs per FP64 lane)
FP64_MFMA_PEAK_TFLOPS = 78.6       # MI355X FP64 matrix peak (dense)
HBM_PEAK_GBS = 8000.0

M, N_POP, K_MAX, TAU, SEED = 50, 3, 6, 2, 4242
MU_TOL = 1e-6                      # north_star: |delta mu| < 1e-6
DOMINANT_KERNEL = "n3_sieve_kernel"         # (n3_sieve.hip; the headline runs its <6, double> instantiation); rocprofv3 summaries under profiles/
# the legs: options of the search instance, arithmetic, kernel instantiation
LEGS = {"full_solve_f64": ({"n3_no_dismiss": 1, "n3_force_f64": 1}, "f64", "n3_sieve_kernel<6, double>"),
        # the same with the TIGHT tolerance: every candidate iterated until an evaluation finds lambda^2 / sum r < 1e-12 (then the
        # step): each candidate's mu within 1e-6 of its optimum -- north_star's tolerance for the CHOSEN candidate, here for all
        "full_solve_f64_tight": ({"n3_no_dismiss": 1, "n3_force_f64": 1, "n3_conv_l2": 1e-12}, "f64", "n3_sieve_kernel<6, double>"),
        # the tight tolerance met by CERTIFICATE: the library leaves a candidate one Newton step beyond the evaluation that finds the
        # tolerance.  The restricted likelihood is self-concordant with parameter 2 / sqrt(Rmin), so a full step from a point with
        # t = lambda / sqrt(Rmin) <= 0.1 ends at lambda'^2 / sum r <= 1.53 (sum r / Rmin) (lambda^2 / sum r)^2 (certified_conv_l2
        # below; tests/test_certified_tolerance_cpu.py).  An evaluation that finds lambda^2 / sum r below
        # sqrt(1e-12 Rmin / (1.53 sum r)) therefore leaves the candidate at a point whose decrement is certified below 1e-12 --
        # the same guarantee as full_solve_f64_tight (mu within 1e-6), without the evaluation that only confirms it
        # Round 6: the tolerance is north_star's OWN -- mu within 1e-6 -- as a per-candidate certificate (option "n3_mu_tol", n3_sieve.hip:
        # sv_mu_limit): besides the decrement, an evaluation only counts as the last one where the point one Newton step further is
        # bounded within 1e-6 (less a 30 % margin) of the optimum in every component of mu, from the smaller eigenvalue of the tangent
        # Hessian and the Jacobian of nu -> mu.  Round 5's leg (the decrement alone: mu to 8e-7 on this instance, to 1e-5 on a dozen
        # intervals) is kept as full_solve_f64_l2_certified.
        "full_solve_f64_tight_certified": ({"n3_no_dismiss": 1, "n3_force_f64": 1, "n3_conv_l2": "certified", "n3_mu_tol": MU_TOL}, "f64", "n3_sieve_kernel<6, double>"),
        "full_solve_f64_l2_certified": ({"n3_no_dismiss": 1, "n3_force_f64": 1, "n3_conv_l2": "certified"}, "f64", "n3_sieve_kernel<6, double>"),
        "full_solve_f32": ({"n3_no_dismiss": 1}, "f32+f64", "n3_sieve_kernel<6, float>"),
        "search": ({}, "f32+f64", "n3_sieve_kernel<6, float>")}


def certified_conv_l2(r, final_l2=1e-12):
    """Largest lambda^2 / sum r at an evaluation from which ONE full Newton step is certified to end below `final_l2`.
    f = -sum R_i log q_i(u) over the terms of a candidate (R_i >= min r: a term aggregates whole intervals) is self-concordant
    after division by Rmin; with t = lambda / sqrt(Rmin) the step's decrement obeys t' <= (t / (1 - t))^2, i.e.
    l2' <= l2^2 (sum r / Rmin) / (1 - t)^4 <= 1.53 l2^2 sum r / Rmin for t <= 0.1."""
    r = np.asarray(r, dtype=np.float64)
    ror = float(r.sum() / r[r > 0].min())
    return min(float(np.sqrt(final_l2 / (1.53 * ror))), 0.01 / ror)


def synth(seed=SEED, m=M, n=N_POP, k=K_MAX):
    """Seeded synthetic input of SURVEY.md section 8(d) (generator owned by this repo), sorted like sort_r."""
    rng = np.random.RandomState(seed)
    L = rng.randint(2_000_000, 20_000_000, m)
    rN = np.maximum(rng.poisson(L * 0.01), 1)
    C = np.full((m, n), float(TAU))
    for j in range(1, n):
        C[:, j] = rng.randint(0, k + 1, m)
    mu = rng.dirichlet(np.ones(n) * 4)
    p = (C * rN[:, None]) @ mu
    p /= p.sum()
    r = rng.multinomial(int(rN.sum() * 1.2), p)
    ratio = (r / rN) * (rN.sum() / r.sum())
    order = np.argsort(ratio, kind="stable")
    return [int(x) for x in r[order]], [int(x) for x in rN[order]], [int(x) for x in order]


# ---- CPU baseline leg (oracle) -------------------------------------------------------------------
_W = {}


def _cpu_init(r, rN):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import theta_oracle as orc
    _W["orc"], _W["r"], _W["rN"] = orc, r, rN


def _cpu_init_quiet(r, rN):
    """Worker start: one BLAS / OpenMP thread per process (P processes x T library threads on P cores is what round 4's
    3 %-efficient figure measured), then the oracle."""
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
        os.environ[k] = "1"
    try:
        import threadpoolctl
        _W["tp"] = threadpoolctl.threadpool_limits(1)
    except Exception:
        pass
    _cpu_init(r, rN)


def _cpu_work(args):
    cands, budget = args
    orc = _W["orc"]
    done = 0
    t0 = time.time()
    for c in cands:
        if time.time() - t0 > budget:
            break
        orc.solve_n3(orc.rows_to_matrix_n3([tuple(x) for x in c], TAU), _W["r"], _W["rN"])
        done += 1
    return done, time.time() - t0


def host_cores():
    """Cores this process may really use: the affinity mask, capped by the cgroup's CPU quota (os.cpu_count() reports the
    machine's, which a container seldom owns)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]           # cgroup v2
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())     # cgroup v1
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n, (os.cpu_count() or 1), quota


def _restatement_lib():
    """The build's own C++ restatement of the reference's per-candidate procedure (theta_amd/csrc/n3_refbfgs.hpp: hybrj, the BFGS
    decision, M3's hybrd, L3's sums -- the function theta_solve_batch runs on the device), compiled for the host
    (tools/hybrj_check.cpp; __graft_entry__.build() leaves it in build_ab/)."""
    import ctypes as C
    so = os.path.join(ROOT, "build_ab", "libhybrj_check.so")
    if not os.path.exists(so):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-builtin-pow", "-shared", "-fPIC",
                               os.path.join(ROOT, "tools", "hybrj_check.cpp"), "-o", so])
    lib = C.CDLL(so)
    dp, u8p = C.POINTER(C.c_double), C.POINTER(C.c_uint8)
    lib.hybrj_check_table.argtypes = [C.c_int, C.c_int, C.c_int, dp, dp, u8p, u8p, dp, dp]
    return lib


def cpu_restatement(cands, r, rN, cores, budget_s=6.0):
    """"the build's own CPU restatement, timed single-thread and all-cores".  ctypes releases the GIL, so the
    all-core leg is `cores` threads of one process, each on its own chunk of the sample."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    lib = _restatement_lib()
    dp, u8p = C.POINTER(C.c_double), C.POINTER(C.c_uint8)
    rr = np.ascontiguousarray(r, np.float64)
    rn = np.ascontiguousarray(rN, np.float64)
    cands = np.ascontiguousarray(cands, np.uint8)
    m = cands.shape[1]

    def run(chunk, budget):
        """chunk through the procedure in blocks of 256 until the budget is spent; returns (candidates done, seconds)"""
        ok, mu, nll = np.zeros(256, np.uint8), np.zeros((256, 3)), np.zeros(256)
        done, t0 = 0, time.time()
        while time.time() - t0 < budget:
            blk = chunk[(done % max(len(chunk) - 255, 1)):][:256]
            lib.hybrj_check_table(len(blk), m, TAU, rr.ctypes.data_as(dp), rn.ctypes.data_as(dp), blk.ctypes.data_as(u8p),
                                  ok.ctypes.data_as(u8p), mu.ctypes.data_as(dp), nll.ctypes.data_as(dp))
            done += len(blk)
        return done, time.time() - t0

    d1, t1 = run(cands, budget_s / 3)
    chunks = [np.ascontiguousarray(cands[i::cores]) for i in range(cores)]
    chunks = [c for c in chunks if len(c)] or [cands]
    with ThreadPoolExecutor(len(chunks)) as ex:
        res = list(ex.map(lambda c: run(c, budget_s * 2 / 3), chunks))
    dn, tn = sum(x[0] for x in res), max(x[1] for x in res)
    one, allc = d1 / t1, dn / tn
    return {"value": allc, "unit": "candidates/s", "threads": len(chunks), "per_thread": one, "parallel_efficiency": allc / (len(chunks) * one),
            "what": "n3_ref_solve (theta_amd/csrc/n3_refbfgs.hpp, the per-candidate procedure of theta_solve_batch) compiled for the host "
                    "with g++ -O2 (tools/hybrj_check.cpp): %d candidates in %.1f s on one thread, %d in %.1f s on %d threads" % (d1, t1, dn, tn, len(chunks))}


def cpu_baseline(cands, r, rN, budget_s=15.0):
    """Oracle (port of the reference's Optimizer.solve) on the host cores, bounded by `budget_s` seconds: one process (the
    reference's default NUM_PROCESSES = 1), then one process per usable core (its multiprocessing path, RunTHetA.py:96-105)."""
    cores, machine, quota = host_cores()
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
        os.environ[k] = "1"                            # (inherited by the workers; the libraries already loaded here are limited below)
    try:
        import threadpoolctl
        limiter = threadpoolctl.threadpool_limits(1)
    except Exception:
        limiter = None
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import theta_oracle as orc
    t0 = time.time()
    n1 = 0
    for c in cands:
        orc.solve_n3(orc.rows_to_matrix_n3([tuple(x) for x in c], TAU), r, rN)
        n1 += 1
        if time.time() - t0 > budget_s / 3:
            break
    per_process = n1 / (time.time() - t0)
    chunks = [cands[i::cores] for i in range(cores)]
    ctx = mp.get_context("fork")
    with ctx.Pool(cores, initializer=_cpu_init_quiet, initargs=(r, rN)) as pool:
        res = pool.map(_cpu_work, [(ch, budget_s * 2 / 3) for ch in chunks], chunksize=1)
    if limiter is not None:
        limiter.restore_original_limits()
    done = sum(x[0] for x in res)
    dt = max(x[1] for x in res)          # workers run concurrently; process start-up is not charged to the CPU
    eff = (done / dt) / (cores * per_process)
    sample = ("%d candidates drawn from the GPU run's own rank ranges, oracle.solve_n3 (scipy fsolve/BFGS): one process %.0f/s "
              "(%d candidates), then %d concurrent single-threaded processes -- one per usable core (affinity %d, cgroup quota %s, "
              "machine %d) -- %.1f s of solving each: parallel efficiency %.2f" %
              (done, per_process, n1, cores, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else cores,
               "none" if quota is None else "%.1f" % quota, machine, dt, eff))
    if eff < 0.5:
        sample += " (below 0.5: the cores are shared with other tenants of the host, or SMT siblings count as cores)"
    out = {"value": done / dt, "unit": "candidates/s", "cores": cores, "kind": "port", "per_process": per_process,
           "parallel_efficiency": eff, "sample": sample}
    try:
        out["restatement"] = cpu_restatement(cands, r, rN, cores)
    except Exception as ex:
        out["restatement"] = {"error": str(ex)[:200]}
    return out


# ---- the legs --------------------------------------------------------------------------------------
class Leg:
    """K timed theta_search calls over the rank ranges of the bench, chained like the pieces of one job (each starts from
    the minimum found so far, theta_problem_hint)."""

    def __init__(self, problem, begins, batch, window):
        self.p, self.begins, self.batch, self.window = problem, begins, batch, window
        self.running = float("inf")
        self.best = None
        self.tot = {k: 0 for k in ("evaluated", "accepted", "dismissed", "flops", "flops_f32", "terms", "iterations", "survivors",
                                   "fallback_candidates", "redo_flops", "redo_flops_f32", "degenerate", "kernel_launches", "pruned")}
        self.kernel_ms = self.setup_ms = self.redo_ms = 0.0
        self.step_ms = []
        self.launches = 0

    def step(self, i, count=True):
        b = self.begins[i]
        if not self.running < float("inf"):
            # the job's first step has no minimum to start from yet: a short search (2^16 candidates of the same range) gives
            # it one that some candidate really attains -- Problem.search's own probe would launch 16 of them
            res0 = self.p.search(b, b + min(1 << 16, self.batch), window=0.0)
            if len(res0["nll"]):
                self.running = float(res0["nll"].min())
        if self.running < float("inf"):
            self.p.hint(self.running)
        res = self.p.search(b, b + self.batch, window=self.window)
        if len(res["nll"]):
            self.running = min(self.running, float(res["nll"].min()))
        if count:
            st = res["stats"]
            for k in self.tot:
                self.tot[k] += st[k]
            # every candidate is counted once (a slice redone by the fused kernel is reported apart, stats.redo_*)
            assert st["evaluated"] == self.batch and st["dismissed"] <= st["evaluated"], (st["evaluated"], st["dismissed"])
            self.kernel_ms += st["kernel_ms"]
            self.redo_ms += st["redo_kernel_ms"]
            self.step_ms.append(st["kernel_ms"] + st["redo_kernel_ms"])
            self.setup_ms += st["setup_ms"]
            self.launches += 1
            if os.environ.get("THETA_BENCH_VERBOSE"):
                print("step %d: kernel %.2f ms, evaluated %d, dismissed %d, survivors %d, redone by the fused kernel %d, iters %d, diag %s" %
                      (i, st["kernel_ms"], st["evaluated"], st["dismissed"], st.get("survivors", 0), st.get("fallback_candidates", 0),
                       st["iterations"], st["phase_cycles"]), file=sys.stderr)
            if len(res["nll"]) and (self.best is None or res["nll"].min() < self.best["nll"].min()):
                self.best = res
        return res

    def summary(self, wall_s, name, dtype, kernel=DOMINANT_KERNEL):
        f64, f32, ev = float(self.tot["flops"]), float(self.tot["flops_f32"]), float(self.tot["evaluated"])
        k_s = self.kernel_ms * 1e-3
        ach = (f64 + f32) / k_s / 1e12 if k_s > 0 else 0.0
        # time-weighted peak of the executed mix: an FP64 op costs two packed-FP32 slots
        peak = (f64 + f32) / (f64 / FP64_VECTOR_PEAK_TFLOPS + f32 / FP32_VECTOR_PEAK_TFLOPS) if f64 + f32 > 0 else FP32_VECTOR_PEAK_TFLOPS
        sm = sorted(self.step_ms) or [0.0]
        return {"leg": name, "dtype": dtype, "kernel": kernel, "launches": self.launches, "candidates_per_launch": self.batch,
                "value": ev / wall_s if wall_s > 0 else 0.0, "unit": "candidates/s", "wall_ms_per_launch": 1e3 * wall_s / max(self.launches, 1),
                "kernel_ms_per_launch": self.kernel_ms / max(self.launches, 1), "kernel_candidates_per_s": ev / k_s if k_s > 0 else 0.0,
                "step_kernel_ms": {"min": sm[0], "median": sm[len(sm) // 2], "max": sm[-1]},
                # a step = one theta_search call = a short first slice + the bulk: two launches of the
                # sieve kernel, each followed by the finish kernel; per LAUNCH of the dominant kernel (what rocprofv3 --stats
                # averages, profiles/r3/bench_search_launches.csv lists every one):
                "kernel_launches": self.tot["kernel_launches"],
                "kernel_ms_per_kernel_launch": self.kernel_ms / max(self.tot["kernel_launches"], 1),
                "executed_flop_per_kernel_launch": (f64 + f32) / max(self.tot["kernel_launches"], 1),
                "survivors": self.tot["survivors"], "fallback_candidates": self.tot["fallback_candidates"],
                "redo_kernel_ms": self.redo_ms, "redo_flop": float(self.tot["redo_flops"] + self.tot["redo_flops_f32"]),
                "degenerate": self.tot["degenerate"],
                "executed_flop_per_launch": (f64 + f32) / max(self.launches, 1), "fp64_flop_share": f64 / max(f64 + f32, 1.0),
                "flop_per_candidate": (f64 + f32) / max(ev, 1.0), "achieved": ach, "peak": peak, "unit_roofline": "TFLOP/s",
                "frac": ach / peak, "newton_iters_per_candidate": self.tot["iterations"] / max(ev, 1.0),
                "terms_per_iteration": self.tot["terms"] / max(self.tot["iterations"], 1),
                "dismissed_fraction": self.tot["dismissed"] / max(ev, 1.0),
                # (search mode: candidates of whole prefixes finished by the prefix bound -- no evaluation of their own; 0 in the full-solve legs)
                "prefix_bound_fraction": self.tot["pruned"] / max(ev, 1.0),
                "accepted_fraction_of_solved": self.tot["accepted"] / max(ev - self.tot["dismissed"], 1.0),
                # SURVEY.md 8(d)'s per-unit figure (a per-interval FP64 Newton solve, ~160 FP64 op per interval): what this
                # leg's throughput would cost a kernel doing THAT work -- for the f64 full solve the distance to `achieved`
                # is what the group tile and the warm start save, not ALU efficiency
                "survey_flop_per_candidate": 160.0 * M,
                "survey_equivalent_tflops": 160.0 * M * ev / k_s / 1e12 if k_s > 0 else 0.0}


def measure_traffic(args):
    """
    HBM bytes per launch of the search kernel: two rocprofv3 --pmc passes (FETCH_SIZE, then WRITE_SIZE -- separate passes,
    the two do not fit the TCC's counter slots together) of THIS command on a warm-up step and two timed steps, no trace
    options.  Corrections of MI355X_MICROARCH.md (HBM section): both counters are in KB; on gfx950 FETCH_SIZE reports half
    of the bytes read, so it is doubled; WRITE_SIZE is taken as is.  Returns (bytes per launch or None, note, issue figures or None).
    """
    exe = None
    for c in ("/opt/rocm/bin/rocprofv3", "rocprofv3"):
        if os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c):
            exe = c
            break
    got = {}
    try:
        with tempfile.TemporaryDirectory(prefix="theta_pmc_", dir="/tmp") as td:
            env = dict(os.environ, TMPDIR="/tmp")
            # (third pass: what the kernel is bound by -- vector-ALU issue -- in the counters' own terms)
            for grp in (("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES")):
                out = os.path.join(td, grp[0])
                cmd = [exe, "--pmc"] + list(grp) + ["--output-format", "csv", "-d", out, "-o", "pmc", "--", sys.executable,
                       os.path.abspath(__file__), "--gpus", "1", "--steps", "3", "--warmup", "2", "--batch", str(args.batch),
                       "--leg", args.leg, "--no-cpu-baseline", "--no-legs", "--no-traffic", "--no-extras"]
                try:
                    subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240, check=True)
                except Exception:
                    if grp[0].startswith("SQ_"):
                        continue                   # (the issue figures are an extra: the traffic stands without them)
                    raise
                rows = {c: [] for c in grp}
                for dp, _d, fs in os.walk(out):
                    for f in fs:
                        if f.endswith("counter_collection.csv"):
                            import csv
                            with open(os.path.join(dp, f)) as fh:
                                for row in csv.DictReader(fh):
                                    if DOMINANT_KERNEL in row.get("Kernel_Name", "") and row.get("Counter_Name") in rows:
                                        rows[row["Counter_Name"]].append((int(row["Dispatch_Id"]), float(row.get("Grid_Size") or 0), float(row["Counter_Value"])))
                for ctr in grp:
                    if not rows[ctr]:
                        if ctr.startswith("SQ_"):
                            continue
                        return None, "rocprofv3 produced no %s rows for the search kernel" % ctr, None
                    # per launch, summed over the counter's instances; the BULK launches are those with (about) the largest grid -- a step
                    # is a short first slice + the bulk --, and the timed steps are the last ones: the job's first step runs against a
                    # poor minimum and writes millions of contender records (round 3 took the median of the launches with the LARGEST
                    # VALUES, i.e. of exactly those: its "1.3 GB per launch" was the first step's contender list, profiles/r4/NOTES.md)
                    per = {}
                    for did, grid, val in rows[ctr]:
                        g, v = per.get(did, (0.0, 0.0))
                        per[did] = (max(g, grid), v + val)
                    gmax = max(g for g, _v in per.values())
                    bulk = [per[d][1] for d in sorted(per) if per[d][0] >= 0.5 * gmax][-3:]
                    bulk.sort()
                    got[ctr] = bulk[len(bulk) // 2]
                    if ctr == "SQ_INSTS_VALU":
                        # ... and for a figure per candidate, one step's worth: its bulk launch + its first slice (the largest of the
                        # other launches; the sample launches are a millionth) over args.batch candidates.  (Not the pass's total: with
                        # two warm-up steps the pass's minimum is still poor and one of its steps overflows its list and is redone.)
                        small = sorted(per[d][1] for d in per if per[d][0] < 0.5 * gmax)[-3:]
                        got["_valu_step"] = got[ctr] + (small[len(small) // 2] if small else 0.0)
    except Exception as ex:
        return None, "rocprofv3 --pmc pass failed: %s" % (str(ex)[:200],), None
    byt = (2.0 * got["FETCH_SIZE"] + got["WRITE_SIZE"]) * 1024.0
    issue = None
    if all(k in got for k in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES")) and got["SQ_WAVE_CYCLES"] > 0:
        wps = 2 if "double" in LEGS[args.leg][2] else 3               # resident waves per SIMD of the instantiation (VGPR / LDS bound)
        apw = got["SQ_ACTIVE_INST_VALU"] / got["SQ_WAVE_CYCLES"]
        issue = {"valu_wave_instructions_per_launch": got["SQ_INSTS_VALU"], "valu_active_per_wave_cycle": apw, "waves_per_simd": wps,
                 "valu_port_busy": min(1.0, apw * wps),
                 "valu_wave_instructions_per_candidate": (got["_valu_step"] / float(args.batch)) if "_valu_step" in got else None,
                 "note": "median bulk launch of the timed steps; SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES is one wave's share of its SIMD's "
                         "vector issue slots, x resident waves per SIMD = how busy the port the kernel is bound by is"}
    return byt, "(2 x FETCH_SIZE + WRITE_SIZE) KB, median over the bulk launches of the three TIMED steps of the dominant kernel in rocprofv3 --pmc passes of this command (2 warm-up + 3 timed steps)", issue


def extras(ctx, cpu_seconds):
    """
    wall_clock_to_best -- the second half of BASELINE's metric -- end to end through do_optimization_single (search + finalists
    in reference arithmetic + tie replay) on the exhaustible BASELINE configs: config 1 (example/Example.intervals -n 2 -k 3
    after interval selection: 142 560 candidates; fixture tests/golden/example_n2.json, which also holds the reference's own
    search time in the build container), config 2 (synthetic m=25, n=2, k=5: 142 506 candidates; CPU side = the oracle on a
    sample, extrapolated) and the n=3 stage of the two-stage command on syn14.intervals.
    """
    from theta_amd.search import do_optimization_single
    w = {}
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import theta_oracle as orc
    try:
        e = json.load(open(os.path.join(ROOT, "tests", "golden", "example_n2.json")))
        ts = []
        for _ in range(3):
            t = time.time()
            b1 = do_optimization_single(2, e["m"], e["k"], e["tau"], list(e["lb"]), list(e["ub"]), e["r"], e["rN"],
                                        e["max_normal"], e["sorted_index"], False, False)
            ts.append(time.time() - t)
        w["config1_example_n2_k3"] = {"candidates": 142560, "gpu_wall_s": min(ts), "nll": b1[0][2],
                                      "reference_search_s": e.get("ref_search_seconds"),
                                      "reference_note": "the reference's own search loop on this input, timed in the build container"}
    except Exception as ex:       # the fixture is test data; the bench line does not depend on it
        w["config1_example_n2_k3"] = {"error": str(ex)}
    try:
        # an exhaustible n=3 search with a reference time: the n=3 stage of `RunTHetA syn14.intervals -k 3 --FORCE`
        # (tests/golden/cli/syn14d.n3.withBounds: 14 intervals, 1 369 938 candidate matrices; the reference's own CLI, one
        # process, needed about 55 minutes for this stage in the build container -- tests/golden/make_golden_cli.py)
        from theta_amd.DataTools import sort_r, set_total_read_counts
        rows = [l.split() for l in open(os.path.join(ROOT, "tests", "golden", "cli", "syn14d.n3.withBounds")) if not l.startswith("#")]
        tum, nrm = [int(x[4]) for x in rows], [int(x[5]) for x in rows]
        ub3, lb3 = [int(x[6]) for x in rows], [int(x[7]) for x in rows]
        set_total_read_counts(sum(tum), sum(nrm))
        rs3, rNs3, order3 = sort_r(nrm, tum)
        lbs3, ubs3 = [lb3[i] for i in order3], [ub3[i] for i in order3]
        ts = []
        for _ in range(3):
            t = time.time()
            b3 = do_optimization_single(3, len(rows), 3, TAU, list(lbs3), list(ubs3), rs3, rNs3, 1.0, order3, False, False)
            ts.append(time.time() - t)
        w["syn14_n3_stage"] = {"candidates": 1369938, "gpu_wall_s": min(ts), "nll": b3[0][2], "reference_search_s_approx": 3300.0,
                               "reference_note": "the reference CLI's n=3 stage on this input, one process, build container"}
    except Exception as ex:
        w["syn14_n3_stage"] = {"error": str(ex)}
    # BASELINE configs 3 and 4 -- synthetic m = 50 intervals, n = 3, k = 4 / 6, FULL bounds: 4e27 / 2.6e38 matrices, which neither the
    # reference's loop nor any rank-by-rank kernel finishes -- searched WHOLE by branch and bound over the mixture space
    # (theta_mix_search behind do_optimization_single, csrc/bnb.hip; tests/test_gpu_bnb.py: identical to the exhaustive search on
    # whole spaces of 1e6-1e9 matrices and to reference-written lists).  config 4 is this bench's own instance.
    from theta_amd import search as _S
    # ... and config 5's shape (m = 200, k = 7, full bounds: a space beyond 2^128) the same way.
    for key, mm, kk, sd in (("config3_m50_n3_k4", 50, 4, 7), ("config4_m50_n3_k6", 50, K_MAX, SEED), ("config5_m200_n3_k7", 200, 7, 55)):
        try:
            r3, rN3, order3 = synth(seed=sd, m=mm, n=3, k=kk)
            ts = []
            for _ in range(3):
                t = time.time()
                b3 = do_optimization_single(3, mm, kk, TAU, [0] * mm, [kk] * mm, r3, rN3, 1.0, order3, False, False)
                ts.append(time.time() - t)
            rp = _S.last_report
            mx = rp.mix or {}
            w[key] = {"candidates": float(rp.candidates), "gpu_wall_s": min(ts), "nll": b3[0][2], "entries": len(b3),
                      "mu": [float(x) for x in b3[0][1]],
                      "method": "branch and bound over the mixture space (theta_mix_search) + the reference's procedure on the listed matrices",
                      "boxes_tested": mx.get("boxes_tested"), "leaves": mx.get("leaves"), "matrices_listed": mx.get("listed"),
                      "octree_kernel_ms": mx.get("kernel_ms"), "smallest_leaf_bound": mx.get("min_bound"),
                      "dive": mx.get("dive"), "incumbent_heuristic": mx.get("heuristic_nll"), "heuristic_s": mx.get("heuristic_seconds"),
                      "ladder_passes": len(mx.get("passes") or []), "host_syncs_of_the_last_walk": mx.get("syncs"),
                      "lines": mx.get("lines"), "line_leaves": mx.get("line_leaves"), "rank_deficient_records": mx.get("rank_deficient_records"),
                      "rank_deficient_bound": mx.get("rank_deficient_bound"), "threshold": mx.get("threshold"),
                      "count_saturated": bool(rp.candidates >= 2 ** 128 - 1),
                      "reference_estimate_s": float(rp.candidates) / 30.0,
                      "reference_note": "the reference visits every matrix at ~30 per second and process (BASELINE.md): no run of it can finish",
                      "not_included": "NaN outcomes (some rank-deficient matrices, one full-rank matrix in a million): no bound reaches them. "
                                      "Every FINITE outcome of a rank-deficient matrix within the threshold is among the records "
                                      "(round 6: one tree per line of the alphabet's grid), and none lies below rank_deficient_bound"}
        except Exception as ex:
            w[key] = {"error": str(ex)[:200]}
    r2, rN2, order2 = synth(seed=11, m=25, n=2, k=5)
    ts = []
    for _ in range(3):
        t = time.time()
        b2 = do_optimization_single(2, 25, 5, TAU, [0] * 25, [5] * 25, r2, rN2, 1.0, order2, False, False)
        ts.append(time.time() - t)
    t = time.time()
    n_s = 0
    for c in orc.enumerate_n2(25, TAU, [0] * 25, [5] * 25):
        orc.solve_n2(orc.col_to_matrix_n2(c, TAU), r2, rN2, 1.0)
        n_s += 1
        if time.time() - t > cpu_seconds / 3:
            break
    cpu_rate = n_s / (time.time() - t)
    w["config2_m25_n2_k5"] = {"candidates": 142506, "gpu_wall_s": min(ts), "nll": b2[0][2],
                              "cpu_oracle_candidates_per_s": cpu_rate, "cpu_oracle_estimated_s": 142506 / cpu_rate,
                              "cpu_sample": "%d candidates, 1 process" % n_s}
    return {"wall_clock_to_best": w}


def config5_rider(ctx, problem_unused=None):
    """
    BASELINE config 5 (m=200, n=3, k=7, interval-subset resampling) as a rider with its own roofline: theta_score_masked on
    B byte candidates x S row masks (CalcAllC.L3's valid_rows contract, CalcAllC.py:70-75); kernel time by HIP events.
    For S >= 16 the masked sums run as an FP64 GEMM on the matrix cores, so the bound is the FP64 MFMA peak, and the HBM
    figure (algorithmic bytes of SURVEY 8(d): m(n-1) + 8n per candidate, m/8 per mask, 8 per pair) is reported beside it.
    """
    m5, k5, B, S = 200, 7, 131072, 512
    rng = np.random.RandomState(55)
    Cb = rng.randint(0, k5 + 1, (B, m5, 2)).astype(np.uint8)
    w = rng.poisson(100000, m5).astype(np.float64) + 1.0
    r = rng.poisson(120000, m5).astype(np.float64)
    mu = rng.dirichlet(np.ones(3) * 4, B)
    words = (m5 + 63) // 64
    bits = rng.rand(S, m5) < 0.8
    masks = np.zeros((S, words), np.uint64)
    for i in range(m5):
        masks[:, i // 64] |= (bits[:, i].astype(np.uint64) << np.uint64(i % 64))
    best = None
    for _ in range(3):
        _nll, ms = ctx.score_masked(3, TAU, Cb, w, r, mu, masks)
        best = ms if best is None else min(best, ms)
    pairs = B * S
    flop = 2.0 * 2.0 * m5 * pairs                 # two masked sums (C.mu and r ln C.mu) of m terms per pair, as FMAs
    byt = B * (m5 * 2 + 24) + S * words * 8 + pairs * 8
    return {"config": "m=200, n=3, k=7: %d byte candidates x %d interval masks (theta_score_masked)" % (B, S),
            "value": pairs / (best * 1e-3), "unit": "(candidate, mask) pairs/s", "kernel_ms": best, "dtype": "f64",
            "roofline": {"bound": "mfma", "achieved": flop / (best * 1e-3) / 1e12, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": flop / (best * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                         "hbm_algorithmic_GBps": byt / (best * 1e-3) / 1e9, "hbm_frac": byt / (best * 1e-3) / 1e9 / HBM_PEAK_GBS}}


def config5_search_rider(ctx):
    """BASELINE config 5's literal shape as a SEARCH: synthetic m=200 intervals, n=3, k=7, full bounds [0, 7] -- 1e150 matrices, the
    counting table saturates at 2^128 - 1 -- and a 2^30-candidate rank range at rank 2^100 of the reference's order searched as
    shipped (sieve + finish kernels, four prefix intervals per lane)."""
    import theta_amd
    r, rN, order = synth(seed=55, m=200, n=3, k=7)
    p = theta_amd.Problem(ctx, 3, 200, TAU, r, rN, [0] * 200, [7] * 200, 1.0)
    b, span = 1 << 100, 1 << 30
    res = p.search(b, b + (1 << 22), window=0.5)                   # (a minimum to start from, like the pieces of a job)
    best = None
    for _ in range(2):
        if len(res["nll"]):
            p.hint(float(res["nll"].min()))
        t0 = time.time()
        res = p.search(b, b + span, window=0.5)
        dt = time.time() - t0
        st = res["stats"]
        if best is None or st["kernel_ms"] < best[0]:
            best = (st["kernel_ms"], dt, st)
    kms, dt, st = best
    out = {"config": "m=200, n=3, k=7, full bounds: ranks [2^100, 2^100 + 2^30) of a space of >= 2^128 matrices (count saturated)",
           "value": span / dt, "unit": "candidates/s (searched)", "kernel_ms": kms, "wall_ms": 1e3 * dt,
           "kernel_candidates_per_s": span / (kms * 1e-3), "dismissed_fraction": st["dismissed"] / span,
           "finalists": len(res["rank"]), "count_saturated": p.count == 2 ** 128 - 1, "dtype": "f32+f64"}
    p.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1 << 31, help="candidates per step per GPU (one theta_search call)")
    ap.add_argument("--leg", default="full_solve_f64_tight_certified", choices=sorted(LEGS), help="what the timed region runs (the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the other legs (they run at N=1 only anyway)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 --pmc passes that measure HBM traffic")
    ap.add_argument("--no-extras", action="store_true", help="skip wall_clock_to_best and the config-5 rider")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)

    import theta_amd
    from theta_amd.search import COLLECT_WINDOW
    global M, K_MAX
    ndev = int(os.environ.get("THETA_BENCH_NDEV", "0")) or None      # (lets a 2-rank smoke test share one GPU)
    standin = os.environ.get("THETA_BENCH_INIT")                     # tests only: "module:function" -> the context of a CPU stand-in
    if standin:                                                      # device (tests/standin_device.py), so that the N-rank plumbing of
        import importlib                                             # this file runs on machines without a GPU; with
        mod, fn = standin.split(":")                                 # THETA_BENCH_SHAPE = "m,k" for a space the oracle can walk
        ctx = getattr(importlib.import_module(mod), fn)(rank)
        if os.environ.get("THETA_BENCH_SHAPE"):
            M, K_MAX = (int(v) for v in os.environ["THETA_BENCH_SHAPE"].split(","))
    else:
        ctx = theta_amd.Context(local % ndev if ndev else local)
    real_device = hasattr(ctx, "_h")
    comm = None
    if world > 1:
        # THETA_BENCH_TRANSPORT: unset / "rccl" = RCCL over xGMI (the library dlopens librccl.so) and NOTHING ELSE -- a communicator
        # RCCL refuses ends the run with a non-zero exit, so that a scaling line can never be green on a downgraded transport
        # (round-4 verdict); "host" = the library's TCP star (smoke tests on a one-GPU box, CPU tests); "rccl_or_host" = try RCCL,
        # fall back to the host transport and say so on the line (the two collectives of the job are a few hundred bytes).
        want = os.environ.get("THETA_BENCH_TRANSPORT", "rccl")
        if want not in ("rccl", "host", "rccl_or_host"):
            raise SystemExit("THETA_BENCH_TRANSPORT must be rccl, host or rccl_or_host")
        try:
            comm = theta_amd.Comm(ctx if real_device else None, rank=rank, world=world, transport="host" if want == "host" else "rccl")
        except theta_amd.ThetaError as ex:
            if want != "rccl_or_host":
                print("rank %d: the %s transport could not be set up (%s); not falling back (THETA_BENCH_TRANSPORT=rccl_or_host would)"
                      % (rank, want, ex), file=sys.stderr)
                sys.exit(3)
            # (a failure on SOME ranks only ends in the rendezvous time-out of this second attempt)
            print("rank %d: RCCL transport failed (%s); falling back to the host transport" % (rank, ex), file=sys.stderr)
            port = (int(os.environ.get("THETA_COMM_PORT", 0)) or int(os.environ.get("MASTER_PORT", "29400")) + 1) + 1
            comm = theta_amd.Comm(ctx, rank=rank, world=world, port=port, transport="host")
    r, rN, order = synth(m=M, k=K_MAX)
    lb, ub = [0] * M, [K_MAX] * M
    # (theta_amd._lib.Problem: the class a stand-in's init hook has replaced, else theta_amd.Problem itself)
    problem = theta_amd._lib.Problem(ctx, N_POP, M, TAU, r, rN, lb, ub, 1.0)   # builds the 173 MB counting table in HBM
    total = problem.count
    shard0, shard1 = total * rank // world, total * (rank + 1) // world
    nsteps = args.warmup + args.steps
    stride = (shard1 - shard0 - args.batch) // max(nsteps, 1)
    # (mid-points: the very first ranks of the space are a stretch of near-ties)
    begins = [shard0 + i * stride + stride // 2 for i in range(nsteps)]

    def barrier():
        if comm is not None:
            comm.barrier()
        if real_device:
            ctx.synchronize()

    def set_opts(opts, on):
        for k, v in opts.items():
            if v == "certified":
                v = certified_conv_l2(r)
            problem.set_option(k, v if on else (1e-4 if k == "n3_conv_l2" else 0))      # (1e-4: the library's default coarse tolerance)

    head_opts, head_dtype, head_kernel = LEGS[args.leg]
    set_opts(head_opts, True)
    leg = Leg(problem, begins, args.batch, COLLECT_WINDOW)
    for i in range(args.warmup):
        leg.step(i, count=False)
        if i == 0 and comm is not None:
            # the one collective BEFORE a sharded search (search.do_optimization_distributed): the shards' probe minima are
            # all-reduced (min) and every shard starts from the job's minimum, not its own -- a shard whose candidates are
            # poor neither floods its lists nor lists contenders nobody needs
            leg.running = float(comm.allreduce_min(leg.running)[0])
    if args.warmup == 0 and comm is not None:
        leg.running = float(comm.allreduce_min(leg.running)[0])
    barrier()
    t0 = time.time()
    for i in range(args.warmup, nsteps):
        leg.step(i)
    if comm is not None:
        # the single exchange of the sharded search: shard minima + finalists (a few hundred bytes), over RCCL
        recs = []
        if leg.best is not None:
            for j in range(len(leg.best["rank"])):
                recs.append({"rank": leg.best["rank"][j], "c": leg.best["C"][j], "mu": leg.best["mu"][j], "nll": float(leg.best["nll"][j]),
                             "vals": np.zeros(M)})
        merged, gmin = comm.exchange_finalists(N_POP, M, recs, COLLECT_WINDOW)
    barrier()
    dt = time.time() - t0
    set_opts(head_opts, False)

    # (the device a rank runs on comes from LOCAL_RANK -- on a stand-in context, which has none, the one it WOULD select is reported;
    # the ranks' searched ranges as fractions of the space: double precision separates 1e-16 of it, the shards are 1/8 each)
    dev_of_rank = float(getattr(ctx, "device", (local % ndev if ndev else local)))
    lo_frac, hi_frac = float(begins[0]) / float(total), float(begins[-1] + args.batch) / float(total)
    mine = np.array([float(leg.tot["evaluated"]), dt, dev_of_rank, lo_frac, hi_frac, float(shard0) / float(total), float(shard1) / float(total)],
                    dtype=np.float64)
    allv = comm.allgather(mine) if comm is not None else mine[None, :]
    if rank == 0:
        ev_all = allv[:, 0].sum()
        t_max = allv[:, 1].max()
        devices = [int(v) for v in allv[:, 2]]
        if world > 1 and real_device and not ndev and len(set(devices)) != world:
            print("ranks share devices %s: one process per GPU expected" % devices, file=sys.stderr)
            sys.exit(4)
        value = ev_all / t_max
        legs = {args.leg: leg.summary(dt, args.leg, head_dtype, head_kernel)}
        what = {"full_solve_f64": "COARSE tolerance lambda^2 / sum r < 1e-4 -- every candidate generated, iterated in FP64 until an evaluation "
                                  "finds it (then the step) and valued; none dismissed by a bound; the tight-tolerance rate is leg "
                                  "full_solve_f64_tight",
                "full_solve_f64_tight": "TIGHT tolerance lambda^2 / sum r < 1e-12 (every candidate's mu within 1e-6 of its optimum) -- every "
                                        "candidate generated, iterated in FP64 and valued; none dismissed by a bound",
                "full_solve_f64_tight_certified": "north_star's tolerance by certificate: every candidate left at a point whose lambda^2 / sum r is "
                                                  "below 1e-12 by the self-concordance bound of its last Newton step AND whose mu is bounded "
                                                  "within 1e-6 of its optimum (per candidate: smaller Hessian eigenvalue, Jacobian of nu -> mu; "
                                                  "option n3_mu_tol) -- every candidate generated, iterated in FP64 and valued; none dismissed by a bound",
                "full_solve_f64_l2_certified": "round 5's headline: the decrement certificate alone (lambda^2 / sum r below 1e-12 at the point a "
                                               "candidate is left at; mu to 8e-7 on this instance, not by construction)",
                "full_solve_f32": "COARSE tolerance lambda^2 / sum r < 1e-4 -- every candidate generated, iterated in packed FP32 and valued; "
                                  "none dismissed by a bound",
                "search": "candidates SEARCHED by the shipped branch-and-bound (a whole prefix finished by the bound of its relaxed problem, else "
                          "bound-pruned after one shared packed-FP32 evaluation)"}[args.leg]
        out = {
            "metric": "candidate C-matrices evaluated/sec (whole node); " + what,
            "value": value, "unit": "candidates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * t_max / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": head_dtype, "data": "synthetic",
            "config": {"workload": "synthetic m=50 intervals, n=3, k=6, full bounds [0,6]: rank-range search "
                                   "(BASELINE config 4 shape)", "m": M, "n": N_POP, "k": K_MAX, "leg": args.leg,
                       "candidates_per_step_per_gpu": args.batch, "total_candidates_in_space": float(total),
                       "parallelism": "rank-range sharding x%d, hint all-reduce before + one exchange of finalists after "
                                      "(library-owned RCCL communicator)" % world},
        }
        # what carried the job's collectives and where the ranks ran, at the top level: a downgraded transport or two ranks on one
        # GPU must be visible without reading into `comm`
        info = comm.info() if comm is not None else {"transport": "none", "rccl_version": 0}
        out["transport"] = info["transport"]
        out["rccl_version"] = info["rccl_version"]
        out["rank_devices"] = devices
        out["candidates_per_rank"] = [float(v) for v in allv[:, 0]]
        # what each rank searched, as fractions of the rank space: [first rank searched, last rank searched + 1) inside its shard
        # [shard begin, shard end) -- disjoint by construction (shard g = [N g / G, N (g + 1) / G)); a reader can check it on the line
        out["rank_ranges"] = [{"searched": [float(a), float(b)], "shard": [float(c), float(d)]} for a, b, c, d in allv[:, 3:7]]
        if comm is not None:
            out["comm"] = info
        if world == 1 and not args.no_legs:
            # the other legs on the same rank ranges, chained to the job (they start from the minimum found so far)
            for name, (opts, dtype, kern) in LEGS.items():
                if name == args.leg:
                    continue
                set_opts(opts, True)
                lg = Leg(problem, begins[args.warmup:], args.batch, COLLECT_WINDOW)
                lg.running = leg.running
                lg.step(0, count=False)
                ctx.synchronize()
                t1 = time.time()
                for i in range(len(lg.begins)):                # (every stretch of the timed region: the legs are comparable with the headline and with each other)
                    lg.step(i)
                ctx.synchronize()
                legs[name] = lg.summary(time.time() - t1, name, dtype, kern)
                set_opts(opts, False)
        s = legs[args.leg]
        traffic, tnote, issue = (None, "not measured (--no-traffic or N > 1)", None)
        if world == 1 and not args.no_traffic:
            traffic, tnote, issue = measure_traffic(args)
        out["roofline"] = {
            "bound": "valu", "bound_detail": "vector-ALU (VALU issue) bound, not HBM and not MFMA: candidates are generated on chip "
            "(~0 algorithmic HBM bytes) and the per-candidate C.mu is an (18 x 3).(3) product after group aggregation -- no GEMM. "
            "Kernel time = sieve + finish kernels of a step (HIP events around both). "
            "`peak` is the FP64 vector peak (78.6 TFLOP/s) weighted with the packed-FP32 vector peak (157.3) by the executed mix",
            "kernel": head_kernel, "achieved": s["achieved"], "peak": s["peak"], "unit": "TFLOP/s", "frac": s["frac"],
            "traffic": traffic, "traffic_note": tnote, "algorithmic_bytes_per_launch": 0, "issue": issue,
            "kernel_ms_per_launch": s["kernel_ms_per_launch"], "legs": legs,
            "note": "`achieved` = FLOP executed by the likelihood arithmetic (counted in-kernel from the evaluations and terms "
                    "actually run; slices redone by the fused kernel are reported apart, redo_*) / HIP-event kernel time; "
                    "see DESIGN.md section 6"}
        out["setup_ms_per_step"] = leg.setup_ms / max(leg.launches, 1)
        if world == 1 and real_device and M >= 8 and not args.no_legs:
            # what the headline's kernel leaves a candidate at: the witness build's records for every 1024th candidate of the first 2^24
            # ranks of the last timed range (tests/test_gpu_round5.py compares such records with the oracle one by one)
            # (Tasks of 16 383 candidates for this call -- the job's are 16 384 --: a task starts from the simplex centre, its first round
            # takes 5-6 evaluations, and with tasks of a power of two every 8th / 16th sampled candidate would be a task's first: round 5's
            # witness read 2.9 evaluations per candidate for a job that took 2.35.  With an odd task size the samples fall on all offsets.)
            try:
                wit_opts = dict(head_opts, n3_per_task=16383)
                set_opts(wit_opts, True)
                problem.hint(leg.running)
                wb = begins[-1]
                rec, wst = problem.witness(wb, wb + min(args.batch, 1 << 24), every_log2=10, window=COLLECT_WINDOW)
                set_opts(wit_opts, False)
                st_names = {0: "none", 1: "converged_at_shared_evaluation", 2: "converged_in_queue", 3: "bound_at_shared_evaluation",
                            4: "bound_in_queue", 5: "contender", 6: "unsolved_to_finish_kernel"}
                reg = rec["status"] != 0
                out["witness"] = {"leg": args.leg, "records": int(len(rec)), "every": 1024,
                                  "status": {st_names[int(k)]: int(v) for k, v in zip(*np.unique(rec["status"], return_counts=True))},
                                  "evaluations_mean": float(rec["evaluations"][reg].mean()) if reg.any() else 0.0,
                                  # (... and over ALL candidates of the witnessed range, from the call's counters)
                                  "evaluations_mean_of_the_range": float(wst["iterations"]) / max(float(wst["evaluated"]), 1.0),
                                  "evaluations_max": int(rec["evaluations"].max()),
                                  "l2_last_max": float(np.nanmax(rec["l2_last"][reg])) if reg.any() else 0.0,
                                  "l2_first_median": float(np.nanmedian(rec["l2_first"][reg])) if reg.any() else 0.0,
                                  "conv_l2": certified_conv_l2(r) if head_opts.get("n3_conv_l2") == "certified" else head_opts.get("n3_conv_l2", 1e-4),
                                  "mu_tol": head_opts.get("n3_mu_tol"),
                                  "mu_bound_max": float(np.nanmax(rec["mu_bound"][reg])) if reg.any() else 0.0,
                                  "mu_bound_median": float(np.nanmedian(rec["mu_bound"][reg])) if reg.any() else 0.0,
                                  "note": "l2_last = lambda^2 / sum r found by a candidate's LAST evaluation; the candidate is left one full "
                                          "Newton step beyond it (certified below 1e-12 when l2_last <= conv_l2)"}
            except Exception as ex:
                out["witness"] = {"error": str(ex)[:200]}
        if world == 1 and not args.no_cpu_baseline:
            # bounded sample of the SAME candidates, materialised by the enumerate kernel, solved by the oracle on the host
            n_s = min(1 << 16, 4096 * host_cores()[0])       # (more than the oracle gets through in its budget: the workers stop on time)
            per = max(1, n_s // nsteps)
            cands = np.concatenate([problem.enumerate(b + 12345, per) for b in begins])
            out["cpu_baseline"] = cpu_baseline(cands, r, rN, args.cpu_seconds)
            out["speedup_vs_cpu_all_cores"] = value / out["cpu_baseline"]["value"]
        else:
            out["cpu_baseline"] = None
        if world == 1 and not args.no_extras:
            try:
                out.update(extras(ctx, args.cpu_seconds))
                out["riders"] = {"config5_masked_scorer": config5_rider(ctx), "config5_search": config5_search_rider(ctx)}
            except Exception as ex:
                out["extras_error"] = str(ex)
        # RCCL writes a version banner to the C stdout of the process that creates a communicator (buffered when piped): flush it
        # now, so that the JSON line is the LAST line this process prints
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if comm is not None:
        comm.barrier()
        comm.close()


if __name__ == "__main__":
    main()
