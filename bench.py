#!/usr/bin/env python3
"""
bench.py -- candidate C-matrices evaluated per second (BASELINE.json metric) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (config.workload): BASELINE config 4's shape -- synthetic m=50 intervals, n=3, k=6, full
bounds [0,6] (2.6e38 admissible matrices, inexhaustible) -- searched as rank ranges of the
reference's enumeration order.  One "step" = one theta_search call over `--batch` consecutive
candidates (enumerate + solve + NLL + arg-min, fused in one kernel).  Each GPU owns the contiguous
shard [N*g/G, N*(g+1)/G) of the rank space and its steps are spread evenly through that shard, so
the sampled candidates are representative of the whole space.  Weak scaling: per-GPU work is fixed.
With N > 1 the per-shard finalists are merged with ONE small RCCL exchange (all-reduce min +
all-gather) inside the timed region.

Prints ONE JSON line (rank 0).  `value` = candidates evaluated by all GPUs / max-over-ranks wall time.
`roofline`: the fused search kernel is FP64-VALU bound (no HBM traffic per candidate by design), so
the achieved figure is executed FP64 operations (counted inside the kernel) over the kernel time
measured with HIP events on its stream, against the MI355X FP64 vector peak.
`cpu_baseline`: the CPU oracle (oracle/theta_oracle.py, a port of the reference's Python) timed on
this host's cores on a bounded sample of the same candidates.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np

FP64_VECTOR_PEAK_TFLOPS = 78.6     # MI355X FP64 vector peak (AMD spec; = 256 CU x 4 SIMD x 16 FMA lanes x 2 x 2.4 GHz)
FP32_VECTOR_PEAK_TFLOPS = 157.3    # MI355X FP32 vector peak (packed v_pk_fma_f32: two FP32 FMAs per FP64 lane)
HBM_PEAK_GBS = 8000.0

M, N_POP, K_MAX, TAU, SEED = 50, 3, 6, 2, 4242


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


def cpu_baseline(cands, r, rN, budget_s=15.0):
    """Oracle (port of the reference's Optimizer.solve) on the host cores, bounded by `budget_s` seconds."""
    cores = os.cpu_count() or 1
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import theta_oracle as orc
    # single process first (the reference's default NUM_PROCESSES=1)
    t0 = time.time()
    n1 = 0
    for c in cands[: max(8, len(cands) // (4 * cores))]:
        orc.solve_n3(orc.rows_to_matrix_n3([tuple(x) for x in c], TAU), r, rN)
        n1 += 1
        if time.time() - t0 > budget_s / 3:
            break
    per_process = n1 / (time.time() - t0)
    # all cores (the reference's multiprocessing path: NUM_PROCESSES = cores)
    chunks = [cands[i::cores] for i in range(cores)]
    ctx = mp.get_context("fork")
    with ctx.Pool(cores, initializer=_cpu_init, initargs=(r, rN)) as pool:
        res = pool.map(_cpu_work, [(ch, budget_s * 2 / 3) for ch in chunks], chunksize=1)
    done = sum(x[0] for x in res)
    dt = max(x[1] for x in res)          # workers run concurrently; process start-up is not charged to the CPU
    return {"value": done / dt, "unit": "candidates/s", "cores": cores, "kind": "port",
            "per_process": per_process,
            "sample": "%d candidates drawn from the GPU run's own rank ranges, oracle.solve_n3 (scipy fsolve/BFGS), "
                      "%d concurrent processes, %.1f s of solving each" % (done, cores, dt)}


def extras(ctx, r, rN, cpu_seconds):
    """
    Secondary figures of the BASELINE metric, N=1 only, outside the timed region:
    * full_solve: the same search with the lower-bound dismissal switched off (THETA_N3_NO_DISMISS=1) -- every candidate is
      iterated to the coarse tolerance and valued, none is finished after one evaluation by its bound;
    * wall_clock_to_best: end-to-end do_optimization_single (search + finalists in reference arithmetic + tie replay) on the
      two exhaustible BASELINE configs -- config 1 (example/Example.intervals -n 2 -k 3 after interval selection: 142 560
      candidates; fixture tests/golden/example_n2.json, which also holds the reference's own search time in the build
      container) and config 2 (synthetic m=25, n=2, k=5: 142 506 candidates; the CPU side is the oracle on a sample of the
      same candidates, extrapolated).
    """
    import theta_amd
    from theta_amd.search import do_optimization_single
    out = {}
    os.environ["THETA_N3_NO_DISMISS"] = "1"
    try:
        p2 = theta_amd.Problem(ctx, N_POP, M, TAU, r, rN, [0] * M, [K_MAX] * M, 1.0)
    finally:
        del os.environ["THETA_N3_NO_DISMISS"]
    total, batch = p2.count, 1 << 29
    ev = 0
    kms = 0.0
    best = float("inf")
    t0 = None
    for i in range(3):
        if i == 1:
            t0 = time.time()
        if best < float("inf"):
            p2.hint(best)
        res = p2.search(total // 7 * (i + 1), total // 7 * (i + 1) + batch, window=0.5)
        if len(res["nll"]):
            best = min(best, float(res["nll"].min()))
        if i >= 1:
            ev += res["stats"]["evaluated"]
            kms += res["stats"]["kernel_ms"]
            dis = res["stats"]["dismissed"]
    dt = time.time() - t0
    out["full_solve"] = {"value": ev / dt, "unit": "candidates/s", "kernel_candidates_per_s": ev / (kms * 1e-3),
                         "dismissed": int(dis), "note": "THETA_N3_NO_DISMISS=1, 2 x 2^29 candidates of the same instance: no "
                         "candidate is finished by its lower bound"}
    p2.close()
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
    out["wall_clock_to_best"] = w
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1 << 31, help="candidates per step per GPU (one theta_search call)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    import torch
    local = local % max(torch.cuda.device_count(), 1)      # (lets a 2-rank smoke test share one GPU)
    torch.cuda.set_device(local)
    backend = os.environ.get("THETA_BENCH_BACKEND", "nccl")  # "nccl" == RCCL over xGMI; "gloo" only for smoke tests
    comm_dev = torch.device("cuda", local) if backend == "nccl" else torch.device("cpu")
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    if args.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)

    import theta_amd
    from theta_amd.search import collect_finalists, exchange_finalists, COLLECT_WINDOW
    ctx = theta_amd.Context(local)
    r, rN, order = synth()
    lb, ub = [0] * M, [K_MAX] * M
    problem = theta_amd.Problem(ctx, N_POP, M, TAU, r, rN, lb, ub, 1.0)   # builds the 173 MB counting table in HBM
    total = problem.count
    shard0, shard1 = total * rank // world, total * (rank + 1) // world
    nsteps = args.warmup + args.steps
    stride = (shard1 - shard0 - args.batch) // max(nsteps, 1)

    running = [float("inf")]

    def step(i):
        # one chunk of the rank-range job; like Problem.search's own chunking, a chunk starts from the minimum the job has
        # found so far (theta_problem_hint) -- that is what keeps tie / suspect lists short in ranges whose own minimum is poor
        b = shard0 + i * stride + stride // 2       # (mid-points: the very first ranks of the space are a stretch of near-ties)
        # (the first warm-up step carries a trivial hint: without one Problem.search would first probe 16 short sub-ranges --
        # 16 extra launches of the same kernel, which would blur the per-launch averages of the rocprofv3 runs of this command)
        problem.hint(running[0] if running[0] < float("inf") else 1e300)
        res = problem.search(b, b + args.batch, window=COLLECT_WINDOW)
        if len(res["nll"]):
            running[0] = min(running[0], float(res["nll"].min()))
        return res

    for i in range(args.warmup):
        step(i)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.time()
    evaluated = flops = flops32 = terms = iters = accepted = dismissed = 0
    kernel_ms = setup_ms = 0.0
    best = None
    for i in range(args.warmup, nsteps):
        res = step(i)
        st = res["stats"]
        evaluated += st["evaluated"]
        accepted += st["accepted"]
        dismissed += st["dismissed"]
        flops += st["flops"]
        flops32 += st["flops_f32"]
        terms += st["terms"]
        iters += st["iterations"]
        kernel_ms += st["kernel_ms"]
        setup_ms += st["setup_ms"]
        if len(res["nll"]) and (best is None or res["nll"].min() < best["nll"].min()):
            best = res
    if dist is not None:
        # the single exchange of the sharded search: shard minima + finalists (a few hundred bytes)
        recs = []
        if best is not None:
            for j in range(len(best["rank"])):
                recs.append({"rank": best["rank"][j], "c": best["C"][j], "mu": best["mu"][j], "nll": float(best["nll"][j]),
                             "vals": np.zeros(M)})
        merged = exchange_finalists(recs, N_POP, M, comm_dev)
    barrier()
    dt = time.time() - t0

    tot = torch.tensor([float(evaluated), dt, float(flops), kernel_ms, float(terms), float(iters), float(accepted), setup_ms,
                        float(flops32), float(dismissed)],
                       dtype=torch.float64, device=comm_dev)
    if dist is not None:
        allv = [torch.zeros_like(tot) for _ in range(world)]
        dist.all_gather(allv, tot)
        allv = torch.stack(allv).cpu().numpy()
    else:
        allv = tot.cpu().numpy()[None, :]
    if rank == 0:
        ev_all = allv[:, 0].sum()
        t_max = allv[:, 1].max()
        value = ev_all / t_max
        # roofline of the dominant kernel (rank 0's device): executed FP64 ops / HIP-event kernel time
        k_ms = allv[0, 3]
        f64, f32 = allv[0, 2], allv[0, 8]
        ach = (f64 + f32) / (k_ms * 1e-3) / 1e12
        # time-weighted peak of the mix: an FP64 op costs two packed-FP32 slots
        peak = (f64 + f32) / (f64 / FP64_VECTOR_PEAK_TFLOPS + f32 / FP32_VECTOR_PEAK_TFLOPS) if f64 + f32 > 0 else FP32_VECTOR_PEAK_TFLOPS
        launches = args.steps
        # HBM bytes per launch of the search kernel from the rocprofv3 PMC passes committed under profiles/
        # (FETCH_SIZE x2 per MI355X_MICROARCH.md's gfx950 correction, + WRITE_SIZE; both in KiB) -- not re-measured here
        traffic = None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r1", "pmc_n3_search_kernel.json")))
            traffic = (2.0 * pm["FETCH_SIZE"]["mean_per_launch"] + pm["WRITE_SIZE"]["mean_per_launch"]) * 1024.0
        except Exception:
            pass
        out = {
            "metric": "candidate C-matrices evaluated/sec (whole node)",
            "value": value, "unit": "candidates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * t_max / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32+f64", "data": "synthetic",
            "config": {"workload": "synthetic m=50 intervals, n=3, k=6, full bounds [0,6]: rank-range search "
                                   "(BASELINE config 4 shape)", "m": M, "n": N_POP, "k": K_MAX,
                       "candidates_per_step_per_gpu": args.batch, "total_candidates_in_space": float(total),
                       "parallelism": "rank-range sharding x%d, one RCCL exchange of finalists" % world},
            "roofline": {"bound": "mfma", "bound_detail": "compute bound on the vector ALUs, not HBM bound by design (this kernel issues no "
                                          "MFMA: no GEMM in the path).  Executed work is the packed-FP32 coarse Newton pass + FP32 screen "
                                          "(157.3 TFLOP/s vector peak) and FP64 for contenders (78.6 TFLOP/s); `peak` is the "
                                          "time-weighted peak of that mix",
                         "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                         "frac": ach / peak, "traffic": traffic,
                         "fp64_flop_share": f64 / max(f64 + f32, 1.0),
                         "kernel": "n3_search_kernel<6,false>", "kernel_ms_per_launch": k_ms / launches,
                         "flop_per_candidate": (f64 + f32) / max(allv[0, 0], 1.0),
                         "newton_iters_per_candidate": allv[0, 5] / max(allv[0, 0], 1.0),
                         "terms_per_iteration": allv[0, 4] / max(allv[0, 5], 1.0),
                         "kernel_candidates_per_s": allv[0, 0] / (k_ms * 1e-3),
                         # SURVEY.md 8(d)'s per-unit figure (a per-interval FP64 Newton solve: ~160 FP64 op per interval, 8.0
                         # kFLOP per candidate at m=50) x the launch's candidates: what the same throughput would cost a kernel
                         # that did that work -- the distance to `achieved` is what the algorithm saves (group tile, warm
                         # start, one packed-FP32 evaluation, dismissal by a lower bound), not ALU efficiency
                         "survey_flop_per_candidate": 160.0 * M,
                         "survey_equivalent_tflops": 160.0 * M * allv[0, 0] / (k_ms * 1e-3) / 1e12,
                         "note": "fused kernel: candidates are generated on chip, HBM bytes/candidate ~ 0 by design; "
                                 "see DESIGN.md section 'Roofline'"},
            "dismissed_fraction": allv[:, 9].sum() / max(ev_all, 1.0),      # finished by the lower bound of their optimum
            "accepted_fraction_of_solved": allv[:, 6].sum() / max(ev_all - allv[:, 9].sum(), 1.0),
            "setup_ms_per_step": allv[0, 7] / launches,
        }
        if world == 1 and not args.no_cpu_baseline:
            # bounded sample of the SAME candidates, materialised by the enumerate kernel, solved by the oracle on the host
            n_s = 96 * (os.cpu_count() or 1)
            per = max(1, n_s // nsteps)
            cands = np.concatenate([problem.enumerate(shard0 + i * stride + stride // 2 + 12345, per) for i in range(nsteps)])
            out["cpu_baseline"] = cpu_baseline(cands, r, rN, args.cpu_seconds)
            out["speedup_vs_cpu_all_cores"] = value / out["cpu_baseline"]["value"]
            try:
                out.update(extras(ctx, r, rN, args.cpu_seconds))
            except Exception as ex:
                out["extras_error"] = str(ex)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
