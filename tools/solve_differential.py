"""theta_solve_batch (the reference's per-candidate n=3 procedure restated: hybrj, the BFGS decision, M3's hybrd, L3) against scipy
itself, candidate by candidate, at scale -- run ON THE GPU BOX, whose host cores run the oracle (oracle.solve_n3: the reference's
calls into scipy.optimize, the very routines the reference uses).  Random chunks of consecutive candidates from seeded instances
of every shape of tests/campaign.py; per candidate: reported or None, NaN or not, NLL to 1e-9 relative, mu to 1e-6 where the
matrix determines it (full rank).

    python tools/solve_differential.py [candidates, default 3e6] [seconds for the oracle side, default 420]
"""
import multiprocessing as mp
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import campaign


def oracle_chunk(job):
    inst, C = job
    warnings.simplefilter("ignore")
    import theta_oracle as orc
    m, tau = inst["m"], inst["tau"]
    out = []
    if inst["n"] == 2:
        for c in C:
            s = orc.solve_n2(orc.col_to_matrix_n2([int(v) for v in c], tau), inst["r"], inst["rN"], inst["mx"])
            out.append(None if s is None else ([float(s[0][0]), float(s[0][1])], float(s[1])))
        return out
    for c in C:
        M = np.zeros((m, 3))
        M[:, 0] = tau
        M[:, 1:] = c
        try:
            s = orc.solve_n3(M, inst["r"], inst["rN"])
        except Exception as e:                               # (the reference would die on it too; none expected)
            s = ("error", repr(e))
        out.append(None if s is None else (s[0].tolist() if hasattr(s[0], "tolist") else list(s[0]), float(s[1])))
    return out


def main():
    want = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_000_000
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 420.0
    n = 2 if (len(sys.argv) > 3 and sys.argv[3] == "n2") else 3          # (n2: theta_solve_batch's brenth restatement against scipy's brenth)
    import theta_amd
    from conftest import rank_deficient
    ctx = theta_amd.Context(0)
    rng = np.random.RandomState(12345)
    jobs, gpu = [], []
    total, seed = 0, 40000
    CH = 256
    while total < want:
        seed += 1
        shape = ("low", "mid", "toy")[seed % 3] if n == 3 else ("mid", "toy", "synth")[seed % 3]
        if shape == "synth":
            import bench
            mm, kk = int(rng.randint(18, 45)), int(rng.randint(3, 7))
            r_, rN_, order_ = bench.synth(seed=seed, m=mm, n=2, k=kk)
            inst = dict(seed=seed, n=2, m=mm, k=kk, tau=2, mx=float(rng.choice([1.0, 0.6])), r=r_, rN=rN_, order=order_, lb=[0] * mm, ub=[kk] * mm)
        else:
            inst = campaign.instance(seed, n, shape)
        cnt = campaign.count_candidates(inst)
        if cnt < 300:
            continue
        p = theta_amd.Problem(ctx, n, inst["m"], inst["tau"], inst["r"], inst["rN"], inst["lb"], inst["ub"], inst["mx"])
        nch = int(min(24, max(1, cnt // CH)))
        for _ in range(nch):
            b = int(rng.randint(0, max(1, cnt - CH)))
            c = int(min(CH, cnt - b))
            C = p.enumerate(b, c)
            ok, mu, nll, _ = ctx.solve_batch(n, inst["tau"], inst["r"], inst["rN"], C, inst["mx"], want_vals=False)
            jobs.append((inst, C))
            gpu.append((ok.copy(), mu.copy(), nll.copy()))
            total += c
        p.close()
    t0 = time.time()
    pool = mp.get_context("fork").Pool(max(1, (os.cpu_count() or 2) - 2))
    handles = [pool.apply_async(oracle_chunk, (j,)) for j in jobs]
    checked = bad_class = bad_nan = bad_nll = bad_mu = n_none = n_nan = n_def = unfinished = 0
    examples = []
    for (inst, C), (ok, mu, nll), h in zip(jobs, gpu, handles):
        try:
            ref = h.get(timeout=max(0.1, t0 + seconds - time.time()))
        except mp.TimeoutError:
            unfinished += len(C)
            continue
        d = rank_deficient(C) if n == 3 else np.zeros(len(C), bool)
        for k, s in enumerate(ref):
            checked += 1
            n_def += int(d[k])
            if s is None or s[0] == "error":
                n_none += 1
                if ok[k]:
                    bad_class += 1
                    examples.append(("class", inst["seed"], C[k].tolist(), s, float(nll[k])))
                continue
            if not ok[k]:
                bad_class += 1
                examples.append(("class", inst["seed"], C[k].tolist(), s, None))
                continue
            rn, gn = s[1] != s[1], nll[k] != nll[k]
            n_nan += int(rn)
            if rn != gn:
                bad_nan += 1
                examples.append(("nan", inst["seed"], C[k].tolist(), s, float(nll[k])))
                continue
            if rn:
                continue
            if abs(nll[k] - s[1]) > 1e-9 * abs(s[1]):
                bad_nll += 1
                examples.append(("nll", inst["seed"], C[k].tolist(), s, float(nll[k])))
            elif not d[k] and max(abs(a - b) for a, b in zip(mu[k], s[0])) >= 1e-6:
                bad_mu += 1
                examples.append(("mu", inst["seed"], C[k].tolist(), s, mu[k].tolist()))
    pool.terminate()
    for e in examples[:8]:
        print("EXCEPTION:", e)
    print("candidates %d (rank-deficient %d, None in the reference %d, NaN likelihood %d), unfinished %d; outcome class differs %d, NaN on one side %d, "
          "NLL beyond 1e-9 %d, mu beyond 1e-6 %d" % (checked, n_def, n_none, n_nan, unfinished, bad_class, bad_nan, bad_nll, bad_mu))
    return 1 if (bad_class or bad_nan or bad_nll or bad_mu) else 0


if __name__ == "__main__":
    sys.exit(main())
