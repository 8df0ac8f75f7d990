"""Round 5 campaign: the mixture-space branch and bound (search.mix_records) against the linear walk on random WHOLE spaces --
full bounds and ragged ones, tau 1-3, read depths from the bench's 1e5-1e6 per interval down to a few hundred.
    python tools/bnb_campaign.py [instances] [seed0] [max candidates]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import campaign
import theta_amd
from theta_amd import search as S
from conftest import rank_deficient


def instance(seed):
    rng = np.random.RandomState(seed)
    m, K = int(rng.randint(9, 15)), int(rng.randint(3, 7))
    tau = int(rng.choice([2, 2, 2, 1, 3]))
    depth = float(rng.choice([0.01, 0.01, 0.001, 0.0001, 0.00002]))
    L = rng.randint(2_000_000, 20_000_000, m)
    rN = np.maximum(rng.poisson(L * depth), 3)
    C = np.full((m, 3), float(tau))
    C[:, 1:] = rng.randint(0, K + 1, (m, 2))
    mu = rng.dirichlet(np.ones(3) * 4)
    p = (C * rN[:, None]) @ mu
    r = np.maximum(rng.multinomial(int(rN.sum() * 1.2), p / p.sum()), 1)
    ratio = (r / rN) * (rN.sum() / r.sum())
    order = np.argsort(ratio, kind="stable")
    r, rN = [int(x) for x in r[order]], [int(x) for x in rN[order]]
    if rng.rand() < 0.4:
        lb = sorted(int(x) for x in rng.randint(0, 2, m))
        ub = sorted(int(x) for x in rng.randint(max(2, K - 2), K + 1, m))
    else:
        lb, ub = [0] * m, [K] * m
    return dict(m=m, K=K, tau=tau, r=r, rN=rN, lb=lb, ub=ub, depth=depth)


def full_rank(recs):
    if not recs:
        return []
    keep = ~rank_deficient(np.array([t["c"] for t in recs]))
    return [t for t, k in zip(recs, keep) if k]


def plain(best):
    return [(np.asarray(t["c"]).tolist(), [float(x) for x in t["mu"]], float(t["nll"])) for t in best]


def main():
    # --never-give-up (round 6): every space is treated like one no walk finishes -- the mixture-space search goes on however flat the
    # likelihood (a clock of 60 s per walk protects the box) -- so that the instances earlier rounds gave up on are compared as well
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if "--never-give-up" in sys.argv:
        S.MIX_WALKABLE = 0
        S.MIX_MAX_MS_LARGE = 60000.0
    want = int(args[0]) if len(args) > 0 else 40
    seed = int(args[1]) if len(args) > 1 else 70000
    cap = float(args[2]) if len(args) > 2 else 3e8
    only = None
    for a in sys.argv[1:]:
        if a.startswith("--seeds="):                      # exactly these instances (e.g. the ones an earlier campaign gave up on: tools/bnb_gave_up_seeds.txt)
            v = a.split("=", 1)[1]
            only = [int(x) for x in (open(v).read() if os.path.exists(v) else v).replace("\n", "").split(",") if x]
            want = len(only)
    ctx = theta_amd.default_context()
    done = same = gave_up = differ = 0
    tot = 0.0
    t_mix = t_ex = 0.0
    while done < want:
        seed = only[done] if only is not None else seed + 1
        inst = instance(seed)
        p = theta_amd.Problem(ctx, 3, inst["m"], inst["tau"], inst["r"], inst["rN"], inst["lb"], inst["ub"], 1.0)
        if only is None and not (1e5 <= p.count <= cap):
            p.close()
            continue
        done += 1
        tot += p.count
        t = time.time()
        try:
            recs, _ = S.mix_records(p, ctx, inst["r"], inst["rN"], 1.0, (list(inst["lb"]), list(inst["ub"])))
        except theta_amd.ThetaError as e:
            gave_up += 1
            print("seed %d m=%d K=%d tau=%d depth %g: %.3g matrices: GAVE UP (%s)" % (seed, inst["m"], inst["K"], inst["tau"], inst["depth"], p.count, str(e)[:90]), flush=True)
            p.close()
            continue
        t_mix += time.time() - t
        if only is not None:
            print("seed %d m=%d K=%d tau=%d depth %g: %.3g matrices: searched in %.2f s, %d records" % (seed, inst["m"], inst["K"], inst["tau"], inst["depth"], p.count, time.time() - t, len(recs)), flush=True)
        t = time.time()
        p.set_option("n3_nan_sweep", 0)
        ex, _st = S.collect_finalists(p, ctx, inst["r"], inst["rN"], 1.0, 0, p.count)
        ex = ex + S.fallback_records(p, ctx, inst["r"], inst["rN"], 1.0, ex)
        # round 6: the COMPLETE lists -- the rank-deficient matrices, valued by the reference's procedure, on both sides (NaN outcomes
        # apart: no bound reaches those)
        listed = set(p.last_degenerate[0])
        ex = [t for t in ex if t["rank"] not in listed] + S.degenerate_records(p, ctx, inst["r"], inst["rN"], 1.0, recs=ex)
        t_ex += time.time() - t
        p.close()
        fin = lambda rc: [t for t in rc if t["nll"] == t["nll"]]
        why = campaign.compare_best(plain(S.replay_records(fin(recs), False)), plain(S.replay_records(fin(ex), False)), tol=1e-9)
        if why:
            differ += 1
            print("seed %d m=%d K=%d tau=%d depth %g: %.3g matrices: DIFFERS: %s" % (seed, inst["m"], inst["K"], inst["tau"], inst["depth"], p.count, why), flush=True)
        else:
            same += 1
    print("instances %d (%.3g matrices): identical %d, differ %d, gave up (flat likelihood) %d; mixture-space search %.1f s, linear walk %.1f s" %
          (done, tot, same, differ, gave_up, t_mix, t_ex))
    return 1 if differ else 0


if __name__ == "__main__":
    sys.exit(main())
