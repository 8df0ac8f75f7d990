"""Which candidates does the reference report with a NaN likelihood -- or with a value BELOW their true minimum?  Whole seeded spaces
through theta_solve_batch (the reference's per-candidate procedure, restated) and through the fused kernel's dump (the true
minimum over the simplex, NaN where it lies outside): every candidate with a NaN outcome, or with an outcome below the dump's
minimum, must be rank-deficient (rows on one line) -- those are the ones the search hands to the procedure wherever they stand.
Run ON THE GPU BOX:  python tools/nan_hunt.py [instances per shape]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import campaign
import theta_amd
from conftest import rank_deficient

ctx = theta_amd.Context(0)
want = int(sys.argv[1]) if len(sys.argv) > 1 else 12
tot = nan_full = low_full = nan_def = n_def = 0
for shape in ("low", "mid", "toy"):
    seed, got = 20000, 0
    while got < want:
        seed += 1
        inst = campaign.instance(seed, 3, shape)
        cnt = campaign.count_candidates(inst)
        if not (2000 <= cnt <= int(os.environ.get("NAN_HUNT_MAX", 4_000_000))):
            continue
        got += 1
        p = theta_amd.Problem(ctx, 3, inst["m"], inst["tau"], inst["r"], inst["rN"], inst["lb"], inst["ub"], inst["mx"])
        assert p.count == cnt
        inst_nan = 0
        for b in range(0, cnt, 1 << 19):
            c = min(1 << 19, cnt - b)
            C = p.enumerate(b, c)
            ok, mu, nll, _ = ctx.solve_batch(3, inst["tau"], inst["r"], inst["rN"], C, inst["mx"], want_vals=False)
            own = (ok == 1)
            dump = p.values(b, c)[0] if inst["m"] <= 64 else np.full(c, np.nan)
            d = rank_deficient(C)
            isnan = own & np.isnan(nll)
            with np.errstate(invalid="ignore"):
                below = own & ~np.isnan(nll) & ~np.isnan(dump) & (nll < dump * (1 - 1e-9))
            tot += c
            n_def += int(d.sum())
            nan_def += int((isnan & d).sum())
            nan_full += int((isnan & ~d).sum())
            inst_nan += int((isnan & ~d).sum())
            low_full += int((below & ~d).sum())
            for k in np.nonzero((isnan | below) & ~d)[0][:3]:
                print("FULL-RANK exception: shape %s seed %d rank %d nll %r dump %r mu %r C %s" % (shape, seed, b + k, nll[k], dump[k], mu[k].tolist(), C[k].tolist()))
        p.close()
        if inst_nan:
            print("instance shape %s seed %d: %d matrices, %d full-rank NaN outcomes" % (shape, seed, cnt, inst_nan))
print("candidates %d, rank-deficient %d (NaN outcome %d); full-rank with NaN outcome %d, full-rank reported below their minimum %d"
      % (tot, n_def, nan_def, nan_full, low_full))
