#!/usr/bin/env python3
"""
Golden vectors for the post-search extension calc_all_c_2 / calc_all_c_3 / calc_all_c_3_multi_event
(python/CalcAllC.py:92-328), produced by calling the reference's own functions (imported as in
make_golden.py).  Inputs AND outputs are stored, so the test needs neither the reference nor a search.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden  # noqa


def main():
    R = make_golden.import_reference()
    import CalcAllC
    from make_golden import fl
    rng = np.random.RandomState(2024)
    cases = []
    for n, m_all, used, seed, with_zero in ((2, 12, [0, 2, 3, 5, 6, 8, 9, 11], 1, False), (2, 12, [0, 2, 3, 5, 6, 8, 9, 11], 1, True),
                                            (3, 11, [1, 2, 4, 6, 7, 9], 2, False), (3, 10, [0, 1, 3, 5, 8, 9], 3, False),
                                            (3, 10, [0, 1, 3, 5, 8, 9], 3, True)):
        rng = np.random.RandomState(seed)
        L = rng.randint(2_000_000, 20_000_000, m_all)
        all_normal = [int(x) for x in rng.poisson(L * 0.002)]
        Ct = np.full((m_all, n), 2.0)
        for j in range(1, n):
            Ct[:, j] = rng.randint(0, 4, m_all)
        mu_t = rng.dirichlet(np.ones(n) * 5)
        p = (Ct * np.array(all_normal)[:, None]) @ mu_t
        p /= p.sum()
        all_tumor = [int(x) for x in rng.multinomial(int(sum(all_normal) * 1.1), p)]
        # one unused interval with zero normal count (rows [2,-1,..], quirk Q10 path)
        zero_i = [i for i in range(m_all) if i not in used][0]
        if with_zero:
            all_normal[zero_i] = 0
        r = [all_tumor[i] for i in used]
        rN = [all_normal[i] for i in used]
        c = Ct[used].copy()
        mu = tuple(float(x) for x in mu_t) if n == 2 else np.array(mu_t)
        best = [(c, mu, 0.0, [0.0] * len(used))]
        outs = {}
        if n == 2:
            res = CalcAllC.calc_all_c_2([(c.copy(), mu, 0.0, [])], list(r), list(rN), list(all_tumor), list(all_normal), list(used))
            outs["calc_all_c_2"] = res
        else:
            outs["calc_all_c_3"] = CalcAllC.calc_all_c_3([(c.copy(), mu, 0.0, [])], list(r), list(rN), list(all_tumor), list(all_normal), list(used))
            outs["calc_all_c_3_multi_event"] = CalcAllC.calc_all_c_3_multi_event([(c.copy(), mu, 0.0, [])], list(r), list(rN), list(all_tumor), list(all_normal), list(used))
        ent = {"n": n, "used": used, "all_tumor": all_tumor, "all_normal": all_normal, "r": r, "rN": rN,
               "c": c.tolist(), "mu": [float(x) for x in mu], "out": {}}
        for name, res in outs.items():
            (c_all, mu_o, like, vals), = res[0]
            ent["out"][name] = {"c_all": np.asarray(c_all).tolist(), "nll": fl(like),
                                "vals": [v if isinstance(v, str) else fl(v) for v in vals]}
        cases.append(ent)
    with open(os.path.join(HERE, "calc_all_c.json"), "w") as f:
        json.dump({"cases": cases}, f, separators=(",", ":"))
    print("wrote calc_all_c.json")


if __name__ == "__main__":
    main()
