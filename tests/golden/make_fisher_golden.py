#!/usr/bin/env python3
"""Generates tests/golden/fisher_golden.json — run in the dev container only (needs mpmath + scipy).

The reference ships no golden vectors for its Fisher's-exact test (doc/JULIET.md:38-42
only names the test), so the pin is an independent 60-digit evaluation of the hypergeometric upper
tail P(X >= a) for the 2x2 table [[a, b], [c, d]], cross-checked against scipy.stats.fisher_exact
where scipy does not underflow (scipy itself is only good to ~1e-9 relative at 1e7 coverage, so the
cross-check is loose; mpmath is the pin).  Nothing here travels to the GPU box; only the JSON does.
"""
import json
import random

import mpmath as mp
from scipy.stats import fisher_exact

mp.mp.dps = 60


def upper_tail(a, b, c, d):
    M, K, n = a + b + c + d, a + c, a + b
    hi = min(K, n)
    den = mp.binomial(M, n)
    s = mp.mpf(0)
    for x in range(a, hi + 1):
        s += mp.binomial(K, x) * mp.binomial(M - K, n - x)
    return s / den


def main():
    rnd = random.Random(20170516)
    tables = []
    # tiny tables, all regimes
    for _ in range(60):
        tables.append(tuple(rnd.randint(0, 30) for _ in range(4)))
    # juliet-shaped: both rows sum to the coverage, second row = expected under the error model
    for cov in (50, 100, 1000, 2529, 2907, 2998, 6000, 25000, 100000, 1000000, 10000000):
        for e in sorted({1, 2, 3, 7, max(1, int(cov * 5.8e-5) + 1), max(1, int(cov * 1e-3))}):
            if e > cov:
                continue
            obs = sorted({1, 2, 3, 5, 8, 13, 20, 21, 22, 29, 40, 75, e, e + 1, max(1, e - 1), 2 * e + 3,
                          max(1, cov // 100), max(1, cov // 10), max(1, cov // 2), cov - 1, cov})
            for a in obs:
                if 1 <= a <= cov:
                    tables.append((a, cov - a, e, cov - e))
    # the example probed in SURVEY.md §8c
    tables.append((29, 2500, 1, 2528))
    out = []
    seen = set()
    for t in tables:
        if t in seen or sum(t) == 0 or t[0] + t[1] == 0 or t[0] + t[2] == 0:
            continue
        seen.add(t)
        a, b, c, d = t
        p = upper_tail(a, b, c, d)
        rec = {"a": a, "b": b, "c": c, "d": d, "p": mp.nstr(p, 25), "log_p": mp.nstr(mp.log(p), 25)}
        if p > mp.mpf("1e-290"):
            sp = fisher_exact([[a, b], [c, d]], alternative="greater")[1]
            assert abs(sp - float(p)) <= 1e-7 * float(p) + 1e-300, (t, sp, float(p))
            rec["scipy"] = repr(float(sp))
        out.append(rec)
    with open(__file__.replace("make_fisher_golden.py", "fisher_golden.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_fisher_golden.py", "dps": 60, "tail": "greater",
                   "tables": out}, f, indent=0)
    print(len(out), "tables")


if __name__ == "__main__":
    main()
