"""Pins the oracle's Fisher's-exact test (doc/JULIET.md:38-42) against the committed mpmath golden vectors."""
import json
import math
import os

import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "fisher_golden.json")


def tables():
    with open(GOLDEN) as f:
        return json.load(f)["tables"]


def test_golden_file_shape():
    t = tables()
    assert len(t) > 900
    assert any(r["a"] + r["b"] == 10_000_000 for r in t)


def test_oracle_matches_golden(oracle):
    worst = 0.0
    for r in tables():
        p, lp = oracle.fisher(r["a"], r["b"], r["c"], r["d"])
        gp, glp = float(r["p"]), float(r["log_p"])
        # absolute 1e-10 is BASELINE.json's bar; the relative bound is what the long-double path really gives
        assert abs(p - gp) <= 1e-10, r
        if gp > 1e-300:
            rel = abs(p - gp) / gp
            worst = max(worst, rel)
            assert rel <= 2e-10, (r, p, gp)
        assert abs(lp - glp) <= 2e-10 * max(1.0, abs(glp)), (r, lp, glp)
    print("worst relative error", worst)


def test_survey_probe(oracle):
    # fisher_exact([[29,2500],[1,2528]], 'greater') = 2.6775665826890507e-08
    p, lp = oracle.fisher(29, 2500, 1, 2528)
    assert p == pytest.approx(2.6775665826890507e-08, rel=1e-12)
    assert lp == pytest.approx(math.log(2.6775665826890507e-08), rel=1e-13)


def test_threshold_anchor(oracle):
    """SPEC §5 sanity anchor: e=1, alpha=0.01, n in [982,1884] => smallest callable count is 21 (A.3: 0.72 % of 2907)."""
    cov = 2907
    for n in (982, 1000, 1884):
        called = [a for a in range(1, 40) if oracle.fisher(a, cov - a, 1, cov - 1)[0] * n < 0.01]
        assert called[0] == 21


def test_degenerate(oracle):
    assert oracle.fisher(0, 10, 0, 10)[0] == 1.0
    assert oracle.fisher(10, 0, 10, 0)[0] == 1.0
    p, _ = oracle.fisher(10, 0, 0, 10)
    assert p == pytest.approx(1.0 / math.comb(20, 10), rel=1e-12)
