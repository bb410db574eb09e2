"""The FP64 Fisher routine the device runs (csrc/jl_fisher.h), compiled for the host, against the mpmath
golden vectors and the long-double oracle.  The GPU parity test (test_gpu_parity.py) repeats this on device."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("shim") / "libfisher_shim.so")
    subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-o", out,
                           os.path.join(HERE, "csrc", "fisher_shim.cpp")])
    lib = C.CDLL(out)
    lib.shim_fisher.restype = C.c_double
    lib.shim_fisher.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]
    return lib


def test_device_algorithm_vs_golden(shim):
    with open(os.path.join(HERE, "golden", "fisher_golden.json")) as f:
        tables = json.load(f)["tables"]
    n_checked, worst = 0, 0.0
    for r in tables:
        if r["a"] + r["b"] != r["c"] + r["d"]:
            continue  # the device routine is specialised to juliet's equal-row tables
        lp = C.c_double()
        p = shim.shim_fisher(r["a"], r["c"], r["a"] + r["b"], C.byref(lp))
        gp, glp = float(r["p"]), float(r["log_p"])
        assert abs(p - gp) <= 1e-10, (r, p)
        if gp > 1e-300:
            rel = abs(p - gp) / gp
            worst = max(worst, rel)
            assert rel <= 5e-12, (r, p, gp)
        assert abs(lp.value - glp) <= 1e-12 * max(1.0, abs(glp)) + 1e-13, (r, lp.value, glp)
        n_checked += 1
    assert n_checked > 800
    print("checked", n_checked, "worst relative error", worst)


def test_device_algorithm_vs_oracle_random(shim, oracle):
    rng = np.random.default_rng(42)
    for _ in range(3000):
        n = int(rng.integers(1, 200000))
        a = int(rng.integers(1, min(n, 400) + 1))
        c = int(rng.integers(0, min(n, 50) + 1))
        lp = C.c_double()
        p = shim.shim_fisher(a, c, n, C.byref(lp))
        op, olp = oracle.fisher(a, n - a, c, n - c)
        assert abs(p - op) <= 1e-10
        assert abs(lp.value - olp) <= 1e-9 * max(1.0, abs(olp)), (a, c, n, lp.value, olp)
