"""Host time per step of the bench loop, split by call (run on the MI355X box).  The GPU is drained before each
fetch, so the numbers are pure host cost: ctypes + HIP runtime enqueue + result copies."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from minorseq_amd import capi, synth  # noqa: E402

n, l = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100_000, 3000)
sp = synth.SynthParams(seed=2)
ref = synth.reference(sp.seed, l)
jl = capi.Juliet(0)
jl.alloc(n, l)
jl.synth_fill(sp, ref)
genes = np.array([(1, l + 1)], dtype=capi.GENE)
prm = capi.default_params()
for _ in range(5):
    jl.run_async(genes, ref, prm, None, True, 10, True)
    jl.run_fetch(True, True, cap_var=64)
K = 200
t = dict(launch=0, sync=0, call_fetch=0, phase_fetch=0, py_fetch=0)
b = jl._bufs(64)
q = b["ptr"]
for _ in range(K):
    t0 = time.perf_counter_ns()
    jl.run_async(genes, ref, prm, None, True, 10, True)
    t1 = time.perf_counter_ns()
    jl.sync()
    t2 = time.perf_counter_ns()
    jl.lib.jl_call_fetch(jl.h, q["variants"], capi.VARIANT_CAP, b["n_ref"])
    t3 = time.perf_counter_ns()
    jl.lib.jl_phase_fetch(jl.h, q["summ"], q["pos_cols"], q["hap_count"], q["hap_pattern"], q["hit"], q["read_hap"], q["cooc"], 64)
    t4 = time.perf_counter_ns()
    jl.run_fetch(True, True, cap_var=64)
    t5 = time.perf_counter_ns()
    t["launch"] += t1 - t0
    t["sync"] += t2 - t1
    t["call_fetch"] += t3 - t2
    t["phase_fetch"] += t4 - t3
    t["py_fetch"] += t5 - t4
print(f"{n} reads x {l} cols; per step, us:", {k: round(v / K / 1000, 2) for k, v in t.items()})
