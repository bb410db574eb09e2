"""Host time of one window's jl_run_async / run_view in the latency loop (one_window_latency.py's set-up)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from minorseq_amd import capi, synth  # noqa: E402
n, l = 100_000, 3000
ref = synth.reference(2, l)
genes = np.array([(1, l + 1)], dtype=capi.GENE)
prm = capi.default_params()
jl = capi.Juliet(0)
jl.alloc(n, l)
jl.synth_fill(synth.SynthParams(seed=1000), ref)
jl.sync()
for _ in range(5):
    jl.run_async(genes, ref, prm, None, True, 10, True)
    jl.run_view()
R = 300
ta = tw = tv = 0
t_all = time.perf_counter_ns()
for _ in range(R):
    t0 = time.perf_counter_ns()
    jl.run_async(genes, ref, prm, None, True, 10, True)
    t1 = time.perf_counter_ns()
    jl.run_wait()
    t2 = time.perf_counter_ns()
    jl.run_view()
    t3 = time.perf_counter_ns()
    ta += t1 - t0; tw += t2 - t1; tv += t3 - t2
t_all = time.perf_counter_ns() - t_all
print(f"per run: {t_all / R / 1e3:.1f} us = run_async {ta / R / 1e3:.1f} + wait {tw / R / 1e3:.1f} + view {tv / R / 1e3:.1f} (+ loop)")
