import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from minorseq_amd import capi, synth
n, l = 100_000, 3000
jl = capi.Juliet(0)
sp = synth.SynthParams(seed=2); ref = synth.reference(sp.seed, l)
jl.alloc(n, l); jl.synth_fill(sp, ref)
genes = np.array([(1, l + 1)], dtype=capi.GENE)
prm = capi.default_params()
def bench(label, fn, k=200):
    for _ in range(10): fn()
    jl.sync()
    t0 = time.perf_counter()
    for _ in range(k): fn()
    jl.sync()
    print(f"{label:55s} {(time.perf_counter()-t0)/k*1e6:8.1f} us/step", flush=True)
def f_full(): jl.run_async(genes, ref, prm, None, True, 10, True); jl.run_fetch(True, True, 64)
def f_nofetch_sync(): jl.run_async(genes, ref, prm, None, True, 10, True); jl.sync()
def f_async_only(): jl.run_async(genes, ref, prm, None, True, 10, True)
def f_norh(): jl.run_async(genes, ref, prm, None, True, 10, False); jl.run_fetch(True, False, 64)
def f_nophase(): jl.run_async(genes, ref, prm, None, False, 10, False); jl.run_fetch(False, False, 64)
bench('graph: run_async + run_fetch (read_hap)', f_full)
bench('graph: run_async + sync', f_nofetch_sync)
bench('graph: run_async back-to-back (no per-step sync)', f_async_only)
bench('graph: no read_hap copy', f_norh)
bench('graph: phasing off', f_nophase)
os.environ['JL_NO_GRAPH'] = '1'
bench('eager: run_async + run_fetch (read_hap)', f_full)
bench('eager: run_async back-to-back', f_async_only)
