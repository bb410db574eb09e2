import os, sys, time, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from minorseq_amd import capi, synth
n, l = 100_000, 3000
sp = synth.SynthParams(seed=2); ref = synth.reference(sp.seed, l)
genes = np.array([(1, l + 1)], dtype=capi.GENE); prm = capi.default_params()
def mk(with_comm):
    jl = capi.Juliet(0); jl.alloc(n, l); jl.synth_fill(sp, ref)
    comm = None
    if with_comm:
        idbuf = np.zeros(128, dtype=np.uint8)
        assert jl.lib.jl_comm_unique_id(idbuf.ctypes.data_as(C.c_void_p)) == 0
        comm = C.c_void_p()
        jl._chk(jl.lib.jl_comm_create(jl.h, idbuf.ctypes.data_as(C.c_void_p), 0, 1, C.byref(comm)))
    return jl, comm
def bench(label, fn, sync, k=300):
    for _ in range(20): fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(k): fn()
    sync()
    print(f"{label:60s} {(time.perf_counter()-t0)/k*1e6:8.1f} us/step", flush=True)
for wc in (False, True):
    for ng in ('', '1'):
        if ng: os.environ['JL_NO_GRAPH'] = '1'
        else: os.environ.pop('JL_NO_GRAPH', None)
        jl, comm = mk(wc)
        bench(f"comm={wc} no_graph={bool(ng)}: run_async back-to-back", lambda: jl.run_async(genes, ref, prm, None, True, 10, True), jl.sync)
        def full():
            jl.run_async(genes, ref, prm, None, True, 10, True); jl.run_fetch(True, True, 64)
        bench(f"comm={wc} no_graph={bool(ng)}: run_async + fetch", full, jl.sync)
        if comm: jl.lib.jl_comm_destroy(comm)
        jl.close()
