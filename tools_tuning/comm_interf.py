import os, sys, time, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from minorseq_amd import capi, synth
n, l = 100_000, 3000
sp = synth.SynthParams(seed=2); ref = synth.reference(sp.seed, l)
genes = np.array([(1, l + 1)], dtype=capi.GENE); prm = capi.default_params()
def mk(with_comm, nn=n):
    jl = capi.Juliet(0); jl.alloc(nn, l); jl.synth_fill(sp, ref)
    comm = None
    if with_comm:
        idbuf = np.zeros(128, dtype=np.uint8)
        assert jl.lib.jl_comm_unique_id(idbuf.ctypes.data_as(C.c_void_p)) == 0
        comm = C.c_void_p()
        jl._chk(jl.lib.jl_comm_create(jl.h, idbuf.ctypes.data_as(C.c_void_p), 0, 1, C.byref(comm)))
    return jl, comm
S = 4
cs = [mk(False) for _ in range(S)]
side, scomm = mk(True, 1000)
side.run_async(genes, ref, prm, None, True, 10, True); side.sync()
rows = np.zeros(4096, dtype=capi.VARIANT); cnt = np.zeros(1, dtype=np.uint32)
for mode in ('none', 'side-nccl-every-step', 'side-nccl-every-4'):
    def steps(k):
        for i in range(k):
            jl, _ = cs[i % S]
            if i >= S: jl.run_fetch(True, True, 64)
            jl.run_async(genes, ref, prm, None, True, 10, True)
            if mode == 'side-nccl-every-step' or (mode == 'side-nccl-every-4' and i % 4 == 0):
                side.lib.jl_allgather_variants_async(side.h, scomm)
        for jl, _ in cs: jl.run_fetch(True, True, 64)
        side.sync()
    steps(40)
    t0 = time.perf_counter(); steps(400); dt = time.perf_counter() - t0
    print(f"{mode}: {dt/400*1e6:8.1f} us/step", flush=True)
