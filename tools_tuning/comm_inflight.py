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
rows = np.zeros(4096, dtype=capi.VARIANT); cnt = np.zeros(1, dtype=np.uint32)
pr, pc = rows.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p)
for wc in (False, True):
    for S in (1, 2, 4):
        cs = [mk(wc) for _ in range(S)]
        def steps(k):
            for i in range(k):
                jl, comm = cs[i % S]
                if i >= S:
                    if comm: jl.lib.jl_allgather_variants(jl.h, comm, pr, pc, 4096)
                    jl.run_fetch(True, True, 64)
                jl.run_async(genes, ref, prm, None, True, 10, True)
            for jl, comm in cs:
                if comm: jl.lib.jl_allgather_variants(jl.h, comm, pr, pc, 4096)
                jl.run_fetch(True, True, 64)
        steps(40)
        t0 = time.perf_counter(); steps(400); dt = time.perf_counter() - t0
        print(f"comm={wc} inflight={S}: {dt/400*1e6:8.1f} us/step", flush=True)
        for jl, comm in cs:
            if comm: jl.lib.jl_comm_destroy(comm)
            jl.close()
