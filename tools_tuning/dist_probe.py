import os, sys, time, ctypes as C
sys.path.insert(0, '.')
import numpy as np
import torch; torch.cuda.set_device(0)
from minorseq_amd import capi, synth
n, l = 100_000, 3000
sp = synth.SynthParams(seed=2); ref = synth.reference(sp.seed, l)
genes = np.array([(1, l + 1)], dtype=capi.GENE); prm = capi.default_params()
S = 4
cs = []
for _ in range(S):
    c = capi.Juliet(0); c.alloc(n, l); c.synth_fill(sp, ref); cs.append(c)
idbuf = np.zeros(128, dtype=np.uint8)
assert cs[0].lib.jl_comm_unique_id(idbuf.ctypes.data_as(C.c_void_p)) == 0
comm = C.c_void_p()
cs[0]._chk(cs[0].lib.jl_comm_create(cs[0].h, idbuf.ctypes.data_as(C.c_void_p), 0, 1, C.byref(comm)))
rows = np.zeros(4096, dtype=capi.VARIANT); cnt = np.zeros(1, dtype=np.uint32)
pr, pc = rows.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p)
for mode in ('nocomm', 'comm'):
    T = dict(run=0.0, agasync=0.0, ag=0.0, fetch=0.0)
    def steps(k):
        for i in range(k):
            c = cs[i % S]
            if i >= S:
                t = time.perf_counter(); c.run_fetch(True, True, 64); T['fetch'] += time.perf_counter() - t
                if mode == 'comm' and i >= 2 * S:   # collect the exchange of the PREVIOUS use of this context
                    t = time.perf_counter(); c.lib.jl_allgather_variants(c.h, comm, pr, pc, 4096); T['ag'] += time.perf_counter() - t
            t = time.perf_counter(); c.run_async(genes, ref, prm, None, True, 10, True); T['run'] += time.perf_counter() - t
            if mode == 'comm':
                t = time.perf_counter(); c.lib.jl_allgather_variants_async(c.h, comm); T['agasync'] += time.perf_counter() - t
        for c in cs:
            c.run_fetch(True, True, 64)
            if mode == 'comm':
                c.lib.jl_allgather_variants(c.h, comm, pr, pc, 4096); c.lib.jl_allgather_variants(c.h, comm, pr, pc, 4096)
    steps(40)
    for k in T: T[k] = 0.0
    t0 = time.perf_counter(); steps(600); dt = time.perf_counter() - t0
    print(mode, f"{dt/600*1e6:.1f} us/step;", {k: round(v / 600 * 1e6, 1) for k, v in T.items()}, flush=True)
cs[0].lib.jl_comm_destroy(comm)
