"""What collecting a finished exchange costs on the host (one-rank communicator): jl_allgather_variants long after the
exchange completed, single contexts and a batch of 8."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from minorseq_amd import capi, synth  # noqa: E402

n, l = 100_000, 3000
sp = synth.SynthParams(seed=2)
ref = synth.reference(sp.seed, l)
genes = np.array([(1, l + 1)], dtype=capi.GENE)
prm = capi.default_params()
ctxs = []
for k in range(8):
    c = capi.Juliet(0)
    c.alloc(n, l)
    c.synth_fill(sp, ref)
    c.sync()
    ctxs.append(c)
jl = ctxs[0]
idbuf = np.zeros(128, dtype=np.uint8)
assert jl.lib.jl_comm_unique_id(idbuf.ctypes.data_as(C.c_void_p)) == 0
comm = C.c_void_p()
jl._chk(jl.lib.jl_comm_create(jl.h, idbuf.ctypes.data_as(C.c_void_p), 0, 1, C.byref(comm)))
rows = np.zeros(8 * 128, dtype=capi.VARIANT)
counts = np.zeros(8, dtype=np.uint32)
g = capi.Group(ctxs)
arr = (C.c_void_p * 8)(*[c.h for c in ctxs])
for rep in range(5):
    g.run_async(genes, ref, prm, True, 10, True)
    jl._chk(jl.lib.jl_allgather_variants_async_many(arr, 8, comm))
    for c in ctxs:
        c.run_view()
    time.sleep(0.01)      # the exchange is long done
    t = []
    for c in ctxs:
        t0 = time.perf_counter_ns()
        c._chk(c.lib.jl_allgather_variants(c.h, comm, capi._p(rows), capi._p(counts), 128))
        t.append((time.perf_counter_ns() - t0) / 1e3)
    print("batch of 8, collected one by one (us):", " ".join(f"{x:.1f}" for x in t), flush=True)
for rep in range(3):
    g.run_async(genes, ref, prm, True, 10, True)
    jl._chk(jl.lib.jl_allgather_variants_async_many(arr, 8, comm))
    for c in ctxs:
        c.run_view()
    time.sleep(0.01)
    t0 = time.perf_counter_ns()
    jl._chk(jl.lib.jl_allgather_variants_many(arr, 8, comm, capi._p(rows), capi._p(counts), 128))
    print(f"batch of 8, one call: {(time.perf_counter_ns() - t0) / 1e3:.1f} us", flush=True)
jl.lib.jl_comm_destroy(comm)
