"""Device-clock timeline of the pipelined step loop (JL_TIMELINE=1): per batch, when the pileup / call / phase stages
started and ended on the GPU, merged over the K contexts in flight.  Tuning aid (adds four one-thread nodes per run)."""
import ctypes as C
import os
import sys

import numpy as np

os.environ["JL_TIMELINE"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from minorseq_amd import capi, synth  # noqa: E402

n, l, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
steps = 200
sp = synth.SynthParams(seed=2)
ref = synth.reference(sp.seed, l)
ctxs = []
for _ in range(K):
    c = capi.Juliet(0)
    c.alloc(n, l)
    c.synth_fill(sp, ref)
    ctxs.append(c)
genes = np.array([(1, l + 1)], dtype=capi.GENE)
prm = capi.default_params()
for i in range(steps):
    c = ctxs[i % K]
    if i >= K:
        c.run_wait()
    c.run_async(genes, ref, prm, None, True, 10, True)
rows = []
for k, c in enumerate(ctxs):
    c.run_wait()
    buf = np.zeros((4096, 8), dtype=np.uint64)
    c.lib.jl_debug_timeline.argtypes = [C.c_void_p, C.c_void_p]
    assert c.lib.jl_debug_timeline(c.h, buf.ctypes.data_as(C.c_void_p)) == 0
    for r in range(steps // K):
        rows.append((buf[r, 0], k, r, buf[r, :4].copy()))
rows.sort(key=lambda x: x[0])
rows = rows[len(rows) // 2: len(rows) // 2 + 14]
t0 = int(rows[0][0])
print("times in us from the first row; stage ends: pileup / call / phase")
prev_pile_end = None
for _, k, r, s in rows:
    a = [(int(x) - t0) / 100.0 for x in s]
    gap = "" if prev_pile_end is None else f" (pileup start - previous pileup end = {a[0] - prev_pile_end:6.1f})"
    print(f"ctx {k} run {r:3d}: start {a[0]:8.1f}  pileup {a[1] - a[0]:6.1f}  call {a[2] - a[1]:6.1f}  phase {a[3] - a[2]:6.1f}{gap}")
    prev_pile_end = a[1]
span = (int(rows[-1][0]) - t0) / 100.0 / (len(rows) - 1)
print(f"{span:.1f} us per step over this window")
