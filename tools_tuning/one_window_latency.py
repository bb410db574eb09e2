"""One window alone through jl_run_async (what `juliet in.bam out.json` does): latency of the whole path, 100k reads x 3 kb,
phasing on, per-read ids to the host.  JL_LIB selects an experimental build of the library."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from minorseq_amd import capi, synth  # noqa: E402

if os.environ.get("JL_LIB"):
    capi.load_library(os.environ["JL_LIB"])
n, l = 100_000, 3000
vp_extra = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ref = synth.reference(2, l)
genes = np.array([(1, l + 1)], dtype=capi.GENE)
prm = capi.default_params()
jl = capi.Juliet(0)
jl.alloc(n, l)
jl.synth_fill(synth.SynthParams(seed=1000), ref)
jl.sync()
for _ in range(5):
    jl.run_async(genes, ref, prm, None, True, 10, True)
    v = jl.run_view()
t0 = time.perf_counter()
R = 200
for _ in range(R):
    jl.run_async(genes, ref, prm, None, True, 10, True)
    v = jl.run_view()
t = (time.perf_counter() - t0) / R
print(f"one window: {t * 1e6:.1f} us per run; {len(v['variants'])} variants, {v['phase']['summary']['n_haplotypes']} haplotypes")
if hasattr(jl.lib, "jl_debug_stamps"):   # -DJL_EXP_STAMPS build: device-clock stamps of the last fused phase launch
    import ctypes as C
    st = np.zeros(64, dtype=np.uint64)
    jl.lib.jl_debug_stamps(st.ctypes.data_as(C.c_void_p))
    t0 = int(st[0])
    for k, nm in ((0, "entry (workgroup 0)"), (1, "plan"), (2, "keys built"), (3, "dominant key"), (4, "LDS table"), (5, "global inserts done"),
                  (6, "before arrival"), (7, "after arrival"), (8, "LAST: start"), (9, "LAST: categories"), (13, "SELECT: groups scanned"),
                  (14, "SELECT: ranked + hit"), (15, "SELECT: resident arrays"), (16, "SELECT: result blocks"), (17, "SELECT: tables emptied"),
                  (19, "LAST: flag released"), (20, "a waiter saw the flag (latest)"), (21, "ids stored (latest)"), (22, "completion word stored")):
        print(f"  {nm:32s} {(int(st[k]) - t0) / 100.0:8.2f}")
