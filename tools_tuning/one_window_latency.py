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
