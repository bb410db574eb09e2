"""Time the pileup kernel alone (HIP events, jl_time_pileup) for a window shape; env JL_PILEUP_* select variants."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from minorseq_amd import capi, synth  # noqa: E402

n, l = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100_000, 3000)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
sp = synth.SynthParams(seed=2)
ref = synth.reference(sp.seed, l)
jl = capi.Juliet(0)
jl.alloc(n, l)
jl.synth_fill(sp, ref)
genes = np.array([(1, l + 1)], dtype=capi.GENE)
jl.pileup_async(genes, ref)
jl.sync()
jl.time_pileup(reps=10)
ms = jl.time_pileup(reps=reps)
print(f"{n} x {l}: pileup {ms * 1000:.2f} us  = {n * l / 2 / (ms * 1e-3) / 1e12:.2f} TB/s  env={ {k: v for k, v in os.environ.items() if k.startswith('JL_')} }")
