"""Is the fused phase launch bound by instruction fetch?  Its code is 55 KB and runs once, front to back; between two
launches of a step loop a pileup streams 150 MB through the L2.  Here the phase stage runs alone, back to back (code warm
in the L2 / instruction cache), then with a pileup in between: compare the kernel's durations in a rocprofv3 kernel trace
(tools_tuning/trace_gaps.py)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from minorseq_amd import capi, synth  # noqa: E402

n, l = 100_000, 3000
ref = synth.reference(2, l)
genes = np.array([(1, l + 1)], dtype=capi.GENE)
jl = capi.Juliet(0)
jl.alloc(n, l)
jl.synth_fill(synth.SynthParams(seed=1000), ref)
jl.sync()
jl.pileup_async(genes, ref)
jl.call_async()
tab = jl.call_fetch()
for _ in range(6):          # A: phase alone, back to back
    jl.phase_async(None, 10)
    jl.sync()
for _ in range(6):          # B: a pileup between two phase launches
    jl.pileup_async(genes, ref)
    jl.call_async()
    jl.phase_async(None, 10)
    jl.sync()
print("done", len(tab))
