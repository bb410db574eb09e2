import os, sys
sys.path.insert(0, '.')
import numpy as np
from minorseq_amd import capi, synth
n, l = 100_000, 3000
jl = capi.Juliet(0); sp = synth.SynthParams(seed=2); ref = synth.reference(2, l)
jl.alloc(n, l); jl.synth_fill(sp, ref)
layouts = {
 'single frame': [(1, 3001)],
 'three genes, three frames, no overlap': [(1, 1000), (1002, 2000), (2003, 3001)],
 'HIV-like: frames differ + 150-col overlap': [(1, 1000), (852, 2000), (2001, 3001)],
 'all three frames everywhere': [(1, 3001), (2, 3000), (3, 3001)],
}
for name, g in layouts.items():
    genes = np.array(g, dtype=capi.GENE)
    for w in ('', '6'):
        if w: os.environ['JL_PILEUP_W'] = w
        else: os.environ.pop('JL_PILEUP_W', None)
        jl.pileup_async(genes, ref); jl.sync()
        t = min(jl.time_pileup(30) for _ in range(3))
        print(f"{name:45s} W={'auto' if not w else w:4s}: {t*1e3:6.1f} us", flush=True)
        # force a re-plan next time (the plan is cached by gene list)
        jl.pileup_async(np.array([(1, 4)], dtype=capi.GENE), ref)
