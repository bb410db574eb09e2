import os, sys
sys.path.insert(0, '.')
import numpy as np
from minorseq_amd import capi, synth
def run(n, l, label, variants):
    jl = capi.Juliet(0)
    sp = synth.SynthParams(seed=2); ref = synth.reference(sp.seed, l)
    jl.alloc(n, l); jl.synth_fill(sp, ref)
    genes = np.array([(1, l + 1)], dtype=capi.GENE)
    for rep in range(2):
      for (w, waves, rs) in variants:
        os.environ['JL_PILEUP_W'] = str(w); os.environ['JL_PILEUP_WAVES'] = str(waves); os.environ['JL_PILEUP_RSPLIT'] = str(rs)
        jl.pileup_async(genes, ref); jl.sync()
        t = min(jl.time_pileup(10 if n > 500000 else 40) for _ in range(3))
        print(f"{label} W={w:3d} waves={waves} rsplit={rs}: {t*1e3:8.1f} us  {n*l/2/t/1e6:8.1f} GB/s", flush=True)
    jl.close()
run(100_000, 3000, 'C2', [(3,1,0),(403,1,0),(103,1,0),(6,1,0),(406,1,0)])
run(4_000_000, 1215, 'C5ish', [(3,4,0),(403,4,0),(6,4,0),(406,4,0)])
