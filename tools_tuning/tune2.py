import os, sys
sys.path.insert(0, '.')
import numpy as np
from minorseq_amd import capi, synth
def run(n, l, genes, label, variants):
    jl = capi.Juliet(0)
    sp = synth.SynthParams(seed=2)
    ref = synth.reference(sp.seed, l)
    jl.alloc(n, l); jl.synth_fill(sp, ref)
    for (w, pipe, waves, rs) in variants:
        os.environ['JL_PILEUP_W'] = str(w); os.environ['JL_PILEUP_PIPE'] = str(pipe); os.environ['JL_PILEUP_WAVES'] = str(waves)
        os.environ['JL_PILEUP_RSPLIT'] = str(rs)
        jl.pileup_async(genes, ref); jl.sync()
        ts = [jl.time_pileup(30) for _ in range(3)]
        t = min(ts)
        print(f"{label} W={w:3d} pipe={pipe} waves={waves} rsplit={rs}: {t*1e3:8.1f} us  {n*l/2/t/1e6:8.1f} GB/s", flush=True)
    jl.close()
g1 = np.array([(1, 3001)], dtype=capi.GENE)
run(100_000, 3000, g1, 'C2', [(6,0,1,0),(106,0,1,0),(206,0,1,0),(3,0,1,0),(9,0,1,0),(6,0,1,1),(6,0,1,2),(6,0,1,3),(6,0,1,4),(106,0,1,1),(106,0,1,2),(106,0,1,4),(3,0,1,1),(3,0,1,2),(9,0,1,3),(12,0,1,3),(12,0,1,4)])
