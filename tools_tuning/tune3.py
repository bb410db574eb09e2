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
        ts = [jl.time_pileup(10 if n > 500000 else 30) for _ in range(3)]
        t = min(ts)
        print(f"{label} W={w:3d} pipe={pipe} waves={waves} rsplit={rs}: {t*1e3:8.1f} us  {n*l/2/t/1e6:8.1f} GB/s", flush=True)
    jl.close()
def g(l, frames=1): return np.array([(1 + f, l + 1) for f in range(frames)], dtype=capi.GENE)
V = [(3,0,1,0),(3,1,1,0),(6,0,1,0),(12,0,1,0),(3,0,2,0),(6,0,2,0),(3,0,4,0)]
run(100_000, 3000, g(3000,3), 'C2-3frame', [(3,0,1,0),(6,0,1,0),(6,0,1,1),(12,0,1,0),(12,0,1,1),(12,0,1,2)])
run(1_000_000, 1250, g(1250), 'C4/gpu 1M x 1250', V)
run(4_000_000, 1215, g(1215), 'C5ish 4M x 1215', V)
run(1_000, 3000, g(3000), 'C1 1k x 3000', V)
run(10_000, 3000, g(3000), '10k x 3000', V)
run(100_000, 9719, g(9719), '100k x 9719', [(3,0,1,0),(6,0,1,0),(3,0,2,0),(6,0,2,0)])
