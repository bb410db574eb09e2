import os, sys, itertools, json
sys.path.insert(0, '.')
import numpy as np
from minorseq_amd import capi, synth
def run(n, l, genes, label, variants):
    jl = capi.Juliet(0)
    sp = synth.SynthParams(seed=2)
    ref = synth.reference(sp.seed, l)
    jl.alloc(n, l); jl.synth_fill(sp, ref)
    base = None
    for (w, pipe, waves) in variants:
        os.environ['JL_PILEUP_W'] = str(w); os.environ['JL_PILEUP_PIPE'] = str(pipe); os.environ['JL_PILEUP_WAVES'] = str(waves)
        jl.pileup_async(genes, ref)
        pf = jl.pileup_fetch()
        chk = (int(pf['col_counts'].astype(np.uint64).sum()), int(pf['hist'].astype(np.uint64).sum()), int((pf['hist'].astype(np.uint64) * np.arange(64, dtype=np.uint64)).sum()))
        if base is None: base = chk
        ts = [jl.time_pileup(20) for _ in range(3)]
        t = min(ts)
        print(f"{label} W={w} pipe={pipe} waves={waves}: {t*1e3:8.1f} us  {n*l/2/t/1e6:8.1f} GB/s  ok={chk==base}", flush=True)
    jl.close()
variants = [(w, p, wv) for w in (6, 12) for p in (0, 1) for wv in (1, 2, 4)]
run(100_000, 3000, np.array([(1, 3001)], dtype=capi.GENE), 'C2 100k x 3000 1-frame', variants)
run(100_000, 3000, np.array([(1, 3001), (2, 3000), (3, 3001)], dtype=capi.GENE), 'C2 100k x 3000 3-frame', variants)
run(4_000_000, 1215, np.array([(1, 1216)], dtype=capi.GENE), 'C5ish 4M x 1215 1-frame', [(6,0,1),(6,1,1),(12,0,1),(12,1,1),(6,0,2),(6,0,4)])
