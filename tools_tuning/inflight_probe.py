import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from minorseq_amd import capi, synth
n, l = 100_000, 3000
sp = synth.SynthParams(seed=2); ref = synth.reference(sp.seed, l)
genes = np.array([(1, l + 1)], dtype=capi.GENE); prm = capi.default_params()
import torch; torch.cuda.set_device(0)
for S in (4, 6):
    cs = []
    for _ in range(S):
        c = capi.Juliet(0); c.alloc(n, l); c.synth_fill(sp, ref); cs.append(c)
    for label, phasing, rh in (('call+phase+ids', True, True), ('call+phase', True, False), ('call only', False, False)):
        def steps(k):
            for i in range(k):
                c = cs[i % S]
                if i >= S: c.run_fetch(phasing, rh, 64)
                c.run_async(genes, ref, prm, None, phasing, 10, rh)
            for c in cs: c.run_fetch(phasing, rh, 64)
        steps(40)
        t0 = time.perf_counter(); steps(600); dt = time.perf_counter() - t0
        print(f"inflight={S} {label:16s}: {dt/600*1e6:7.1f} us/step", flush=True)
    # host-side cost of one launch+fetch pair with the GPU idle-ish
    c = cs[0]
    t0 = time.perf_counter()
    for _ in range(300):
        c.run_async(genes, ref, prm, None, True, 10, True)
    th = (time.perf_counter() - t0) / 300
    c.sync()
    print(f"  host time of run_async alone: {th*1e6:.1f} us", flush=True)
    for c in cs: c.close()
