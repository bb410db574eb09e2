"""What torch.cuda.synchronize() costs when nothing is pending, as a function of how many contexts (streams) exist."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from minorseq_amd import capi, synth  # noqa: E402

n, l = 100_000, 3000
sp = synth.SynthParams(seed=2)
ref = synth.reference(sp.seed, l)
genes = np.array([(1, l + 1)], dtype=capi.GENE)
prm = capi.default_params()
ctxs = []
for target in (1, 8, 20, 32):
    while len(ctxs) < target:
        c = capi.Juliet(0)
        c.alloc(n, l)
        c.synth_fill(sp, ref)
        c.sync()
        ctxs.append(c)
    for rep in range(3):
        for c in ctxs:
            c.run_async(genes, ref, prm, None, True, 10, True)
        for c in ctxs:
            c.run_view()
        t0 = time.perf_counter_ns()
        torch.cuda.synchronize()
        t1 = time.perf_counter_ns()
        torch.cuda.synchronize()
        t2 = time.perf_counter_ns()
    print(f"{target} contexts: synchronize right after the last result {(t1 - t0) / 1e3:.1f} us, again {(t2 - t1) / 1e3:.1f} us")
