"""Host cost of Juliet.run_view / run_wait on a completed run (MI355X box)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from minorseq_amd import capi, synth  # noqa: E402

n, l = 100000, 3000
sp = synth.SynthParams(seed=2)
ref = synth.reference(sp.seed, l)
c = capi.Juliet(0)
c.alloc(n, l)
c.synth_fill(sp, ref)
genes = np.array([(1, l + 1)], dtype=capi.GENE)
c.run_async(genes, ref, capi.default_params(), None, True, 10, True)
c.run_wait()
for name, fn in (("run_view", c.run_view), ("run_wait", c.run_wait), ("run_done", c.run_done)):
    fn()
    t0 = time.perf_counter_ns()
    for _ in range(2000):
        fn()
    print(f"{name}: {(time.perf_counter_ns() - t0) / 2000 / 1000:.2f} us per call")
