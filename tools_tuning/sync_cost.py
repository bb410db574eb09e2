"""Cost of hipStreamSynchronize on an already-idle stream (second of two back-to-back syncs), MI355X box."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from minorseq_amd import capi, synth  # noqa: E402
n, l = 2000, 300
sp = synth.SynthParams(seed=2)
ref = synth.reference(sp.seed, l)
c = capi.Juliet(0); c.alloc(n, l); c.synth_fill(sp, ref)
genes = np.array([(1, l + 1)], dtype=capi.GENE); prm = capi.default_params()
for _ in range(3):
    c.run_async(genes, ref, prm, None, True, 10, True); c.sync()
a = b = 0
for _ in range(200):
    c.run_async(genes, ref, prm, None, True, 10, True)
    time.sleep(0.0005)   # the batch is long finished
    t0 = time.perf_counter_ns(); c.sync(); t1 = time.perf_counter_ns(); c.sync(); t2 = time.perf_counter_ns()
    a += t1 - t0; b += t2 - t1
print(f"sync after completed work: {a/200/1000:.2f} us; second sync on idle stream: {b/200/1000:.2f} us")
