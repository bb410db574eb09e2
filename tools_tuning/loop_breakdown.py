"""Where the bench loop's wall time goes with K batches in flight: launch / wait / fetch per step (MI355X box)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from minorseq_amd import capi, synth  # noqa: E402

n, l, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
steps = 400
VIEW = os.environ.get('VIEW', '1') == '1'
sp = synth.SynthParams(seed=2)
ref = synth.reference(sp.seed, l)
ctxs = []
for _ in range(K):
    c = capi.Juliet(0)
    c.alloc(n, l)
    c.synth_fill(sp, ref)
    ctxs.append(c)
genes = np.array([(1, l + 1)], dtype=capi.GENE)
prm = capi.default_params()
for c in ctxs:
    for _ in range(3):
        c.run_async(genes, ref, prm, None, True, 10, True)
        c.run_fetch(True, True, cap_var=64)
t = dict(launch=0, wait=0, fetch=0)
T0 = time.perf_counter_ns()
for i in range(steps):
    c = ctxs[i % K]
    if i >= K:
        t0 = time.perf_counter_ns()
        if VIEW:
            c.run_wait()
        else:
            c.sync()
        t1 = time.perf_counter_ns()
        if VIEW:
            c.run_view()
        else:
            c.run_fetch(True, True, cap_var=64)
        t2 = time.perf_counter_ns()
        t["wait"] += t1 - t0
        t["fetch"] += t2 - t1
    t0 = time.perf_counter_ns()
    c.run_async(genes, ref, prm, None, True, 10, True)
    t["launch"] += time.perf_counter_ns() - t0
for c in ctxs:
    c.sync()
T1 = time.perf_counter_ns()
print(f"view={VIEW} {n}x{l} inflight {K}: {(T1 - T0) / steps / 1000:.1f} us/step;",
      {k: round(v / steps / 1000, 2) for k, v in t.items()})
