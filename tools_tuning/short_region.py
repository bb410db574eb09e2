"""Where does a SHORT timed region go?  The driver runs `bench.py --steps 20 --warmup 5`: 20 windows, 0.7 ms.
Repeats that region R times back to back in one process (with an idle gap before each, like the fence of bench.py) and
prints, per region: host time to issue the launches, time until each launch's last window completes, total.
usage: short_region.py <schedule> [idle_ms]   schedule = comma list of windows per launch, e.g. 8,8,4 or 5,5,5,5 or 20"""
import os
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from minorseq_amd import capi, synth  # noqa: E402

if os.environ.get("JL_LIB"):
    capi.load_library(os.environ["JL_LIB"])

sched = [int(x) for x in sys.argv[1].split(",")]
idle_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
n, l = 100_000, 3000
sp = synth.SynthParams(seed=2)
ref = synth.reference(sp.seed, l)
genes = np.array([(1, l + 1)], dtype=capi.GENE)
prm = capi.default_params()
groups = []
for g in sched:
    ctxs = []
    for _ in range(g):
        c = capi.Juliet(0)
        c.alloc(n, l)
        c.synth_fill(sp, ref)
        c.sync()
        ctxs.append(c)
    groups.append(capi.Group(ctxs) if g > 1 else ctxs[0])


def launch(g):
    if isinstance(g, capi.Group):
        g.run_async(genes, ref, prm, True, 10, True)
        return g.ctxs
    g.run_async(genes, ref, prm, None, True, 10, True)
    return [g]


for g in groups:
    for _ in range(3):
        for c in launch(g):
            c.run_view()
R = 12
for r in range(R):
    if idle_ms:
        time.sleep(idle_ms / 1e3)
    t0 = time.perf_counter_ns()
    members = [launch(g) for g in groups]
    t1 = time.perf_counter_ns()
    done = []
    for m in members:
        for c in m:
            c.run_view()
        done.append(time.perf_counter_ns())
    steps = sum(sched)
    print(f"region {r}: issue {(t1 - t0) / 1e3:6.1f} us; launches done at " + " ".join(f"{(d - t0) / 1e3:6.1f}" for d in done) +
          f" us; {(done[-1] - t0) / 1e3 / steps:5.2f} us/step", flush=True)
