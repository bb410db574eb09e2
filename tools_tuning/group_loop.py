"""Throughput of the step loop with group runs: NG groups of G windows in flight (MI355X box)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from minorseq_amd import capi, synth  # noqa: E402

if os.environ.get("JL_LIB"):
    capi.load_library(os.environ["JL_LIB"])

n, l, G, NG = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
rounds = 100
sp = synth.SynthParams(seed=2)
ref = synth.reference(sp.seed, l)
groups = []
for _ in range(NG):
    ctxs = []
    for _ in range(G):
        c = capi.Juliet(0)
        c.alloc(n, l)
        c.synth_fill(sp, ref)
        c.sync()
        ctxs.append(c)
    groups.append(capi.Group(ctxs))
genes = np.array([(1, l + 1)], dtype=capi.GENE)
prm = capi.default_params()
skip = os.environ.get("JL_GROUP_SKIP", "0") in ("1", "2")   # stages skipped: no result block, only the completion word


def collect(c):
    return c.run_wait() if skip else c.run_view()


for g in groups:
    for _ in range(3):
        g.run_async(genes, ref, prm, True, 10, True)
        for c in g.ctxs:
            collect(c)
t = dict(launch=0, wait=0)
T0 = time.perf_counter_ns()
for i in range(rounds):
    g = groups[i % NG]
    if i >= NG:
        t0 = time.perf_counter_ns()
        for c in g.ctxs:
            collect(c)
        t["wait"] += time.perf_counter_ns() - t0
    t0 = time.perf_counter_ns()
    g.run_async(genes, ref, prm, True, 10, True)
    t["launch"] += time.perf_counter_ns() - t0
for g in groups:
    for c in g.ctxs:
        c.run_wait()
T1 = time.perf_counter_ns()
steps = rounds * G
v = None if skip else groups[0].ctxs[0].run_view()
print(f"{n}x{l} groups of {G}, {NG} in flight: {(T1 - T0) / steps / 1000:.1f} us/step ({(T1 - T0) / rounds / 1000:.1f} us per group);",
      {k: round(x / rounds / 1000, 2) for k, x in t.items()}, "variants", None if v is None else len(v["variants"]),
      "haplotypes", None if v is None or "phase" not in v else v["phase"]["summary"]["n_haplotypes"],
      "JL_GROUP_SKIP", os.environ.get("JL_GROUP_SKIP"))
