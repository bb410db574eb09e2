"""Isolated grouped pileup launch (8 windows x 100k x 3 kb) for the library named by JL_LIB: compares builds."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from minorseq_amd import capi, synth  # noqa: E402

if os.environ.get("JL_LIB"):
    capi.load_library(os.environ["JL_LIB"])
n, l, G, NG = 100_000, 3000, 8, 4
sp = synth.SynthParams(seed=2)
ref = synth.reference(sp.seed, l)
genes = np.array([(1, l + 1)], dtype=capi.GENE)
prm = capi.default_params()
groups = []
for _ in range(NG):
    ctxs = []
    for _ in range(G):
        c = capi.Juliet(0)
        c.alloc(n, l)
        c.synth_fill(sp, ref)
        c.sync()
        ctxs.append(c)
    g = capi.Group(ctxs)
    g.run_async(genes, ref, prm, True, 10, True)
    for c in ctxs:
        c.run_view()
    groups.append(g)
for reps in (200, 1000):
    ms, nb = capi.time_pileup_groups(groups, reps=reps)
    print(os.environ.get("JL_LIB", "default"), f"reps {reps}: {ms * 1e3:.1f} us per launch, {nb / ms / 1e9:.2f} TB/s, frac {nb / ms / 1e6 / 8e6:.3f}", flush=True)
