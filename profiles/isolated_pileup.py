"""The dominant kernel ALONE, for a profile whose average is one population (the bench's stats file pools
isolated and pipelined launches).  Same residency as bench.py's default (4 groups of 8 windows of 100k reads x 3 kb,
every window different reads); every group runs once (its argument tables), then ONLY the back-to-back rotating launches
of jl_group_time_pileup.  Under `rocprofv3 --kernel-trace --stats` the `pileup_fold_group_kernel` row (round 6: the
pileup launch carries the Fisher stage in its epilogue; `JL_NO_FOLD_CALL=1`: `pileup_planes_group_kernel`) is
4 (set-up) + 4 (warm-up) + REPS launches of the same shape, none overlapping another kernel.

usage: python3 profiles/isolated_pileup.py [reps]   -> one JSON line with the HIP-event average"""
import json
import os
import sys

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from minorseq_amd import capi, synth  # noqa: E402

if os.environ.get("JL_LIB"):   # tuning aid: an alternative build of the library
    capi.load_library(os.environ["JL_LIB"])
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n, l, G, NG = 100_000, 3000, 8, 4
ref = synth.reference(2, l)
genes = np.array([(1, l + 1)], dtype=capi.GENE)
prm = capi.default_params()
groups = []
for u in range(NG):
    ctxs = []
    for k in range(G):
        c = capi.Juliet(0)
        c.alloc(n, l)
        c.synth_fill(synth.SynthParams(seed=1000 + u * G + k), ref)
        c.sync()
        ctxs.append(c)
    g = capi.Group(ctxs)
    g.run_async(genes, ref, prm, True, 10, True)
    for c in ctxs:
        c.run_view()
    groups.append(g)
ms, nbytes = capi.time_pileup_groups(groups, reps=reps)
# nbytes = algorithmic bytes of one launch: 3 bits per cell of the resident bit planes, every cell read once
print(json.dumps({"kernel": "pileup_fold_group_kernel (the pileup with the Fisher stage in its epilogue: what the runs launch)", "launches": reps, "kernel_ms": ms, "algorithmic_bytes_per_launch": nbytes,
                  "achieved_GBs": nbytes / (ms * 1e-3) / 1e9, "frac_of_8TBs": nbytes / (ms * 1e-3) / 1e9 / 8000.0,
                  "frac_in_nibble_units": nbytes * (4.0 / 3.0) / (ms * 1e-3) / 1e9 / 8000.0}))
