"""configs[3] whole on one GPU through the reads-sharded phasing path (bench.py config3_strong at world 1): stage times."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from minorseq_amd import capi, sharding, synth  # noqa: E402

n, l = 1_000_000, 10_000
sp = synth.SynthParams(seed=4)
ref = synth.reference(sp.seed, l)
genes = np.array([(1, 3 * (l // 3) + 1)], dtype=capi.GENE)
prm = capi.default_params(n_tests=sharding.default_n_tests(genes))
win = capi.Juliet(0)
win.alloc(n, l, win_begin=0)
win.synth_fill_window(sp, ref)
win.sync()
pc = capi.Juliet(0)
T = {}


def lap(name, t0):
    t1 = time.perf_counter()
    T[name] = T.get(name, 0.0) + (t1 - t0)
    return t1


def step():
    t = time.perf_counter()
    win.run_async(genes, ref, prm, None, False, 10, False)
    v = win.run_fetch(False, False)["variants"]
    t = lap("pileup + call + fetch", t)
    merged = sharding.merge_tables([v], [0])
    t = lap("merge_tables", t)
    remapped, pos_global = pc.xwin_assemble_slice_local([win], merged, 0, n)
    t = lap("assemble slice", t)
    pc.phase_groups_async(remapped)
    t = lap("groups_async (enqueue)", t)
    mine = pc.phase_groups_fetch()
    t = lap("groups_fetch", t)
    patterns, gcounts, index = sharding.merge_groups([mine])
    t = lap("merge_groups (numpy)", t)
    ph = sharding.select_haplotypes(patterns, gcounts, remapped, mine["pos_cols"], 10, [mine["summary"]])
    t = lap("select_haplotypes (numpy)", t)
    pc.phase_regroup(ph["hap_of_merged"][index[0]].astype(np.uint16), ph["summary"]["n_haplotypes"], False)
    t = lap("regroup", t)
    return mine, ph


for _ in range(4):
    mine, ph = step()
T.clear()
R = 20
t0 = time.perf_counter()
for _ in range(R):
    step()
tt = (time.perf_counter() - t0) / R
for k, v in T.items():
    print(f"{k:36s} {v / R * 1e3:8.3f} ms")
print(f"{'step':36s} {tt * 1e3:8.3f} ms; groups {len(mine['counts'])}, haplotypes {ph['summary']['n_haplotypes']}")
