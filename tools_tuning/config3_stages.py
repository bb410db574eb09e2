"""configs[3] whole on one GPU (bench.py config3_strong at world 1): where a step's time goes, stage by stage.  Run twice
in one process: the second generation of contexts (after the first were destroyed) is what bench.py measures."""
import ctypes as C
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


def generation(tag):
    win = capi.Juliet(0)
    win.alloc(n, l, win_begin=0)
    win.synth_fill_window(sp, ref)
    win.sync()
    pc = capi.Juliet(0)
    remapped = np.zeros(capi.VARIANT_CAP, dtype=capi.VARIANT)
    pos_global = np.zeros(capi.VARIANT_CAP, dtype=np.uint32)
    vp = C.c_uint32()
    T = {}

    def lap(name, t0):
        t1 = time.perf_counter()
        T[name] = T.get(name, 0.0) + (t1 - t0)
        return t1

    def step():
        t = time.perf_counter()
        win.run_async(genes, ref, prm, None, False, 10, False)
        t = lap("run_async (enqueue)", t)
        v = win.run_fetch(False, False)["variants"]
        t = lap("run_fetch (pileup + call + wait)", t)
        merged = sharding.merge_tables([v], [0])
        t = lap("merge_tables (python)", t)
        arr = (C.c_void_p * 1)(win.h)
        pc._chk(pc.lib.jl_xwin_assemble_local(pc.h, arr, 1, capi._p(merged), len(merged), capi._p(remapped), capi._p(pos_global), C.byref(vp)))
        t = lap("xwin_assemble_local", t)
        pc._shape(n, 3 * vp.value, win.col_stride)
        pc.phase_async(remapped[: len(merged)], 10)
        t = lap("phase_async (enqueue)", t)
        ph = pc.phase_fetch(want_reads=False, cap_var=max(8, len(merged)))
        t = lap("phase_fetch (wait + copy)", t)
        return ph

    for i in range(4):
        step()
    T.clear()
    R = 20
    t0 = time.perf_counter()
    for _ in range(R):
        step()
    tt = (time.perf_counter() - t0) / R
    print(f"--- {tag}")
    for k, v in T.items():
        print(f"{k:36s} {v / R * 1e3:8.3f} ms")
    print(f"{'step':36s} {tt * 1e3:8.3f} ms; pileup kernel alone {win.time_pileup(reps=5):.3f} ms", flush=True)
    if close:
        pc.close()
        win.close()
    else:
        KEEP.append((pc, win))


KEEP = []
close = os.environ.get("C3_KEEP") != "1"
generation("first contexts of the process")
generation("second generation" + ("" if close else " (first kept alive)"))
