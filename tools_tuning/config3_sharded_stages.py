"""configs[3] whole on one GPU through the session (bench.py config3_strong at world 1): host time by stage
(jl_xwin_stage_us) next to the step and the pileup kernel.  `python tools_tuning/config3_sharded_stages.py [comm]`:
with `comm` the session gets a one-rank communicator, i.e. every collective is issued (all but the wire)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from minorseq_amd import capi, sharding, synth  # noqa: E402

if os.environ.get("JL_LIB"):   # an experimental build of the library
    capi.load_library(os.environ["JL_LIB"])

n, l = 1_000_000, 10_000
sp = synth.SynthParams(seed=4)
ref = synth.reference(sp.seed, l)
genes = np.array([(1, 3 * (l // 3) + 1)], dtype=capi.GENE)
prm = capi.default_params(n_tests=sharding.default_n_tests(genes))
win = capi.Juliet(0)
win.alloc(n, l, win_begin=0)
win.synth_fill_window(sp, ref)
win.sync()
comm = None
if len(sys.argv) > 1 and sys.argv[1] == "comm":
    idbuf = np.zeros(128, dtype=np.uint8)
    assert win.lib.jl_comm_unique_id(idbuf.ctypes.data_as(C.c_void_p)) == 0
    comm = C.c_void_p()
    win._chk(win.lib.jl_comm_create(win.h, idbuf.ctypes.data_as(C.c_void_p), 0, 1, C.byref(comm)))
xw = capi.Xwin([win], [0], [l], [0], [0, n], comm)
T, t_run = {}, 0.0


def step():
    global t_run
    t0 = time.perf_counter()
    win.run_async(genes, ref, prm, None, False, 10, False)
    t_run += time.perf_counter() - t0
    r = xw.phase_raw(10)
    for k, v in xw.stage_us().items():
        T[k] = T.get(k, 0.0) + v
    return r


for _ in range(4):
    step()
T.clear()
t_run = 0.0
R = 20
t0 = time.perf_counter()
for _ in range(R):
    r = step()
tt = (time.perf_counter() - t0) / R
t_k = win.time_pileup(reps=5)
print(f"{'jl_run_async (enqueue)':40s} {t_run / R * 1e6:8.1f} us")
for k, v in T.items():
    print(f"{k:40s} {v / R:8.1f} us")
print(f"{'step':40s} {tt * 1e6:8.1f} us; pileup kernel {t_k * 1e3:.1f} us; residue {tt * 1e6 - t_k * 1e3:.1f} us; "
      f"groups {r.n_groups}, haplotypes {r.n_haplotypes}, communicator {'yes' if comm else 'no'}")

if hasattr(win.lib, "jl_debug_stamps"):   # -DJL_EXP_STAMPS build: device-clock stamps of the last fused phase launch
    st = np.zeros(64, dtype=np.uint64)
    win.lib.jl_debug_stamps(st.ctypes.data_as(C.c_void_p))
    names = ["entry", "plan read", "keys built", "dominant key", "LDS table", "global inserts + slots", "before arrival",
             "after arrival", "LAST: start", "LAST: categories", "LAST: export done", "LAST: stores drained", "LAST: signalled"]
    t0 = int(st[0])
    print("fused phase launch, workgroup 0 then the last arriver (us after workgroup 0's entry):")
    for k, nm in enumerate(names):
        print(f"  {nm:28s} {(int(st[k]) - t0) / 100.0:8.2f}")
