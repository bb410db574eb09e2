"""What a bench.py `collect(final=True)` costs the host once the launch is complete: the pieces, timed one by one."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from minorseq_amd import capi, synth  # noqa: E402

n, l, G = 100_000, 3000, 8
ref = synth.reference(2, l)
genes = np.array([(1, l + 1)], dtype=capi.GENE)
prm = capi.default_params()
ctxs = []
for k in range(G):
    c = capi.Juliet(0)
    c.alloc(n, l)
    c.synth_fill(synth.SynthParams(seed=1000 + k), ref)
    c.sync()
    ctxs.append(c)
grp = capi.Group(ctxs)
grp.run_async(genes, ref, prm, True, 10, True)
exp = [bench.signature(c.run_view()) for c in ctxs]
acc = dict(views=0.0, run_view=0.0, same=0.0, sync=0.0)
R = 200
for _ in range(R):
    grp.run_async(genes, ref, prm, True, 10, True)
    for c in ctxs:
        c.run_wait()
    time.sleep(0.0005)          # the launch is long complete
    t0 = time.perf_counter()
    vw = grp.views()
    ok = vw["complete"].all() and (vw["n_variants"] == 5).all()
    t1 = time.perf_counter()
    a = ctxs[3].run_view()
    b = ctxs[7].run_view()
    t2 = time.perf_counter()
    s = bench.same(exp[3], a)
    t3 = time.perf_counter()
    ctxs[0].sync()
    t4 = time.perf_counter()
    assert ok and s
    acc["views"] += t1 - t0
    acc["run_view"] += t2 - t1
    acc["same"] += t3 - t2
    acc["sync"] += t4 - t3
print({k: round(1e6 * v / R, 2) for k, v in acc.items()}, "us per final collect of 8 windows")
