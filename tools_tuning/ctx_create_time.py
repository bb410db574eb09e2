"""What a context costs: the process's first jl_ctx_create (the HIP runtime starts in it) and the ones after it.   (on the GPU box)"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
t0 = time.perf_counter()
from minorseq_amd import capi  # noqa: E402
t1 = time.perf_counter()
cs = []
ts = []
for i in range(4):
    a = time.perf_counter()
    cs.append(capi.Juliet(0))
    ts.append(1e3 * (time.perf_counter() - a))
print("import %.1f ms; jl_ctx_create: first %.1f ms, then %s ms" % (1e3 * (t1 - t0), ts[0], " ".join("%.1f" % t for t in ts[1:])))
