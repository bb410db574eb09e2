"""jl_ctx_create behind an initialised runtime (torch has created its context; no host-to-device copy yet): its own time, then the
first large pageable upload.  With JL_LIB = a -DJL_TUNING build and JL_CTX_TIMES=1 the library prints its steps."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
torch.cuda.init()
x = torch.empty(1 << 20, device="cuda")
torch.cuda.synchronize()
from minorseq_amd import capi  # noqa: E402

capi.load_library(os.environ.get("JL_LIB"))
t0 = time.perf_counter()
jl = capi.Juliet(0)
t1 = time.perf_counter()
a = np.ones(64 << 20, dtype=np.uint8)
d = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
t2 = time.perf_counter()
d.copy_(torch.from_numpy(a))
torch.cuda.synchronize()
t3 = time.perf_counter()
print(f"jl_ctx_create {1e3 * (t1 - t0):6.2f} ms, first 64 MB pageable copy {1e3 * (t3 - t2):6.2f} ms")
