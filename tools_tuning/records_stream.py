"""Upload of aligned records in one call vs in chunks (jl_records_begin / _append / _finish): where the time goes."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from minorseq_amd import capi  # noqa: E402

n, l = 100_000, 3000
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.default_rng(1)
# every read: one '=' run over the whole window
pos = np.zeros(n, dtype=np.int32)
cigar = np.full(n, (l << 4) | 7, dtype=np.uint32)
cig_off = np.arange(n + 1, dtype=np.uint64)
seq4 = rng.integers(0, 256, size=n * (l // 2), dtype=np.uint8)
seq4 = (np.uint8(1) << (seq4 & 3)) | ((np.uint8(1) << ((seq4 >> 2) & 3)) << 4)
seq_off = np.arange(n + 1, dtype=np.uint64) * (l // 2)
jl = capi.Juliet(0)
_p = capi._p
for rep in range(3):
    t0 = time.perf_counter()
    jl.ingest_records(l, 0, pos, cigar, cig_off, seq4, seq_off)
    t1 = time.perf_counter()
    print(f"one call: {(t1 - t0) * 1e3:.1f} ms")
    for hints in ((0, 0, 0, 0), (n, n, len(seq4), 0)):
        t0 = time.perf_counter()
        jl._chk(jl.lib.jl_records_begin(jl.h, *hints))
        tb = time.perf_counter()
        ta = []
        for a in range(0, n, chunk):
            b = min(n, a + chunk)
            t = time.perf_counter()
            jl._chk(jl.lib.jl_records_append(jl.h, b - a, _p(pos[a:b]), _p(cigar), _p(cig_off[a:b + 1]), _p(seq4), _p(seq_off[a:b + 1]), None, None))
            ta.append((time.perf_counter() - t) * 1e3)
        tf = time.perf_counter()
        jl._chk(jl.lib.jl_records_finish(jl.h, l, 0, 0))
        te = time.perf_counter()
        print(f"chunks of {chunk}, hints {hints}: begin {(tb - t0) * 1e3:.1f} ms, {len(ta)} appends {sum(ta):.1f} ms (max {max(ta):.1f}, median {sorted(ta)[len(ta) // 2]:.2f}), finish {(te - tf) * 1e3:.1f} ms")
