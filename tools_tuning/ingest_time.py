"""Record ingest on the device: kernel time of jl_records_finish for 100k x 3000 synthetic records (and parity of the
two builds of the kernel when JL_TUNING switches are compiled in)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from minorseq_amd import capi, synth, msa  # noqa: E402

n, l = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000, int(sys.argv[2]) if len(sys.argv) > 2 else 3000
sp = synth.SynthParams(seed=2)
ref = synth.reference(sp.seed, l)
t0 = time.perf_counter()
rows = synth.rows(sp, l, 0, n, ref)
# records: per read runs of = / X / D, N for uncovered stretches inside, soft clips outside
pos = np.zeros(n, dtype=np.int32)
cig, cig_off, seqs, seq_off = [], [0], [], [0]
nt16 = np.array([1, 2, 4, 8, 15, 15], dtype=np.uint8)
for r in range(n):
    row = rows[r]
    cov = np.flatnonzero(row != 6)
    if len(cov) == 0:
        cig_off.append(len(cig)); seq_off.append(seq_off[-1]); continue
    b, e = cov[0], cov[-1] + 1
    pos[r] = b
    seg = row[b:e]
    kind = np.where(seg == 4, 2, np.where(seg == 6, 3, np.where(seg == ref[b:e], 7, 8))).astype(np.uint32)
    ch = np.flatnonzero(np.diff(kind)) + 1
    st = np.concatenate(([0], ch)); en = np.concatenate((ch, [len(kind)]))
    cig.extend(((en - st).astype(np.uint32) << 4 | kind[st]).tolist())
    cig_off.append(len(cig))
    bases = nt16[seg[(seg != 4) & (seg != 6)]]
    if len(bases) & 1:
        bases = np.concatenate((bases, [0]))
    packed = (bases[0::2] << 4) | bases[1::2]
    seqs.append(packed.astype(np.uint8))
    seq_off.append(seq_off[-1] + len(packed))
cigar = np.array(cig, dtype=np.uint32); cig_off = np.array(cig_off, dtype=np.uint64)
seq4 = np.concatenate(seqs); seq_off = np.array(seq_off, dtype=np.uint64)
print(f"records built in {time.perf_counter() - t0:.1f} s: {len(cigar) / n:.1f} ops per read", flush=True)
jl = capi.Juliet(0)
want = msa.pack_columns(rows) if hasattr(msa, "pack_columns") else None
for rep in range(4):
    jl._chk(jl.lib.jl_records_begin(jl.h, n, len(cigar), len(seq4), 0))
    jl._chk(jl.lib.jl_records_append(jl.h, n, capi._p(pos), capi._p(cigar), capi._p(cig_off), capi._p(seq4), capi._p(seq_off), None, None))
    t = time.perf_counter()
    jl._chk(jl.lib.jl_records_finish(jl.h, l, 0, 0))
    dt = time.perf_counter() - t
    jl._shape(n, l, jl.lib.jl_col_stride(n))
    got = msa.unpack_columns(jl.download_columns(), n)
    print(f"finish {dt * 1e3:.2f} ms; matrix equals the rows: {bool((got == rows).all())}", flush=True)
