"""Randomised check of the device ingest against synth.rows: many seeds, indel and mask rates, window offsets, with and
without qualities (the helpers of tests/test_gpu_parity.py).  usage: ingest_stress.py [rounds [seed]]"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from minorseq_amd import capi, msa, synth  # noqa: E402
from test_gpu_parity import rows_to_records  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 24
jl = capi.Juliet(0)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
for k in range(rounds):
    n = int(rng.integers(1, 6000))
    l = int(rng.integers(30, 1500))
    sp = synth.SynthParams(seed=1000 + k, partial_rate=float(rng.uniform(0, 0.6)), del_rate=float(rng.choice([0.0, 0.002, 0.05, 0.3])),
                           mask_rate=float(rng.choice([0.0, 0.02, 0.4])), sub_rate=0.01)
    ref = synth.reference(sp.seed, l)
    rows = synth.rows(sp, l, 0, n, ref)
    if n > 8:
        rows[5] = 6
        rows[3, : l // 2] = 6
    rec = rows_to_records(rows, ref, rng)
    pos, cigar, cig_off, seq4, seq_off, qual, qual_off = rec
    b = int(rng.integers(0, max(1, l // 2)))
    e = int(rng.integers(b + 1, l + 1))
    jl.ingest_records(e - b, b, pos, cigar, cig_off, seq4, seq_off)
    got = msa.unpack_columns(jl.download_columns(), n)
    assert (got == rows[:, b:e]).all(), (k, n, l, b, e)
    jl.ingest_records(e - b, b, pos, cigar, cig_off, seq4, seq_off, qual, qual_off, min_qv=50)
    assert (msa.unpack_columns(jl.download_columns(), n) == rows[:, b:e]).all(), ("qv", k)
    print(f"round {k}: {n} reads x {l} columns, window [{b}, {e}), del {sp.del_rate}, mask {sp.mask_rate}: ok", flush=True)
print("all rounds ok")
