"""Randomised check of the device ingest's QUALITY path against the numpy statement (tests/records_expand.py): juliet-synth raw
records in the rich-QV shape with insertions, clips, extra poor qualities and deletion rates drawn at random, random windows that
begin and end inside the reads, random thresholds — every cell.  usage: ingest_stress_qv.py [rounds [seed]]"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import records_expand  # noqa: E402
from minorseq_amd import capi, msa, synth  # noqa: E402

if os.environ.get("JL_LIB"):
    capi.load_library(os.environ["JL_LIB"])
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 606)
jl = capi.Juliet(0)
w = capi.Juliet(0)
for k in range(rounds):
    n = int(rng.integers(1, 9000))
    l = int(rng.integers(40, 2600))
    extra = ["--rich-qv"] if rng.random() < 0.7 else []
    extra += ["--ins-ppm", str(int(rng.choice([0, 800, 5000, 30000])))]
    extra += ["--low-qv-ppm", str(int(rng.choice([0, 20000, 300000])))]
    extra += ["--del", str(float(rng.choice([0.0, 0.0013, 0.01, 0.08])))]
    extra += ["--partial", str(float(rng.choice([0.0, 0.3, 0.9])))]
    if rng.random() < 0.5:
        extra += ["--clips"]
    rec = synth.raw_records(3000 + k, n, l, extra=tuple(extra))
    jl.records_upload(rec["pos"], rec["cigar"], rec["cig_off"], rec["seq4"], rec["seq_off"], rec["qual"], rec["qual_off"])
    for _ in range(2):
        b = int(rng.integers(0, max(1, l // 2)))
        e = int(rng.integers(b + 1, l + 1))
        min_qv = int(rng.choice([1, 5, 13, 20, 60, 94, 127]))
        w.records_window(jl, e - b, b, min_qv)
        got = msa.unpack_columns(w.download_columns(), n)
        exp = records_expand.expand(rec, e - b, b, min_qv)
        if not (got == exp).all():
            bad = np.argwhere(got != exp)[0]
            raise SystemExit(f"round {k}: read {bad[0]} column {bad[1]}: got {got[bad[0], bad[1]]}, expected {exp[bad[0], bad[1]]}  ({n} x {l}, window [{b}, {e}), min_qv {min_qv}, {extra})")
    jl.records_drop()
    print(f"round {k}: {n} reads x {l} columns, {len(rec['cigar']) / n:.1f} ops a read, {' '.join(extra)}: ok", flush=True)
print("all rounds ok")
