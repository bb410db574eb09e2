"""Where the device ingest differs from the rows it was built from (the first case of test_device_ingest_matches_rows)."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from minorseq_amd import capi, msa, synth
from test_gpu_parity import rows_to_records
if os.environ.get("JL_LIB"):   # a tuning build of the library: its kernels check every address and report instead of faulting
    capi.load_library(os.environ["JL_LIB"])
n, l, partial, win = 300, 120, 0.3, (0, 120)
if len(sys.argv) > 1:
    n, l, partial, win = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), (int(sys.argv[4]), int(sys.argv[5]))
rng = np.random.default_rng(n + l)
sp = synth.SynthParams(seed=n + l, partial_rate=partial, del_rate=0.02, mask_rate=0.03, sub_rate=0.01)
ref = synth.reference(sp.seed, l)
rows = synth.rows(sp, l, 0, n, ref)
rows[5, 40:60] = 6
rows[7] = 6
pos, cigar, cig_off, seq4, seq_off, qual, qual_off = rows_to_records(rows, ref, rng)
b, e = win
jl = capi.Juliet(0)
OPS = "MIDNSHP=X"
for rep in range(3):
    jl.ingest_records(e - b, b, pos, cigar, cig_off, seq4, seq_off)
    got = msa.unpack_columns(jl.download_columns(), n)
    bad = np.argwhere(got != rows[:, b:e])
    print("rep", rep, "mismatches:", len(bad), "reads:", sorted(set(bad[:, 0].tolist()))[:20])
    for r in sorted(set(bad[:, 0].tolist()))[:4]:
        cols = bad[bad[:, 0] == r][:, 1]
        print(" read", r, "pos", pos[r], "cols", cols.min(), "..", cols.max(), "n", len(cols),
              "cigar", "".join(f"{int(w) >> 4}{OPS[int(w) & 15]}" for w in cigar[int(cig_off[r]):int(cig_off[r + 1])]))
        print("   got", "".join(str(x) for x in got[r]))
        print("   exp", "".join(str(x) for x in rows[r, b:e]))
