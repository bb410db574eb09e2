"""Where the device ingest of juliet-synth's raw records differs from synth.rows: reads / columns / sweeps of the mismatches."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from minorseq_amd import capi, msa, synth
if os.environ.get("JL_LIB"):
    capi.load_library(os.environ["JL_LIB"])
n, l = int(sys.argv[1]), int(sys.argv[2])
rec = synth.raw_records(2, n, l)
rows = synth.rows(synth.SynthParams(seed=2), l, 0, n, synth.reference(2, l))
jl = capi.Juliet(0)
jl.ingest_records(l, 0, rec["pos"], rec["cigar"], rec["cig_off"], rec["seq4"], rec["seq_off"])
got = msa.unpack_columns(jl.download_columns(), n)
bad = np.argwhere(got != rows)
print("mismatches:", len(bad))
if len(bad):
    r, c = bad[:, 0], bad[:, 1]
    print("reads (mod 128) histogram:", np.bincount(r % 128, minlength=128).tolist())
    print("columns / 8 (mod 32) histogram:", np.bincount((c // 8) % 32, minlength=32).tolist())
    print("first:", bad[:10].tolist())
    for rr in sorted(set(r.tolist()))[:3]:
        cc = c[r == rr]
        print(" read", rr, "cols", cc.min(), "..", cc.max(), "n", len(cc), "seq_off", int(rec["seq_off"][rr]) % 16)
        lo = max(0, cc.min() - 8); hi = min(l, cc.max() + 9)
        print("   got", "".join(str(x) for x in got[rr, lo:hi]))
        print("   exp", "".join(str(x) for x in rows[rr, lo:hi]))
