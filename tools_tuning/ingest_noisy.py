"""The record ingest on reads with many indels (juliet-synth --ins-ppm / --del): how many (read, sweep) pairs leave the tiles for
the column-by-column path, and what the kernels take.  usage: ingest_noisy.py reads cols ins_ppm del_rate   (under rocprofv3 for the times)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from minorseq_amd import capi, synth, msa
import records_expand
if os.environ.get("JL_LIB"):
    capi.load_library(os.environ["JL_LIB"])
n, l, ins_ppm, del_rate = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
rec = synth.raw_records(7, n, l, extra=("--ins-ppm", ins_ppm, "--del", del_rate))
print(f"{len(rec['cigar']) / n:.1f} ops per read", flush=True)
jl = capi.Juliet(0)
jl.records_upload(rec["pos"], rec["cigar"], rec["cig_off"], rec["seq4"], rec["seq_off"])
w = capi.Juliet(0)
for rep in range(4):
    t0 = time.perf_counter()
    w.records_window(jl, l, 0, 0)
    print(f"build {rep}: {1e6 * (time.perf_counter() - t0):.0f} us (host clock, blocking)", flush=True)
m = min(n, 20000)
exp = records_expand.expand(rec, l, read_end=m)
got = msa.unpack_columns(np.ascontiguousarray(w.download_columns()[:, : (m + 1) // 2]), m)
print("first", m, "reads equal:", bool((got == exp).all()))
