"""333 333 reads x 1000 columns through the device ingest against synth.rows (326 groups of 1024 reads: the groups do not divide among the eight XCDs evenly)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from minorseq_amd import capi, synth, msa
n, l = 333_333, 1000
rec = synth.raw_records(5, n, l)
jl = capi.Juliet(0)
jl.records_upload(rec["pos"], rec["cigar"], rec["cig_off"], rec["seq4"], rec["seq_off"])
w = capi.Juliet(0)
w.records_window(jl, l, 0, 0)
rows = synth.rows(synth.SynthParams(seed=5), l, 0, n, synth.reference(5, l))
got = msa.unpack_columns(w.download_columns(), n)
print("equal:", bool((got == rows).all()), got.shape)
