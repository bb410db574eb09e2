"""Record ingest on the device: 100k x 3000 synthetic records (juliet-synth --raw-out) -> planes, `reps` builds rotating over
four record copies and two windows; run under rocprofv3 --kernel-trace --stats for the per-kernel times.
usage: ingest_time.py [reads] [cols] [reps] [min_qv]   (min_qv > 0: rich-QV records, qualities resident)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from minorseq_amd import capi, synth, msa  # noqa: E402

if os.environ.get("JL_LIB"):   # a tuning build of the library (tools_tuning/build_tuning_lib.sh)
    capi.load_library(os.environ["JL_LIB"])

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
l = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
min_qv = int(sys.argv[4]) if len(sys.argv) > 4 else 0
# min_qv > 0: the documented `ccs --richQVs` shape — filtered bases keep their letter and carry a low quality (ten ops a read)
rec = synth.raw_records(2, n, l, extra=("--rich-qv",) if min_qv else ())
print(f"{len(rec['cigar']) / n:.1f} ops per read, {sum(v.nbytes for k, v in rec.items() if min_qv or not k.startswith('qual')) / 1e6:.1f} MB of records",
      flush=True)
recs = []
for k in range(4):
    c = capi.Juliet(0)
    if min_qv:
        c.records_upload(rec["pos"], rec["cigar"], rec["cig_off"], rec["seq4"], rec["seq_off"], rec["qual"], rec["qual_off"])
    else:
        c.records_upload(rec["pos"], rec["cigar"], rec["cig_off"], rec["seq4"], rec["seq_off"])
    recs.append(c)
wins = [capi.Juliet(0), capi.Juliet(0)]
for w in wins:
    w.records_window(recs[0], l, 0, min_qv)
if n <= 20000:
    rows = synth.rows(synth.SynthParams(seed=2), l, 0, n, synth.reference(2, l))
    assert (msa.unpack_columns(wins[0].download_columns(), n) == rows).all()
    print("matrix = synth.rows", flush=True)
one = os.environ.get("JL_ING_ONE_STREAM") == "1"     # builds one after the other on ONE stream: every kernel alone on the device
t0 = time.perf_counter()
for q in range(reps):
    wins[0 if one else q % 2].records_window(recs[q % 4], l, 0, min_qv, wait=False)
for w in wins:
    w.sync()
print(f"{reps} builds: {1e6 * (time.perf_counter() - t0) / reps:.1f} us per build (host clock, {'one stream' if one else 'two streams'})", flush=True)
