"""A window with more than ten variant positions: 100k reads x 3 kb with 16 positions (49 variants, ~125 haplotypes), the
case the reference's own screenshots show (doc/JULIET.md:350, 362).  Up to 20 positions take the two-word fused launch;
`generic` as argument forces the multi-word pipeline for comparison (what every such window took before)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from minorseq_amd import capi, msa, synth  # noqa: E402

if os.environ.get("JL_LIB"):
    capi.load_library(os.environ["JL_LIB"])

n, l = 100_000, 3000
sp = synth.SynthParams(seed=1000)
ref = synth.reference(2, l)
jl = capi.Juliet(0)
jl.alloc(n, l)
jl.synth_fill(sp, ref)
rows = msa.unpack_columns(jl.download_columns(), n)
rng = np.random.default_rng(3)
for k in range(11):                       # eleven more edited codons, about 3 % of the reads each
    who = rng.choice(n, n // 30, replace=False)
    c0 = 3 * (100 + 61 * k)
    rows[who, c0:c0 + 3] = (rows[who, c0:c0 + 3] + 1 + k % 3) % 4
jl.upload_rows(rows)
genes = np.array([(1, l + 1)], dtype=capi.GENE)
prm = capi.default_params()
for _ in range(4):
    jl.run_async(genes, ref, prm, None, True, 10, True)
    out = jl.run_view() or jl.run_fetch(True, True, cap_var=64)
s = out["phase"]["summary"]
t0 = time.perf_counter()
R = 100
for _ in range(R):
    jl.run_async(genes, ref, prm, None, True, 10, True)
    v = jl.run_view()
    if v is None:
        v = jl.run_fetch(True, True, cap_var=64)
t = (time.perf_counter() - t0) / R
print(f"{s['n_positions']} positions, {len(out['variants'])} variants, {s['n_haplotypes']} haplotypes: {t * 1e6:.1f} us per run "
      f"({'zero-copy view' if jl.run_view() is not None else 'copying fetch'})")
if "--check" in sys.argv:
    import oracle_lib
    orc = oracle_lib.load()
    ev = orc.call(rows, genes, refseq=ref)
    ep = orc.phase(rows, ev)
    got = jl.run_fetch(True, True, cap_var=64)
    assert ep["summary"] == got["phase"]["summary"], (ep["summary"], got["phase"]["summary"])
    assert (ep["hap_count"] == got["phase"]["hap_count"]).all() and (ep["read_hap"] == got["phase"]["read_hap"]).all()
    assert (ep["hap_pattern"] == got["phase"]["hap_pattern"]).all()
    print("matches the oracle")
if hasattr(jl.lib, "jl_debug_stamps"):   # -DJL_EXP_STAMPS build: device-clock stamps of the last fused phase launch
    import ctypes as C
    st = np.zeros(64, dtype=np.uint64)
    jl.lib.jl_debug_stamps(st.ctypes.data_as(C.c_void_p))
    t0 = int(st[0])
    for k, nm in ((0, "entry (workgroup 0)"), (1, "plan"), (2, "keys built"), (3, "dominant key (last round)"), (4, "LDS table (last round)"),
                  (5, "global inserts done"), (6, "before arrival"), (7, "after arrival"), (8, "LAST: start"), (9, "LAST: categories"),
                  (13, "SELECT: groups scanned"), (23, "SELECT: two-word keys"), (24, "SELECT: rank counted"), (25, "SELECT: ranked"), (14, "SELECT: hit"), (26, "SELECT: resident hap arrays"), (15, "SELECT: resident arrays"), (16, "SELECT: result blocks"),
                  (17, "SELECT: tables emptied"), (19, "LAST: flag released"), (20, "a waiter saw the flag (latest)"),
                  (21, "ids stored (latest)"), (22, "completion word stored")):
        print(f"  {nm:32s} {(int(st[k]) - t0) / 100.0:8.2f}")
    print("  groups in the table:", int(st[18]))
