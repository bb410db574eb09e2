import os, sys
import numpy as np
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import oracle_lib
from minorseq_amd import capi, msa, synth
capi.load_library(os.path.join(R, "tools_tuning", "lib_exp", "libjuliet_hip.so"))
n, l = 5000, 300
sp = synth.SynthParams(seed=31, minor_permille=(70, 60, 50, 40), partial_rate=0.1)
ref = synth.reference(sp.seed, l)
rows = synth.rows(sp, l, 0, n, ref)
genes = np.array([(1, l + 1)], dtype=capi.GENE)
orc = oracle_lib.load()
ev = orc.call(rows, genes, refseq=ref)
ep = orc.phase(rows, ev)
jl = capi.Juliet(0)
jl.upload_rows(rows)
for rep in range(3):
    out = jl.run(genes, ref)
    ph = out["phase"]
    print(rep, ph["summary"], ep["summary"], "ids equal:", int((ph["read_hap"] == ep["read_hap"]).sum()), "of", n, flush=True)
    bad = np.nonzero(ph["read_hap"] != ep["read_hap"])[0]
    print("   first bad", bad[:10], ph["read_hap"][bad[:10]], ep["read_hap"][bad[:10]])
