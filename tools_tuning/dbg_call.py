import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import oracle_lib
from minorseq_amd import capi, msa, synth
from test_gpu_parity import oracle_params
orc = oracle_lib.load()
jl = capi.Juliet(0)
n, l, use_ref = 3000, 300, True
sp = synth.SynthParams(seed=n + 7 * l, minor_permille=(40, 30, 20, 15), partial_rate=0.1)
ref = synth.reference(sp.seed, l)
rows = synth.rows(sp, l, 0, n, ref)
jl.upload_columns(msa.pack_columns(rows), n)
genes = np.array([(1, l + 1), (2, l - 1)], dtype=capi.GENE)
for i, prm in enumerate((capi.default_params(), capi.default_params(alpha=0.5, n_tests=1.0),
            capi.default_params(chemistry="permissive", expected_round=1),
            capi.default_params(alpha=0.5, n_tests=1.0, expected_round=2))):
    jl.pileup_async(genes, ref)
    jl.call_async(prm)
    got = jl.call_fetch()
    exp = orc.call(rows, genes, refseq=ref, params=oracle_params(prm))
    gs = {(int(r['gene']), int(r['codon_pos']), int(r['codon'])): r for r in got}
    es = {(int(r['gene']), int(r['codon_pos']), int(r['codon'])): r for r in exp}
    print(i, len(got), len(exp))
    for k in sorted(set(gs) ^ set(es)):
        print('  only in', 'gpu' if k in gs else 'oracle', (gs.get(k) if k in gs else es.get(k)))
        r = es.get(k) if k in es else gs.get(k)
        p, lp = jl.fisher_eval([r['count']], [r['expected']], [r['coverage']])
        print('   device fisher', p, lp, 'oracle', orc.fisher(int(r['count']), int(r['coverage']-r['count']), int(r['expected']), int(r['coverage']-r['expected'])))
