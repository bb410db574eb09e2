"""What a window with MORE than ten variant positions costs: the single-word phasing launch flags it, the fetch re-runs the
multi-word pipeline (keys, grouping, selection, ids: four launches); from then on the context takes that pipeline at once."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from minorseq_amd import capi, msa, synth  # noqa: E402

n, l = 100_000, 3000
genes = np.array([(1, l + 1)], dtype=capi.GENE)
prm = capi.default_params()
for extra in (0, 12):
    sp = synth.SynthParams(seed=2)
    ref = synth.reference(sp.seed, l)
    rows = synth.rows(sp, l, 0, n, ref)
    rng = np.random.default_rng(3)
    for k in range(extra):
        who = rng.choice(n, n // 30, replace=False)
        c0 = 3 * (50 + 70 * k)
        rows[who, c0:c0 + 3] = (rows[who, c0:c0 + 3] + 1 + k % 3) % 4
    c = capi.Juliet(0)
    c.upload_columns(msa.pack_columns(rows), n)
    t0 = time.perf_counter()
    c.run_async(genes, ref, prm, None, True, 10, True)
    out = c.run_view() or c.run_fetch(True, True, cap_var=64)
    first = time.perf_counter() - t0
    for _ in range(3):
        c.run_async(genes, ref, prm, None, True, 10, True)
        out = c.run_view() or c.run_fetch(True, True, cap_var=64)
    R = 50
    t0 = time.perf_counter()
    for _ in range(R):
        c.run_async(genes, ref, prm, None, True, 10, True)
        out = c.run_view() or c.run_fetch(True, True, cap_var=64)
    dt = (time.perf_counter() - t0) / R
    ph = out["phase"]
    print(f"{len(out['variants'])} variants at {ph['summary']['n_positions']} positions, {ph['summary']['n_haplotypes']} haplotypes: "
          f"first run {first * 1e3:.2f} ms, then {dt * 1e6:.1f} us per run", flush=True)
    c.close()
