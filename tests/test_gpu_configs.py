"""BASELINE.json configs[3] and configs[4] at their real sizes against the oracle (one MI355X holds either):

  configs[3]  1M CCS reads x 10 kb reference, window-sharded: ALL of it on one GPU as the eight column windows the
              8-GPU run gives one rank each (1M x 1252 columns = the per-GPU share), every window through jl_run_async,
              tables merged, the variant columns of all windows assembled (jl_xwin_assemble_local), phasing across windows;
  configs[4]  the per-GPU share of 10M CCS reads x 9719-column full-HIV reference: 10M x 1217 columns (6.08 GB), nine
              overlapping ORFs in three frames.

The oracle runs at FULL size too — by column chunks (calls are independent per column given the global Bonferroni
factor, docs/SPEC.md §9), so it never needs more than one chunk by-row in host memory.  Integer outputs bit-exact,
p-values within 1e-10, plus the size-independent properties (column sums, coverage bounds, category arithmetic)."""
import os

import numpy as np
import pytest

import oracle_lib
from minorseq_amd import capi, msa, sharding, synth
from test_gpu_parity import assert_phase_equal, assert_variants_equal, oracle_params

pytestmark = pytest.mark.gpu

THREADS = min(16, os.cpu_count() or 1)


def oracle_window(oracle, packed, n, genes, win_begin, ref, prm, chunk_cols=160):
    """orc_pileup + orc_call over a column-packed window, chunk by chunk (chunks overlap by two columns so that every
    codon lies inside exactly one).  Returns (col_counts[n_cols][6], variant rows with window-relative columns)."""
    l = packed.shape[0]
    k = max(1, (l + chunk_cols - 1) // chunk_cols)
    counts = np.zeros((l, 6), dtype=np.uint32)
    tables = []
    for b, e in sharding.window_bounds(l, k):
        rows = msa.unpack_columns(packed[b:e], n)
        counts[b:e] = oracle.pileup(rows)
        v = oracle.call(rows, genes, win_begin=win_begin + b, refseq=ref, params=prm)
        v["col"] += np.uint32(b)
        tables.append(v)
        del rows
    allv = np.concatenate(tables)
    return counts, allv[np.lexsort((allv["codon"], allv["codon_pos"], allv["gene"]))]


def variant_columns_rows(packed, n, cols):
    """By-row matrix of only the codon columns `cols` (window-relative first columns): what orc_phase needs."""
    idx = np.concatenate([np.arange(c, c + 3) for c in cols])
    return msa.unpack_columns(packed[idx], n)


def compact_table(v, cols):
    """The variant rows with `col` renumbered into the compact matrix of variant_columns_rows."""
    out = v.copy()
    order = {int(c): 3 * k for k, c in enumerate(cols)}
    out["col"] = [order[int(c)] for c in v["col"]]
    return out


def check_pileup_properties(pf, n, full_span):
    if full_span:
        assert (pf["col_counts"].sum(axis=1) == n).all()              # one symbol per read and column
    assert (pf["hist"].sum(axis=1) == pf["coverage"]).all()
    depth = pf["col_counts"][:, :4].sum(axis=1)
    assert (pf["coverage"] <= np.minimum.reduce([depth[pf["pos_col"] + k] for k in range(3)])).all()


def test_config3_whole_reference_as_eight_windows_on_one_gpu(oracle):
    n, l, world = 1_000_000, 10_000, 8
    sp = synth.SynthParams(seed=4)
    ref = synth.reference(sp.seed, l)
    genes = np.array([(1, 3 * (l // 3) + 1)], dtype=capi.GENE)
    prm = capi.default_params(n_tests=sharding.default_n_tests(genes))
    oprm = oracle_params(prm)
    wb = sharding.window_bounds(l, world)
    assert max(e - b for b, e in wb) == 1252                           # the per-GPU share of configs[3]
    oracle.set_threads(THREADS)
    try:
        ctxs, tables, exp_tables = [], [], []
        for b, e in wb:
            c = capi.Juliet(0)
            c.alloc(n, e - b, win_begin=b)
            c.synth_fill_window(sp, ref)
            c.run_async(genes, ref, prm, None, False, 10, False)       # call only: phasing follows across windows
            got = c.run_fetch(False, False)["variants"].copy()
            pf = c.pileup_fetch()
            check_pileup_properties(pf, n, True)
            packed = c.download_columns()
            counts, ev = oracle_window(oracle, packed, n, genes, b, ref, oprm)
            assert (pf["col_counts"] == counts).all()
            assert_variants_equal(got, ev)
            ctxs.append(c)
            tables.append(got)
            exp_tables.append(ev)
            del packed
        merged = sharding.merge_tables(tables, [b for b, _ in wb])
        assert len(merged) == 5 and len({int(c) // 1250 for c in merged["col"]}) >= 2    # variants in several windows
        # phasing across windows on the device vs the oracle on the same variant columns of the same reads
        ph, pos_global = capi.phase_across_windows(ctxs, merged)
        cols_rows = []
        for c0 in pos_global:
            w = next(k for k, (b, e) in enumerate(wb) if b <= c0 and c0 + 3 <= e)
            cols_rows.append((w, int(c0) - wb[w][0]))
        packed_cols = np.concatenate([ctxs[w].download_columns()[c: c + 3] for w, c in cols_rows])
        rows_v = msa.unpack_columns(packed_cols, n)
        ev = merged.copy()
        ev["col"] = [3 * int(np.searchsorted(pos_global, c)) for c in merged["col"]]
        exp = oracle.phase(rows_v, ev)
        exp["pos_cols"] = pos_global
        assert_phase_equal(dict(ph, pos_cols=pos_global), exp, len(merged))
        s = ph["summary"]
        assert s["reported_reads"] + s["insufficient_reads"] + s["damaged_reads"] == n
        # at 1e6 reads single substitutions inside a variant codon reach 10 reads too: dozens of haplotypes, the first
        # five are the planted ones (wild type, then the four 1 % minors; Y181C + G190A travel together, A.3)
        assert s["n_haplotypes"] >= 5 and ph["hit"][:, 0].sum() == 0
        assert sorted(int(ph["hit"][:, h].sum()) for h in range(1, 5)) == [1, 1, 1, 2]
        assert (ph["hap_count"][1:5] > 0.005 * n).all() and (ph["hap_count"][5:] < 0.001 * n).all()
        # the per-GPU share with phasing on, through one jl_run_async (rank 1 holds three of the five variants)
        out = ctxs[1].run(genes, ref, prm, phasing=True)
        assert_variants_equal(out["variants"], exp_tables[1])
        rows_v1 = variant_columns_rows(ctxs[1].download_columns(), n, np.unique(exp_tables[1]["col"]))
        e1 = oracle.phase(rows_v1, compact_table(exp_tables[1], np.unique(exp_tables[1]["col"])))
        e1["pos_cols"] = np.unique(exp_tables[1]["col"])
        assert_phase_equal(out["phase"], e1, len(exp_tables[1]))
        for c in ctxs:
            c.close()
    finally:
        oracle.set_threads(1)


def test_config3_sharded_equals_unsharded_oracle_on_a_reduced_read_copy(oracle):
    """The same eight-window pipeline on the first 20 000 reads of configs[3] against the oracle run UNSHARDED over
    the whole 20 000 x 10 000 matrix: windowing, global Bonferroni factor, merge and cross-window phasing change nothing."""
    n, l, world = 20_000, 10_000, 8
    sp = synth.SynthParams(seed=4, minor_permille=(30, 25, 20, 15))     # minors a 20k-read sample can carry
    ref = synth.reference(sp.seed, l)
    rows = synth.rows(sp, l, 0, n, ref)
    genes = np.array([(1, 3 * (l // 3) + 1)], dtype=capi.GENE)
    prm = capi.default_params(n_tests=sharding.default_n_tests(genes))
    full_v = oracle.call(rows, genes, refseq=ref, params=oracle_params(prm))
    exp = oracle.phase(rows, full_v)
    wb = sharding.window_bounds(l, world)
    ctxs, tables = [], []
    for b, e in wb:
        c = capi.Juliet(0)
        c.alloc(n, e - b, win_begin=b)
        c.synth_fill_window(sp, ref)
        assert (msa.unpack_columns(c.download_columns(), n) == rows[:, b:e]).all()    # windows hold the same reads
        c.run_async(genes, ref, prm, None, False, 10, False)
        tables.append(c.run_fetch(False, False)["variants"].copy())
        ctxs.append(c)
    merged = sharding.merge_tables(tables, [b for b, _ in wb])
    assert_variants_equal(merged, full_v)
    ph, pos_global = capi.phase_across_windows(ctxs, merged)
    assert (pos_global == exp["pos_cols"]).all()
    assert_phase_equal(dict(ph, pos_cols=pos_global), exp, len(full_v))
    for c in ctxs:
        c.close()


HIV_GENES = [(1, 634), (790, 1186), (1186, 1879), (1879, 1921), (1921, 2086), (2086, 2134), (2134, 2292), (2253, 2550),
             (2550, 4230), (4230, 5096), (5041, 5619), (5559, 5850), (6062, 6310), (6225, 8795), (8797, 9417)]


def test_config4_per_gpu_share_ten_million_reads(oracle):
    """10M reads x 1217 columns = rank 1's window of the 9719-column reference split eight ways (configs[4]); genes as
    on juliet_target.png (5'LTR ... Protease, p6 / Protease overlapping in different frames) continued with HIV-like
    ORFs in all three frames.  Pileup with the reads of a column split over several blocks, Fisher at 1e7 coverage
    (p underflows, log-p carried), phasing over 10M reads."""
    n, l, world, rank = 10_000_000, 9719, 8, 1
    sp = synth.SynthParams(seed=5)
    ref = synth.reference(sp.seed, l)
    genes = np.array(HIV_GENES, dtype=capi.GENE)
    prm = capi.default_params(n_tests=sharding.default_n_tests(genes))
    b, e = sharding.window_bounds(l, world)[rank]
    assert e - b == 1217
    c = capi.Juliet(0)
    c.alloc(n, e - b, win_begin=b)
    c.synth_fill_window(sp, ref)
    out = c.run(genes, ref, prm, phasing=True)
    pf = c.pileup_fetch()
    check_pileup_properties(pf, n, True)
    packed = c.download_columns()
    c.close()
    oracle.set_threads(THREADS)
    try:
        counts, ev = oracle_window(oracle, packed, n, genes, b, ref, oracle_params(prm), chunk_cols=100)
        assert (pf["col_counts"] == counts).all()
        assert len(ev) >= 3 and len(np.unique(ev["gene"])) >= 2      # the planted edits are seen by overlapping ORFs
        assert_variants_equal(out["variants"], ev)
        assert (ev["p_value"] == 0.0).any() and np.isfinite(ev["log_p"]).all()
        cols = np.unique(ev["col"])
        rows_v = variant_columns_rows(packed, n, cols)
        del packed
        exp = oracle.phase(rows_v, compact_table(ev, cols))
        exp["pos_cols"] = cols
        assert_phase_equal(out["phase"], exp, len(ev))
        s = out["phase"]["summary"]
        assert s["reported_reads"] + s["insufficient_reads"] + s["damaged_reads"] == n
    finally:
        oracle.set_threads(1)


def test_config4_whole_reference_eight_windows_resident_on_one_gpu(oracle):
    """configs[4] WHOLE on one MI355X: 10M CCS reads x the 9719-column full-HIV reference (48.6 GB of 288) as the eight
    column windows an 8-GPU run gives one rank each, all resident together; fifteen ORFs in three frames, several of
    them overlapping across window borders.  Checked at full size:
      * every window's column counts and histograms by their size-independent properties (one symbol per read and column,
        histogram sums = coverage <= column depth);
      * calls and column counts against the oracle over ALL 9719 columns of the full 10M reads, by column chunks (calls are
        independent per column given the global Bonferroni factor), bit-exact;
      * phasing across the eight windows — once through the session (jl_xwin_phase_sharded) and once with the reads cut into
        eight slices as eight ranks would hold them — against the oracle on the variant columns of all 10M reads;
      * the read categories add up to 10M (doc/JULIET.md:378-379)."""
    n, l, world = 10_000_000, 9719, 8
    sp = synth.SynthParams(seed=5)
    ref = synth.reference(sp.seed, l)
    genes = np.array(HIV_GENES, dtype=capi.GENE)
    prm = capi.default_params(n_tests=sharding.default_n_tests(genes))
    oprm = oracle_params(prm)
    wb = sharding.window_bounds(l, world)
    ctxs, tables = [], []
    for b, e in wb:                                              # all eight windows resident before anything runs
        c = capi.Juliet(0)
        c.alloc(n, e - b, win_begin=b)
        c.synth_fill_window(sp, ref)
        ctxs.append(c)
    for c in ctxs:
        c.run_async(genes, ref, prm, None, False, 10, False)     # call only, every window with the GLOBAL Bonferroni factor
    oracle.set_threads(THREADS)
    try:
        var_cols = {}                                            # global column -> packed [3][stride] of the variant codon
        chunk = 100
        checked_chunks = total_chunks = 0
        for (b, e), c in zip(wb, ctxs):
            got = c.run_fetch(False, False)["variants"].copy()
            tables.append(got)
            pf = c.pileup_fetch()
            check_pileup_properties(pf, n, True)
            packed = c.download_columns()
            lw = e - b
            k = max(1, (lw + chunk - 1) // chunk)
            bounds = sharding.window_bounds(lw, k)
            total_chunks += k
            for i, (cb, ce) in enumerate(bounds):
                checked_chunks += 1
                rows = msa.unpack_columns(packed[cb:ce], n)
                assert (pf["col_counts"][cb:ce] == oracle.pileup(rows)).all()
                ev = oracle.call(rows, genes, win_begin=b + cb, refseq=ref, params=oprm)
                ev["col"] += np.uint32(cb)
                # the device's rows whose codon lies inside this chunk (chunks overlap by two columns: a codon is in one)
                own_hi = ce - 2 if i + 1 < k else ce
                mine = got[(got["col"] >= cb) & (got["col"] < own_hi)]
                ev = ev[(ev["col"] >= cb) & (ev["col"] < own_hi)]
                assert_variants_equal(mine, ev[np.lexsort((ev["codon"], ev["codon_pos"], ev["gene"]))])
                del rows
            for col in np.unique(got["col"]):
                var_cols[b + int(col)] = packed[int(col): int(col) + 3].copy()
            del packed
        assert checked_chunks == total_chunks >= 96
        merged = sharding.merge_tables(tables, [b for b, _ in wb])
        assert len(merged) >= 5 and len(np.unique(merged["gene"])) >= 3 and len({int(c) * world // l for c in merged["col"]}) >= 2   # the planted edits span two windows and three ORFs
        assert (merged["p_value"] == 0.0).any() and np.isfinite(merged["log_p"]).all()      # p underflows at 1e7 reads, log-p is carried
        # the oracle's phasing on the variant columns of all 10M reads
        pos = np.array(sorted(var_cols), dtype=np.uint32)
        assert (pos == np.unique(merged["col"])).all()
        rows_v = msa.unpack_columns(np.concatenate([var_cols[int(c)] for c in pos]), n)
        ev = merged.copy()
        ev["col"] = [3 * int(np.searchsorted(pos, c)) for c in merged["col"]]
        exp = oracle.phase(rows_v, ev)
        exp["pos_cols"] = pos
        del rows_v
        # (a) the session: one call of the C ABI
        xw = capi.Xwin(ctxs, [b for b, _ in wb], [e - b for b, e in wb], [0] * world, [0, n])
        res = xw.phase(10)
        xw.close()
        assert_variants_equal(res["merged"], merged)
        assert_phase_equal(res, exp, len(merged))
        s = res["summary"]
        assert s["reported_reads"] + s["insufficient_reads"] + s["damaged_reads"] == n
        assert s["n_haplotypes"] >= 5 and res["hit"][:, 0].sum() == 0
        # (b) the reads in eight slices, as eight ranks would hold them
        ph, pos_global = capi.phase_sharded_by_reads(ctxs, merged, world)
        assert (pos_global == pos).all()
        assert_phase_equal(dict(ph, pos_cols=pos_global), exp, len(merged))
    finally:
        oracle.set_threads(1)
        for c in ctxs:
            c.close()
