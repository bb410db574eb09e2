"""GPU parity: the HIP path through the C ABI vs the CPU oracle on the same inputs.

Integer outputs (column counts, codon histograms, coverage, variant rows, haplotypes, read assignments,
co-occurrence) must be bit-exact; p-values within 1e-10 absolute (BASELINE.json north_star), log-p within
1e-9 relative.  Runs only on a real MI355X: `pytest -m gpu`.
"""
import json
import os

import numpy as np
import pytest

import oracle_lib
from minorseq_amd import capi, msa, synth

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
P_ABS_TOL = 1e-10     # north_star: "p-values within 1e-10"
LOGP_REL_TOL = 1e-9


@pytest.fixture(scope="module")
def jl():
    j = capi.Juliet(0)
    yield j
    j.close()


def oracle_params(prm: capi.Params) -> oracle_lib.Params:
    return oracle_lib.Params(prm.alpha, prm.n_tests,
                             oracle_lib.ErrorModel(prm.err.match, prm.err.substitution, prm.err.deletion),
                             prm.expected_round, prm.tail)


def assert_variants_equal(got, exp):
    assert len(got) == len(exp), (len(got), len(exp))
    for k in ("gene", "codon_pos", "col", "ref_codon", "codon", "count", "coverage", "expected"):
        assert (got[k] == exp[k]).all(), k
    assert np.abs(got["p_value"] - exp["p_value"]).max(initial=0.0) <= P_ABS_TOL
    fin = np.isfinite(exp["log_p"])
    assert (np.isfinite(got["log_p"]) == fin).all()
    if fin.any():
        rel = np.abs(got["log_p"][fin] - exp["log_p"][fin]) / np.maximum(1.0, np.abs(exp["log_p"][fin]))
        assert rel.max() <= LOGP_REL_TOL


def assert_phase_equal(got, exp, n_var):
    assert got["summary"] == exp["summary"]
    h = exp["summary"]["n_haplotypes"]
    assert (got["pos_cols"] == exp["pos_cols"]).all()
    assert (got["hap_count"] == exp["hap_count"]).all()
    assert (got["hap_pattern"] == exp["hap_pattern"]).all()
    assert (got["hit"][:n_var, :h] == exp["hit"]).all()
    assert (got["read_hap"] == exp["read_hap"]).all()
    if got["cooc"] is not None and n_var <= 256:
        assert (got["cooc"][:n_var, :n_var] == exp["cooc"]).all()


# --------------------------------------------------------------------------------------------- layout
@pytest.mark.parametrize("n,l", [(1, 3), (7, 10), (255, 33), (256, 36), (257, 5), (1000, 100), (8193, 41)])
def test_pack_rows_on_device_matches_numpy(jl, n, l):
    rows = np.random.default_rng(n * 131 + l).integers(0, 7, size=(n, l), dtype=np.uint8)
    jl.upload_rows(rows)
    assert (jl.download_columns() == msa.pack_columns(rows)).all()


@pytest.mark.parametrize("n,l,partial", [(1000, 300, 0.0), (4100, 90, 0.3), (513, 3000, 0.1)])
def test_synth_fill_matches_numpy_mirror(jl, n, l, partial):
    sp = synth.SynthParams(seed=n + l, partial_rate=partial)
    ref = synth.reference(sp.seed, l)
    jl.alloc(n, l)
    jl.synth_fill(sp, ref)
    dev = msa.unpack_columns(jl.download_columns(), n)
    assert (dev == synth.rows(sp, l, 0, n, ref)).all()
    # padding reads are uncovered
    full = msa.unpack_columns(jl.download_columns(), jl.col_stride * 2)
    assert (full[n:] == msa.SYM_NONE).all()


# --------------------------------------------------------------------------------------------- numerics
def test_fisher_device_vs_golden(jl):
    with open(os.path.join(HERE, "golden", "fisher_golden.json")) as f:
        tables = [r for r in json.load(f)["tables"] if r["a"] + r["b"] == r["c"] + r["d"]]
    a = np.array([r["a"] for r in tables], dtype=np.uint32)
    c = np.array([r["c"] for r in tables], dtype=np.uint32)
    cov = np.array([r["a"] + r["b"] for r in tables], dtype=np.uint32)
    p, lp = jl.fisher_eval(a, c, cov)
    gp = np.array([float(r["p"]) for r in tables])
    glp = np.array([float(r["log_p"]) for r in tables])
    assert np.abs(p - gp).max() <= P_ABS_TOL
    big = gp > 1e-300
    assert (np.abs(p[big] - gp[big]) / gp[big]).max() <= 1e-11
    assert (np.abs(lp - glp) <= 1e-12 * np.maximum(1.0, np.abs(glp)) + 1e-13).all()


def test_fisher_device_vs_oracle_random(jl, oracle):
    rng = np.random.default_rng(9)
    cov = rng.integers(1, 3_000_000, size=4000).astype(np.uint32)
    a = np.minimum(cov, rng.integers(1, 500, size=4000)).astype(np.uint32)
    c = np.minimum(cov, rng.integers(0, 60, size=4000)).astype(np.uint32)
    p, lp = jl.fisher_eval(a, c, cov)
    for i in range(len(a)):
        op, olp = oracle.fisher(int(a[i]), int(cov[i] - a[i]), int(c[i]), int(cov[i] - c[i]))
        assert abs(p[i] - op) <= P_ABS_TOL
        assert abs(lp[i] - olp) <= LOGP_REL_TOL * max(1.0, abs(olp))


# --------------------------------------------------------------------------------------------- pileup
SHAPES = [  # n_reads, n_cols, partial_rate  — ragged sizes around the 8192-read tile and the 12-column chunk
    (1, 3, 0.0), (2, 4, 0.0), (63, 12, 0.0), (64, 13, 0.5), (300, 11, 0.2), (1000, 36, 0.0), (1001, 37, 0.3),
    (8192, 24, 0.0), (8193, 25, 0.1), (20000, 50, 0.2), (33000, 14, 0.0),
]


@pytest.mark.parametrize("n,l,partial", SHAPES)
def test_pileup_and_histograms_bit_exact(jl, oracle, n, l, partial):
    sp = synth.SynthParams(seed=3 * n + l, partial_rate=partial, mask_rate=0.05, del_rate=0.03, sub_rate=0.02,
                           minor_permille=(100, 50, 30, 20))
    ref = synth.reference(sp.seed, l)
    rows = synth.rows(sp, l, 0, n, ref)
    jl.upload_columns(msa.pack_columns(rows), n)
    # three overlapping genes, all three frames, one running past the window end
    genes = np.array([(1, l + 1), (2, l + 1), (3, l + 7)], dtype=capi.GENE)
    for refseq in (None, ref):
        jl.pileup_async(genes, refseq)
        got = jl.pileup_fetch()
        assert (got["col_counts"] == oracle.pileup(rows)).all()
        hist, cov = oracle.codon_hist(rows, got["pos_col"])
        assert (got["hist"] == hist).all()
        assert (got["coverage"] == cov).all()
        assert (got["col_counts"].sum(axis=1) == (rows != msa.SYM_NONE).sum(axis=0)).all()


def test_pileup_empty_and_degenerate(jl, oracle):
    # all-uncovered matrix, all-N matrix, single column
    for fill in (msa.SYM_NONE, msa.SYM_MASK, msa.SYM_GAP):
        rows = np.full((500, 9), fill, dtype=np.uint8)
        jl.upload_columns(msa.pack_columns(rows), 500)
        jl.pileup_async(np.array([(1, 10)], dtype=capi.GENE))
        got = jl.pileup_fetch()
        assert (got["col_counts"] == oracle.pileup(rows)).all()
        assert got["hist"].sum() == 0 and (got["coverage"] == 0).all()
        jl.call_async(capi.default_params())
        assert len(jl.call_fetch()) == 0
    rows = np.zeros((10, 1), dtype=np.uint8)
    jl.upload_columns(msa.pack_columns(rows), 10)
    jl.pileup_async(np.array([(1, 2)], dtype=capi.GENE))
    got = jl.pileup_fetch()
    assert len(got["pos_col"]) == 0 and got["col_counts"][0, 0] == 10
    with pytest.raises(capi.JulietError):
        jl.upload_columns(np.zeros((0, 128), dtype=np.uint8), 10)
    # symbol codes outside 0..6 are rejected at upload (SPEC §1)
    for bad in (0x07, 0x70, 0x08, 0xF0):
        packed = msa.pack_columns(np.zeros((300, 4), dtype=np.uint8))
        packed[2, 77] = bad
        with pytest.raises(capi.JulietError) as e:
            jl.upload_columns(packed, 300)
        assert e.value.status == -1 and "0..6" in str(e.value)
    with pytest.raises(ValueError):
        msa.pack_columns(np.full((3, 3), 7, dtype=np.uint8))


def test_consensus_of_pileup(jl, oracle):
    rows = synth.rows(synth.SynthParams(seed=61, del_rate=0.02, partial_rate=0.3), 200, 0, 3000)
    rows[:, 17] = msa.SYM_GAP          # a column whose majority is a deletion
    rows[:, 50] = msa.SYM_NONE         # a column nobody covers
    rows[:, 51] = msa.SYM_MASK         # only filtered bases: no A C G T - at all
    jl.upload_columns(msa.pack_columns(rows), 3000)
    jl.pileup_async(np.array([(1, 201)], dtype=capi.GENE))
    got = jl.consensus()
    col = oracle.pileup(rows)[:, :5]
    exp = np.where(col.max(axis=1) == 0, 5, col.argmax(axis=1)).astype(np.uint8)
    assert (got == exp).all() and got[17] == 4 and got[50] == 5 and got[51] == 5


def test_seed_never_changes_results(jl, oracle):
    """A wrong reference (bad seed for the codon fast path) must give the same histograms."""
    sp = synth.SynthParams(seed=77, minor_permille=(200, 100, 100, 100))
    ref = synth.reference(sp.seed, 60)
    rows = synth.rows(sp, 60, 0, 5000, ref)
    jl.upload_columns(msa.pack_columns(rows), 5000)
    genes = np.array([(1, 61)], dtype=capi.GENE)
    wrong = ((ref.astype(np.int64) + 1) % 4).astype(np.uint8)
    jl.pileup_async(genes, wrong)
    got = jl.pileup_fetch()
    hist, cov = oracle.codon_hist(rows, got["pos_col"])
    assert (got["hist"] == hist).all() and (got["coverage"] == cov).all()


# --------------------------------------------------------------------------------------------- call
@pytest.mark.parametrize("n,l,use_ref", [(3000, 300, True), (3000, 300, False), (12000, 99, True), (700, 48, False)])
def test_call_matches_oracle(jl, oracle, n, l, use_ref):
    sp = synth.SynthParams(seed=n + 7 * l, minor_permille=(40, 30, 20, 15), partial_rate=0.1)
    ref = synth.reference(sp.seed, l)
    rows = synth.rows(sp, l, 0, n, ref)
    jl.upload_columns(msa.pack_columns(rows), n)
    genes = np.array([(1, l + 1), (2, l - 1)], dtype=capi.GENE)
    for prm in (capi.default_params(), capi.default_params(alpha=0.3, n_tests=1.0),
                capi.default_params(chemistry="permissive", expected_round=1),
                capi.default_params(alpha=0.3, n_tests=1.0, expected_round=2)):
        jl.pileup_async(genes, ref if use_ref else None)
        jl.call_async(prm)
        got = jl.call_fetch()
        exp = oracle.call(rows, genes, refseq=ref if use_ref else None, params=oracle_params(prm))
        assert_variants_equal(got, exp)


def test_call_window_offset_and_nonacgt_reference(jl, oracle):
    sp = synth.SynthParams(seed=5, minor_permille=(60, 60, 60, 60))
    L = 400
    ref = synth.reference(sp.seed, L)
    rows = synth.rows(sp, L, 0, 2500, ref)
    ref_n = ref.copy()
    ref_n[[30, 31, 200]] = 4          # non-ACGT reference bases: those codons are skipped (SPEC §4)
    genes = np.array([(10, 390), (101, 161)], dtype=capi.GENE)
    win = slice(9, 330)
    jl.upload_columns(msa.pack_columns(rows[:, win]), 2500, win_begin=9)
    jl.pileup_async(genes, ref_n)
    jl.call_async(capi.default_params())
    got = jl.call_fetch()
    exp = oracle.call(rows[:, win], genes, win_begin=9, refseq=ref_n)
    assert len(exp) >= 3
    assert_variants_equal(got, exp)


def test_filters_min_max_perc_and_drm(jl, oracle):
    sp = synth.SynthParams(seed=21, minor_permille=(300, 60, 30, 10))
    ref = synth.reference(sp.seed, 3000)
    rows = synth.rows(sp, 3000, 0, 4000, ref)
    jl.upload_columns(msa.pack_columns(rows), 4000)
    genes = np.array([(1, 3001)], dtype=capi.GENE)
    base = oracle.call(rows, genes, refseq=ref)
    perc = 100.0 * base["count"] / base["coverage"]
    assert len(base) == 5
    for lo, hi in ((-1.0, -1.0), (2.0, -1.0), (-1.0, 5.0), (2.0, 20.0)):
        jl.pileup_async(genes, ref)
        jl.call_async(capi.default_params(min_perc=lo, max_perc=hi))
        keep = np.ones(len(base), dtype=bool)
        if lo >= 0:
            keep &= perc > lo          # doc/JULIET.md:342-344
        if hi >= 0:
            keep &= perc < hi          # doc/JULIET.md:352-354
        assert_variants_equal(jl.call_fetch(), base[keep])
    # --drm-only (doc/JULIET.md:370): per-position codon masks
    jl.pileup_async(genes, ref)
    masks = np.zeros(1000, dtype=np.uint64)
    for r in base[:2]:
        masks[r["codon_pos"] - 1] |= np.uint64(1) << np.uint64(r["codon"])
    jl.call_async(capi.default_params(), drm_masks=masks)
    assert_variants_equal(jl.call_fetch(), base[:2])


# --------------------------------------------------------------------------------------------- phase
@pytest.mark.parametrize("n,l,partial", [(1000, 3000, 0.0), (5000, 300, 0.2), (9000, 120, 0.0), (64, 30, 0.0)])
def test_phase_matches_oracle(jl, oracle, n, l, partial):
    sp = synth.SynthParams(seed=n + l, minor_permille=(80, 60, 50, 40), partial_rate=partial)
    ref = synth.reference(sp.seed, l)
    rows = synth.rows(sp, l, 0, n, ref)
    jl.upload_columns(msa.pack_columns(rows), n)
    genes = np.array([(1, l + 1)], dtype=capi.GENE)
    out = jl.run(genes, ref, capi.default_params(), phasing=True, min_reads=10)
    exp_v = oracle.call(rows, genes, refseq=ref)
    assert_variants_equal(out["variants"], exp_v)
    exp = oracle.phase(rows, exp_v, min_reads=10)
    assert_phase_equal(out["phase"], exp, len(exp_v))
    s = out["phase"]["summary"]
    if s["n_positions"]:   # SPEC §8: no variant positions => no phasing, all counters 0
        assert s["reported_reads"] + s["insufficient_reads"] + s["damaged_reads"] == n   # doc/JULIET.md:378-379


def test_phase_with_host_table_many_positions(jl, oracle):
    """Host-supplied table (the all-gather case), > 40 positions so the key spans several words and the
    resident key buffer has to grow (re-run protocol)."""
    n, l = 6000, 600
    sp = synth.SynthParams(seed=31, sub_rate=0.01, minor_permille=(50, 50, 50, 50))
    ref = synth.reference(sp.seed, l)
    rows = synth.rows(sp, l, 0, n, ref)
    jl.upload_columns(msa.pack_columns(rows), n)
    genes = np.array([(1, l + 1)], dtype=capi.GENE)
    loose = capi.default_params(alpha=0.9, n_tests=1.0)
    jl.pileup_async(genes, None)
    jl.call_async(loose)
    table = jl.call_fetch()
    assert len(np.unique(table["col"])) > 45
    exp_v = oracle.call(rows, genes, params=oracle_params(loose))
    assert_variants_equal(table, exp_v)
    for sub in (table, table[::3], table[:11], table[:0]):
        jl.phase_async(sub, min_reads=3)
        got = jl.phase_fetch(cap_var=max(1, len(sub)))
        exp = oracle.phase(rows, sub, min_reads=3)
        assert_phase_equal(got, exp, len(sub))
    # resident-table path with many positions: grows the key buffer transparently
    j2 = capi.Juliet(0)
    j2.upload_columns(msa.pack_columns(rows), n)
    j2.pileup_async(genes, None)
    j2.call_async(loose)
    j2.phase_async(None, min_reads=3)
    got = j2.phase_fetch(cap_var=len(table))
    assert_phase_equal(got, oracle.phase(rows, table, min_reads=3), len(table))
    j2.close()


def test_phase_all_damaged_and_threshold(jl, oracle):
    # every read has a deletion in the variant codon: no haplotypes (doc/JULIET.md:285-288)
    rows = np.zeros((300, 6), dtype=np.uint8)
    rows[:, 1] = msa.SYM_GAP
    var = np.zeros(1, dtype=capi.VARIANT)
    var["col"], var["codon"] = 0, 5
    jl.upload_columns(msa.pack_columns(rows), 300)
    jl.pileup_async(np.array([(1, 7)], dtype=capi.GENE))
    jl.phase_async(var)
    got = jl.phase_fetch(cap_var=1)
    assert_phase_equal(got, oracle.phase(rows, var), 1)
    assert got["summary"]["n_haplotypes"] == 0 and got["summary"]["damaged_reads"] == 300
    # >= 10 reads to report (doc/JULIET.md:253-254)
    rows = np.zeros((200, 3), dtype=np.uint8)
    rows[:10, 0] = 1
    rows[10:19, 0] = 2
    var = np.zeros(2, dtype=capi.VARIANT)
    var["codon"] = [msa.codon_index("CAA"), msa.codon_index("GAA")]
    jl.upload_columns(msa.pack_columns(rows), 200)
    jl.pileup_async(np.array([(1, 4)], dtype=capi.GENE))
    jl.phase_async(var, min_reads=10)
    got = jl.phase_fetch(cap_var=2)
    assert_phase_equal(got, oracle.phase(rows, var, min_reads=10), 2)
    assert got["hap_count"].tolist() == [181, 10] and got["summary"]["insufficient_reads"] == 9


def test_permutation_invariance_on_device(jl):
    sp = synth.SynthParams(seed=13, minor_permille=(70, 60, 50, 40))
    ref = synth.reference(sp.seed, 300)
    rows = synth.rows(sp, 300, 0, 7000, ref)
    perm = np.random.default_rng(1).permutation(len(rows))
    genes = np.array([(1, 301)], dtype=capi.GENE)
    jl.upload_columns(msa.pack_columns(rows), len(rows))
    a = jl.run(genes, ref)
    jl.upload_columns(msa.pack_columns(rows[perm]), len(rows))
    b = jl.run(genes, ref)
    assert (a["variants"] == b["variants"]).all()
    assert a["phase"]["summary"] == b["phase"]["summary"]
    assert (a["phase"]["hap_pattern"] == b["phase"]["hap_pattern"]).all()
    assert (a["phase"]["read_hap"][perm] == b["phase"]["read_hap"]).all()


# --------------------------------------------------------------------------------------------- full size
def test_config2_full_size_against_oracle(jl, oracle):
    """BASELINE.json configs[1]/[2]: 100k CCS reads x 3 kb, call + phase, device-generated reads."""
    n, l = 100_000, 3000
    sp = synth.SynthParams(seed=2)
    ref = synth.reference(sp.seed, l)
    jl.alloc(n, l)
    jl.synth_fill(sp, ref)
    genes = np.array([(1, l + 1)], dtype=capi.GENE)
    out = jl.run(genes, ref)
    rows = msa.unpack_columns(jl.download_columns(), n)
    pf = jl.pileup_fetch()
    # size-independent properties
    assert (pf["col_counts"].sum(axis=1) == n).all()                 # full-span reads: one symbol per read and column
    assert (pf["hist"].sum(axis=1) == pf["coverage"]).all()
    real_depth = pf["col_counts"][:, :4].sum(axis=1)
    assert (pf["coverage"] <= np.minimum.reduce([real_depth[pf["pos_col"] + k] for k in range(3)])).all()
    s = out["phase"]["summary"]
    assert s["reported_reads"] + s["insufficient_reads"] + s["damaged_reads"] == n
    assert abs((100.0 * out["phase"]["hap_count"] / s["reported_reads"]).sum() - 100.0) < 1e-9
    assert len(out["variants"]) == 5 and s["n_haplotypes"] == 5
    # and the oracle on the very same reads
    assert (pf["col_counts"] == oracle.pileup(rows)).all()
    exp_v = oracle.call(rows, genes, refseq=ref)
    assert_variants_equal(out["variants"], exp_v)
    assert_phase_equal(out["phase"], oracle.phase(rows, exp_v), len(exp_v))


# --------------------------------------------------------------------------------------------- graph replay
def test_run_graph_replay_tracks_new_data_and_parameters(jl, oracle):
    """jl_run_async replays a captured graph while nothing it bakes in changes: new reads in the same
    buffers must flow through the replay, and changed parameters / genes / shapes must re-capture."""
    n, l = 6000, 300
    genes = np.array([(1, l + 1)], dtype=capi.GENE)
    j = capi.Juliet(0)
    j.alloc(n, l)
    for seed, prm in ((41, capi.default_params()), (42, capi.default_params()), (43, capi.default_params()),
                      (43, capi.default_params(alpha=0.3, n_tests=1.0)), (44, capi.default_params(min_perc=3.0))):
        sp = synth.SynthParams(seed=seed, minor_permille=(70, 50, 40, 30), partial_rate=0.1)
        ref = synth.reference(41, l)          # same reference: the plan (and the graph) stay valid across seeds
        j.synth_fill(sp, ref)
        rows = msa.unpack_columns(j.download_columns(), n)
        out = j.run(genes, ref, prm)
        o = oracle_params(prm)
        exp_v = oracle.call(rows, genes, refseq=ref, params=o)
        if prm.min_perc >= 0:
            exp_v = exp_v[100.0 * exp_v["count"] / exp_v["coverage"] > prm.min_perc]
        assert_variants_equal(out["variants"], exp_v)
        assert_phase_equal(out["phase"], oracle.phase(rows, exp_v), len(exp_v))
    # different gene set and a different shape on the same context
    genes2 = np.array([(1, 151), (152, 299)], dtype=capi.GENE)
    out = j.run(genes2, ref, capi.default_params())
    assert_variants_equal(out["variants"], oracle.call(rows, genes2, refseq=ref))
    rows3 = synth.rows(synth.SynthParams(seed=5, minor_permille=(60, 60, 60, 60)), 90, 0, 2000)
    j.upload_columns(msa.pack_columns(rows3), 2000)
    g3 = np.array([(1, 91)], dtype=capi.GENE)
    out = j.run(g3, None, capi.default_params())
    exp_v = oracle.call(rows3, g3)
    assert_variants_equal(out["variants"], exp_v)
    assert_phase_equal(out["phase"], oracle.phase(rows3, exp_v), len(exp_v))
    # a configuration runs eagerly the first time and as a captured graph from its second run on: same answers
    prm3 = capi.default_params()
    for _ in range(3):
        out2 = j.run(g3, None, prm3)
        assert (out2["variants"] == out["variants"]).all()
        assert out2["phase"]["summary"] == out["phase"]["summary"]
        assert (out2["phase"]["read_hap"] == out["phase"]["read_hap"]).all()
    j.close()


def test_run_results_larger_than_the_pack(jl, oracle):
    """More variants / positions than the pinned result block holds: the fetch calls fall back to piecewise copies."""
    n, l = 6000, 600
    sp = synth.SynthParams(seed=31, sub_rate=0.01, minor_permille=(50, 50, 50, 50))
    ref = synth.reference(sp.seed, l)
    rows = synth.rows(sp, l, 0, n, ref)
    j = capi.Juliet(0)
    j.upload_columns(msa.pack_columns(rows), n)
    genes = np.array([(1, l + 1)], dtype=capi.GENE)
    loose = capi.default_params(alpha=0.9, n_tests=1.0)
    out = j.run(genes, None, loose, min_reads=3)
    exp_v = oracle.call(rows, genes, params=oracle_params(loose))
    assert len(exp_v) > 128
    assert_variants_equal(out["variants"], exp_v)
    assert_phase_equal(out["phase"], oracle.phase(rows, exp_v, min_reads=3), len(exp_v))
    j.close()


def test_run_view_and_completion_word(jl, oracle):
    """jl_run_wait / jl_run_view_get: completion through the pinned sequence word (no HIP sync) and results read
    in place must equal the copying fetch and the oracle, over many replays and with several contexts in flight;
    a result too large for the pinned block yields no view."""
    n, l = 9000, 300
    genes = np.array([(1, l + 1)], dtype=capi.GENE)
    ctxs, exps = [], []
    for k in range(3):
        sp = synth.SynthParams(seed=70 + k, minor_permille=(70, 50, 40, 30), partial_rate=0.05)
        ref = synth.reference(70, l)
        j = capi.Juliet(0)
        j.alloc(n, l)
        j.synth_fill(sp, ref)
        rows = msa.unpack_columns(j.download_columns(), n)
        exp_v = oracle.call(rows, genes, refseq=ref)
        ctxs.append((j, ref))
        exps.append((exp_v, oracle.phase(rows, exp_v)))
    prm = capi.default_params()
    for rep in range(12):
        for j, ref in ctxs:
            j.run_async(genes, ref, prm, None, True, 10, True)
        for (j, ref), (exp_v, exp_p) in zip(ctxs, exps):
            j.run_wait()
            assert j.run_done()
            v = j.run_view()
            assert v is not None
            assert_variants_equal(v["variants"], exp_v)
            assert_phase_equal(v["phase"], exp_p, len(exp_v))
            f = j.run_fetch(True, True, cap_var=64)
            assert (f["variants"] == v["variants"]).all()
            assert (f["phase"]["read_hap"] == v["phase"]["read_hap"]).all()
    # jl_time_run: the same runs launched and waited for inside the library leave the last run's results behind
    j, ref = ctxs[1]
    ms = j.time_run(genes, ref, prm, None, True, 10, True, reps=5)
    assert 0.0 < ms < 50.0
    v = j.run_view()
    assert_variants_equal(v["variants"], exps[1][0])
    assert_phase_equal(v["phase"], exps[1][1], len(exps[1][0]))
    # phasing off: the view carries only the table
    j, ref = ctxs[0]
    j.run_async(genes, ref, prm, None, False, 10, False)
    v = j.run_view()
    assert v is not None and "phase" not in v
    assert_variants_equal(v["variants"], exps[0][0])
    # too many rows for the block: no view, the fetch calls still work
    loose = capi.default_params(alpha=0.9, n_tests=1.0)
    j.run_async(genes, None, loose, None, True, 3, True)
    assert j.run_view() is None
    for j, _ in ctxs:
        j.close()


@pytest.mark.parametrize("fold", [True, False])
def test_group_run_equals_single_runs_and_oracle(jl, oracle, fold):
    """jl_group_run_async: several windows through the path in three launches (blockIdx.z = window).  Windows of
    different depth and noise, the same genes: every window's results must equal the oracle's (and so a single
    run's), over graph replays, after new reads were generated into the same buffers, and in majority-codon mode.
    fold = False: windows deep enough that the phasing launch has more workgroups than may wait for each other (128):
    the layout large groups use (per-read ids from a launch of their own, nothing waits in a launch)."""
    l = 300
    genes = np.array([(1, l + 1)], dtype=capi.GENE)
    k_ = 1 if fold else 15
    shapes = [(9000 * k_, 0.05), (5000 * k_, 0.0), (12345 * k_, 0.2), (2048 * k_, 0.1)]
    ref = synth.reference(90, l)
    ctxs = []
    for k, (n, partial) in enumerate(shapes):
        j = capi.Juliet(0)
        j.alloc(n, l)
        ctxs.append(j)
    grp = capi.Group(ctxs)
    prm = capi.default_params()

    def fill_and_expect(seed0, use_ref):
        exp = []
        for k, ((n, partial), j) in enumerate(zip(shapes, ctxs)):
            sp = synth.SynthParams(seed=seed0 + k, minor_permille=(70, 50, 40, 30), partial_rate=partial)
            j.synth_fill(sp, ref)
            j.sync()
            rows = msa.unpack_columns(j.download_columns(), n)
            exp_v = oracle.call(rows, genes, refseq=ref if use_ref else None)
            exp.append((exp_v, oracle.phase(rows, exp_v)))
        return exp

    for seed0, use_ref in ((90, True), (130, True), (170, False), (250, True)):
        exp = fill_and_expect(seed0, use_ref)
        for rep in range(4):
            grp.run_async(genes, ref if use_ref else None, prm, True, 10, True)
            if rep % 2:   # all windows' views in one call (jl_group_views) first: it waits for every window
                vw = grp.views()
                assert vw["complete"].all() and vw["phased"].all()
                assert list(vw["n_variants"]) == [len(e[0]) for e in exp]
                assert list(vw["n_haplotypes"]) == [len(e[1]["hap_count"]) for e in exp]
                assert list(vw["n_positions"]) == [e[1]["summary"]["n_positions"] for e in exp]
            for j, (exp_v, exp_p) in zip(ctxs, exp):
                v = j.run_view()
                assert v is not None
                assert_variants_equal(v["variants"], exp_v)
                assert_phase_equal(v["phase"], exp_p, len(exp_v))
        # the copying fetches read the same blocks
        f = ctxs[2].run_fetch(True, True, cap_var=64)
        assert_variants_equal(f["variants"], exp[2][0])
        assert_phase_equal(f["phase"], exp[2][1], len(exp[2][0]))
    # a single run on a member context still works afterwards (its own stream and graph)
    exp = fill_and_expect(210, True)
    out = ctxs[1].run(genes, ref, prm)
    assert_variants_equal(out["variants"], exp[1][0])
    assert_phase_equal(out["phase"], exp[1][1], len(exp[1][0]))
    grp.close()
    for j in ctxs:
        j.close()


def test_bench_configuration_group_of_eight_full_size_windows(oracle):
    """What bench.py times: ONE jl_group_run_async over 8 resident windows of 100k reads x 3 kb (BASELINE.json
    configs[2]), per-read ids from their own launch.  Every window against the oracle on the very same reads, twice
    (the second launch replays the captured graph), plus the size-independent properties."""
    n, l, g = 100_000, 3000, 8
    genes = np.array([(1, l + 1)], dtype=capi.GENE)
    ref = synth.reference(2, l)
    ctxs = []
    for k in range(g):
        j = capi.Juliet(0)
        j.alloc(n, l)
        j.synth_fill(synth.SynthParams(seed=2 + 1000 * k), ref)
        j.sync()
        ctxs.append(j)
    grp = capi.Group(ctxs)
    prm = capi.default_params()
    try:
        grp.run_async(genes, ref, prm, True, 10, True)
        first = []
        for j in ctxs:
            v = j.run_view()
            assert v is not None
            first.append((v["variants"].copy(), {k: (x.copy() if hasattr(x, "copy") else x) for k, x in v["phase"].items()}))
        grp.run_async(genes, ref, prm, True, 10, True)
        for k, j in enumerate(ctxs):
            v = j.run_view()
            rows = msa.unpack_columns(j.download_columns(), n)
            exp_v = oracle.call(rows, genes, refseq=ref)
            exp_p = oracle.phase(rows, exp_v)
            for got_v, got_p in ((v["variants"], v["phase"]), first[k]):
                assert_variants_equal(got_v, exp_v)
                assert_phase_equal(got_p, exp_p, len(exp_v))
            sm = v["phase"]["summary"]
            assert sm["reported_reads"] + sm["insufficient_reads"] + sm["damaged_reads"] == n
            assert int(v["phase"]["hap_count"].sum()) == sm["reported_reads"]
            assert len(exp_v) == 5 and sm["n_haplotypes"] >= 5   # a sixth: an error pattern seen in >= 10 reads
            del rows
    finally:
        grp.close()
        for j in ctxs:
            j.close()


@pytest.mark.parametrize("fold", [True, False])
def test_group_run_with_more_windows_than_a_stage_launch_takes(oracle, fold):
    """A group of 19 windows = chunks of 8 + 8 + 3 pipelined inside one captured graph (the counting of a chunk beside
    the phasing of the previous one, side streams forked and joined by events).  Every window against the oracle, over
    graph replays.  fold = False: deeper windows, the per-read ids come from a launch of their own."""
    l = 150
    genes = np.array([(1, l + 1)], dtype=capi.GENE)
    ref = synth.reference(41, l)
    ctxs, exp = [], []
    for k in range(19):
        n = (1500 + 433 * k) * (1 if fold else 12)
        j = capi.Juliet(0)
        j.alloc(n, l)
        j.synth_fill(synth.SynthParams(seed=41 + k, minor_permille=(60, 50, 40, 30), partial_rate=0.02 * k), ref)
        j.sync()
        rows = msa.unpack_columns(j.download_columns(), n)
        ev = oracle.call(rows, genes, refseq=ref)
        exp.append((ev, oracle.phase(rows, ev)))
        ctxs.append(j)
    grp = capi.Group(ctxs)
    try:
        for rep in range(3):
            grp.run_async(genes, ref, capi.default_params(), True, 10, True)
            for j, (ev, ep) in zip(ctxs, exp):
                v = j.run_view()
                assert v is not None
                assert_variants_equal(v["variants"], ev)
                assert_phase_equal(v["phase"], ep, len(ev))
    finally:
        grp.close()
        for j in ctxs:
            j.close()


def test_group_of_more_than_32_windows_is_refused():
    """The per-window argument blocks travel by value in the kernel arguments: a group holds at most 32 windows."""
    ctxs = [capi.Juliet(0) for _ in range(33)]
    try:
        with pytest.raises(capi.JulietError):
            capi.Group(ctxs)
        capi.Group(ctxs[:32]).close()
    finally:
        for j in ctxs:
            j.close()


def test_group_run_window_with_many_positions_falls_back(jl, oracle):
    """A window of a group with more than 10 variant positions: the grouped launch flags it (no view), its fetch calls
    re-run the multi-word pipeline and return the oracle's answer; the other window of the same launch is unaffected;
    afterwards that context is refused by group runs and served by jl_run_async."""
    l, n = 300, 6000
    genes = np.array([(1, l + 1)], dtype=capi.GENE)
    ref = synth.reference(77, l)
    loose = capi.default_params(alpha=0.9, n_tests=1.0)
    ctxs, rows_all = [], []
    for k, sub in enumerate((1.75e-4, 0.01)):   # the second window is noisy: dozens of called positions
        sp = synth.SynthParams(seed=77 + k, sub_rate=sub, minor_permille=(50, 50, 50, 50))
        rows = synth.rows(sp, l, 0, n, ref)
        j = capi.Juliet(0)
        j.upload_columns(msa.pack_columns(rows), n)
        j.sync()
        ctxs.append(j)
        rows_all.append(rows)
    grp = capi.Group(ctxs)
    grp.run_async(genes, ref, loose, True, 3, True)
    exp = []
    for rows in rows_all:
        ev = oracle.call(rows, genes, refseq=ref, params=oracle_params(loose))
        exp.append((ev, oracle.phase(rows, ev, min_reads=3)))
    assert len(np.unique(exp[1][0]["col"])) > 10
    v0 = ctxs[0].run_view()
    if len(np.unique(exp[0][0]["col"])) <= 10:
        assert v0 is not None
        assert_variants_equal(v0["variants"], exp[0][0])
        assert_phase_equal(v0["phase"], exp[0][1], len(exp[0][0]))
    assert ctxs[1].run_view() is None
    cap = capi.VARIANT_CAP
    f = ctxs[1].run_fetch(True, True, cap_var=cap)
    assert_variants_equal(f["variants"], exp[1][0])
    assert_phase_equal(f["phase"], exp[1][1], len(exp[1][0]))
    with pytest.raises(capi.JulietError):
        grp.run_async(genes, ref, loose, True, 3, True)   # that context now needs the multi-word pipeline
    out = ctxs[1].run(genes, ref, loose, min_reads=3)
    assert_variants_equal(out["variants"], exp[1][0])
    assert_phase_equal(out["phase"], exp[1][1], len(exp[1][0]))
    grp.close()
    for j in ctxs:
        j.close()


# --------------------------------------------------------------------------------------------- the collective
def test_allgather_variants_single_rank_communicator(jl, oracle):
    """jl_allgather_variants over a real RCCL communicator (world = 1 is all one GPU allows here): the payload
    layout and the host unpacking are the ones the 8-GPU run uses."""
    import ctypes as C
    sp = synth.SynthParams(seed=23, minor_permille=(70, 60, 50, 40))
    ref = synth.reference(sp.seed, 300)
    rows = synth.rows(sp, 300, 0, 5000, ref)
    genes = np.array([(1, 301)], dtype=capi.GENE)
    jl.upload_columns(msa.pack_columns(rows), 5000)
    idbuf = np.zeros(128, dtype=np.uint8)
    assert jl.lib.jl_comm_unique_id(idbuf.ctypes.data_as(C.c_void_p)) == 0
    comm = C.c_void_p()
    jl._chk(jl.lib.jl_comm_create(jl.h, idbuf.ctypes.data_as(C.c_void_p), 0, 1, C.byref(comm)))
    try:
        jl.run_async(genes, ref, capi.default_params(), None, True, 10, True)
        all_rows = np.zeros(capi.VARIANT_CAP, dtype=capi.VARIANT)
        counts = np.zeros(1, dtype=np.uint32)
        jl._chk(jl.lib.jl_allgather_variants(jl.h, comm, all_rows.ctypes.data_as(C.c_void_p),
                                             counts.ctypes.data_as(C.c_void_p), capi.VARIANT_CAP))
        exp = oracle.call(rows, genes, refseq=ref)
        assert counts[0] == len(exp)
        assert_variants_equal(all_rows[: counts[0]], exp)
        out = jl.run_fetch(True, True)
        assert_variants_equal(out["variants"], exp)
    finally:
        jl.lib.jl_comm_destroy(comm)


def test_allgather_of_a_group_run_in_one_rccl_group(jl, oracle):
    """jl_allgather_variants_async_many: the exchanges of the windows of a group run issued as one RCCL group by the
    worker thread (no HIP call on the launching thread), two rounds pending per context, collected oldest first."""
    import ctypes as C
    l = 300
    genes = np.array([(1, l + 1)], dtype=capi.GENE)
    ref = synth.reference(51, l)
    ctxs, exp = [], []
    for k, n in enumerate((4000, 7000, 2500)):
        sp = synth.SynthParams(seed=51 + k, minor_permille=(70, 60, 50, 40))
        rows = synth.rows(sp, l, 0, n, ref)
        j = capi.Juliet(0)
        j.upload_columns(msa.pack_columns(rows), n)
        j.sync()
        ctxs.append(j)
        exp.append(oracle.call(rows, genes, refseq=ref))
    grp = capi.Group(ctxs)
    idbuf = np.zeros(128, dtype=np.uint8)
    assert jl.lib.jl_comm_unique_id(idbuf.ctypes.data_as(C.c_void_p)) == 0
    comm = C.c_void_p()
    ctxs[0]._chk(jl.lib.jl_comm_create(ctxs[0].h, idbuf.ctypes.data_as(C.c_void_p), 0, 1, C.byref(comm)))
    arr = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
    all_rows = np.zeros(capi.VARIANT_CAP, dtype=capi.VARIANT)
    counts = np.zeros(1, dtype=np.uint32)
    try:
        prm = capi.default_params()
        for rnd in range(2):   # two exchanges pending per context before the first is collected
            grp.run_async(genes, ref, prm, True, 10, True)
            assert jl.lib.jl_allgather_variants_async_many(arr, len(ctxs), comm) == 0
        with pytest.raises(capi.JulietError):   # a third run would overwrite a result block an exchange still reads
            grp.run_async(genes, ref, prm, True, 10, True)
        assert jl.lib.jl_allgather_variants_async_many(arr, len(ctxs), comm) == -4   # and so is a third exchange
        for rnd in range(2):
            for c, e in zip(ctxs, exp):
                c._chk(jl.lib.jl_allgather_variants(c.h, comm, all_rows.ctypes.data_as(C.c_void_p),
                                                    counts.ctypes.data_as(C.c_void_p), capi.VARIANT_CAP))
                assert counts[0] == len(e)
                assert_variants_equal(all_rows[: counts[0]], e)
        for c, e in zip(ctxs, exp):
            assert_variants_equal(c.run_view()["variants"], e)
    finally:
        jl.lib.jl_comm_destroy(comm)
        grp.close()
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("staged", [False, True])
def test_group_run_carries_its_exchange(jl, oracle, staged):
    """jl_group_exchange_bind over RCCL (one rank): the heads of the windows' tables are written by the run's last kernels
    into the pinned region the all-gather works in (staged: into its device stage, copied behind the collective); two
    exchanges pending per group, collected oldest first; a third run is refused; with and without phasing; a table of more
    than 128 rows takes the full stride."""
    import ctypes as C
    import subprocess
    import sys
    if staged:   # the form is chosen once per communicator from the environment: a child process
        env = dict(os.environ, JL_EXCHANGE_STAGED="1")
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", __file__, "-k", "test_group_run_carries_its_exchange and False"],
                           env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        return
    l = 300
    genes = np.array([(1, l + 1)], dtype=capi.GENE)
    ref = synth.reference(51, l)
    ctxs, exp = [], []
    for k, n in enumerate((4000, 7000, 2500)):
        sp = synth.SynthParams(seed=51 + k, minor_permille=(70, 60, 50, 40))
        rows = synth.rows(sp, l, 0, n, ref)
        j = capi.Juliet(0)
        j.upload_columns(msa.pack_columns(rows), n)
        j.sync()
        ctxs.append(j)
        exp.append(oracle.call(rows, genes, refseq=ref))
    grp = capi.Group(ctxs)
    idbuf = np.zeros(128, dtype=np.uint8)
    assert jl.lib.jl_comm_unique_id(idbuf.ctypes.data_as(C.c_void_p)) == 0
    comm = C.c_void_p()
    ctxs[0]._chk(jl.lib.jl_comm_create(ctxs[0].h, idbuf.ctypes.data_as(C.c_void_p), 0, 1, C.byref(comm)))
    try:
        prm = capi.default_params()
        with pytest.raises(capi.JulietError):
            grp.exchange_collect(1)                       # nothing bound
        grp.bind_exchange(comm)
        with pytest.raises(capi.JulietError):
            grp.exchange_collect(1)                       # nothing pending
        for phasing in (True, False):
            for rnd in range(2):   # two exchanges pending before the first is collected
                grp.run_async(genes, ref, prm, phasing, 10, phasing)
            with pytest.raises(capi.JulietError):
                grp.run_async(genes, ref, prm, phasing, 10, phasing)
            for rnd in range(2):
                rows_, counts = grp.exchange_collect(1, cap_rows=capi.VARIANT_CAP)
                for k, e in enumerate(exp):
                    assert counts[k, 0] == len(e)
                    assert_variants_equal(rows_[k, 0, : counts[k, 0]], e)
            for c, e in zip(ctxs, exp):
                assert_variants_equal(c.run_view()["variants"], e)
        # a loose threshold: more than 128 rows in a window -> the full stride, window by window
        loose = capi.default_params(alpha=1e9)
        grp.run_async(genes, ref, loose, False, 10, False)
        rows_, counts = grp.exchange_collect(1, cap_rows=capi.VARIANT_CAP)
        for k, c in enumerate(ctxs):
            own = c.call_fetch()
            assert len(own) > 128 and counts[k, 0] == len(own)
            assert_variants_equal(rows_[k, 0, : counts[k, 0]], own)
        # a run that fails on this rank before anything is launched still issues its collective, with empty heads: the
        # caller gets the run's own error, the collecting call (here and on every peer) that rank's failure
        with pytest.raises(capi.JulietError, match="tail"):
            grp.run_async(genes, ref, capi.default_params(tail=2), True, 10, True)
        with pytest.raises(capi.JulietError, match="did not complete"):
            grp.exchange_collect(1)
        grp.run_async(genes, ref, prm, True, 10, True)    # and the group goes on
        rows_, counts = grp.exchange_collect(1)
        assert [int(c) for c in counts[:, 0]] == [len(e) for e in exp]
        grp.bind_exchange(None)
        grp.run_async(genes, ref, prm, True, 10, True)    # unbound again: a plain group run
        for c, e in zip(ctxs, exp):
            assert_variants_equal(c.run_view()["variants"], e)
    finally:
        grp.close()
        jl.lib.jl_comm_destroy(comm)
        for c in ctxs:
            c.close()


def test_bound_exchange_of_a_group_cut_into_chunks(jl, oracle):
    """Ten windows = two chunks of the group pipeline (the first chunk's tail runs on a side stream beside the second chunk's
    pileup): every window's head must be in the region before the all-gather behind the graph reads it."""
    import ctypes as C
    l = 300
    genes = np.array([(1, l + 1)], dtype=capi.GENE)
    ref = synth.reference(61, l)
    ctxs, exp = [], []
    for k in range(10):
        sp = synth.SynthParams(seed=61 + k, minor_permille=(70, 60, 50, 40))
        rows = synth.rows(sp, l, 0, 1500 + 100 * k, ref)
        j = capi.Juliet(0)
        j.upload_columns(msa.pack_columns(rows), len(rows))
        j.sync()
        ctxs.append(j)
        exp.append(oracle.call(rows, genes, refseq=ref))
    grp = capi.Group(ctxs)
    idbuf = np.zeros(128, dtype=np.uint8)
    assert jl.lib.jl_comm_unique_id(idbuf.ctypes.data_as(C.c_void_p)) == 0
    comm = C.c_void_p()
    ctxs[0]._chk(jl.lib.jl_comm_create(ctxs[0].h, idbuf.ctypes.data_as(C.c_void_p), 0, 1, C.byref(comm)))
    try:
        grp.bind_exchange(comm)
        for rnd in range(3):
            grp.run_async(genes, ref, capi.default_params(), True, 10, True)
            rows_, counts = grp.exchange_collect(1)
            for k, e in enumerate(exp):
                assert counts[k, 0] == len(e), (rnd, k)
                assert_variants_equal(rows_[k, 0, : counts[k, 0]], e)
    finally:
        grp.close()
        jl.lib.jl_comm_destroy(comm)
        for c in ctxs:
            c.close()


# --------------------------------------------------------------------------------------------- device ingest
def rows_to_records(rows, ref, rng, with_noise_ops=True):
    """Re-express by-row symbols as BAM-style records (pos, cigar words, 4-bit packed bases, qualities):
    '='/'X' for bases, 'D' for '-', 'N' bases for filtered ones, plus insertions / soft clips / hard clips that
    an ingest must ignore (doc/JULIET.md:26-27, 53)."""
    OPS = {"M": 0, "I": 1, "D": 2, "N": 3, "S": 4, "H": 5, "P": 6, "=": 7, "X": 8}
    nt16 = {0: 1, 1: 2, 2: 4, 3: 8, 4: 15}
    pos, cigar, cig_off, seq4, seq_off, qual, qual_off = [], [], [0], [], [0], [], [0]
    for row in rows:
        cov = np.nonzero(row != 6)[0]
        ops, bases, quals = [], [], []
        if len(cov) == 0:
            pos.append(0)
            ops.append((OPS["S"], 1)); bases.append(0); quals.append(30)
        else:
            a, b = int(cov[0]), int(cov[-1]) + 1
            pos.append(a)
            if with_noise_ops and rng.random() < 0.5:
                ops.append((OPS["H"], 3))
            if with_noise_ops and rng.random() < 0.5:
                k = int(rng.integers(1, 4)); ops.append((OPS["S"], k)); bases += [int(x) for x in rng.integers(0, 4, k)]; quals += [20] * k
            for c in range(a, b):
                s = int(row[c])
                if s == 4:
                    ops.append((OPS["D"], 1))
                elif s == 6:
                    ops.append((OPS["N"], 1))
                else:
                    base = 4 if s == 5 else s
                    ops.append((OPS["="] if s == ref[c] else OPS["X"], 1)); bases.append(base); quals.append(93)
                    if with_noise_ops and rng.random() < 0.01:
                        k = int(rng.integers(1, 3)); ops.append((OPS["I"], k)); bases += [int(x) for x in rng.integers(0, 4, k)]; quals += [10] * k
            if with_noise_ops and rng.random() < 0.3:
                ops.append((OPS["S"], 2)); bases += [1, 2]; quals += [5, 5]
        merged = []
        for op, ln in ops:
            if merged and merged[-1][0] == op:
                merged[-1][1] += ln
            else:
                merged.append([op, ln])
        cigar += [(ln << 4) | op for op, ln in merged]
        cig_off.append(len(cigar))
        packed = []
        for i in range(0, len(bases), 2):
            hi = nt16[bases[i]]
            lo = nt16[bases[i + 1]] if i + 1 < len(bases) else 0
            packed.append((hi << 4) | lo)
        seq4 += packed
        seq_off.append(len(seq4))
        qual += quals
        qual_off.append(len(qual))
    return (np.array(pos, dtype=np.int32), np.array(cigar, dtype=np.uint32), np.array(cig_off, dtype=np.uint64),
            np.array(seq4, dtype=np.uint8), np.array(seq_off, dtype=np.uint64), np.array(qual, dtype=np.uint8),
            np.array(qual_off, dtype=np.uint64))


@pytest.mark.parametrize("n,l,partial,win", [(300, 120, 0.3, (0, 120)), (1000, 400, 0.2, (37, 351)), (70, 3000, 0.1, (0, 3000)),
                                              (4000, 500, 0.3, (64, 500))])
def test_device_ingest_matches_rows(jl, n, l, partial, win):
    rng = np.random.default_rng(n + l)
    sp = synth.SynthParams(seed=n + l, partial_rate=partial, del_rate=0.02, mask_rate=0.03, sub_rate=0.01)
    ref = synth.reference(sp.seed, l)
    rows = synth.rows(sp, l, 0, n, ref)
    rows[5, 40:60] = 6            # a reference skip inside a read (cigar N): stays uncovered
    rows[7] = 6                   # a read with no aligned base at all
    pos, cigar, cig_off, seq4, seq_off, qual, qual_off = rows_to_records(rows, ref, rng)
    b, e = win
    jl.ingest_records(e - b, b, pos, cigar, cig_off, seq4, seq_off)
    got = msa.unpack_columns(jl.download_columns(), n)
    assert (got == rows[:, b:e]).all()
    # QV masking: soft-clip / insertion qualities never matter; masking the '=' / 'X' bases turns them all into N
    jl.ingest_records(e - b, b, pos, cigar, cig_off, seq4, seq_off, qual, qual_off, min_qv=94)
    got = msa.unpack_columns(jl.download_columns(), n)
    exp = rows[:, b:e].copy()
    exp[exp < 4] = 5
    assert (got == exp).all()
    jl.ingest_records(e - b, b, pos, cigar, cig_off, seq4, seq_off, qual, qual_off, min_qv=50)
    assert (msa.unpack_columns(jl.download_columns(), n) == rows[:, b:e]).all()
    # cigar M is rejected (doc/JULIET.md:53)
    bad = cigar.copy()
    bad[0] = (bad[0] >> 4 << 4) | 0
    with pytest.raises(capi.JulietError) as err:
        jl.ingest_records(e - b, b, pos, bad, cig_off, seq4, seq_off)
    assert "cigar M" in str(err.value)
    # every 4-bit base that is not exactly A, C, G or T — '=', the IUPAC ambiguity codes, N — is a filtered base: the N's of the
    # records replaced by all twelve of them in turn give the same matrix
    other = np.array([0, 3, 5, 6, 7, 9, 10, 11, 12, 13, 14, 15], dtype=np.uint8)
    amb = seq4.copy()
    hi, lo = amb >> 4, amb & 15
    hi = np.where(hi == 15, other[rng.integers(0, len(other), len(amb))], hi)
    lo = np.where(lo == 15, other[rng.integers(0, len(other), len(amb))], lo)
    amb = ((hi << 4) | lo).astype(np.uint8)
    assert (amb != seq4).any()
    jl.ingest_records(e - b, b, pos, cigar, cig_off, amb, seq_off)
    assert (msa.unpack_columns(jl.download_columns(), n) == rows[:, b:e]).all()


# ---- the ingest at the sizes the bench runs it at (VERDICT r04 task 2): every cell of every read, against the numpy statement of
# "records -> matrix" (tests/records_expand.py, pinned on the CPU by tests/test_records_expand.py) and, for three groups of 1024
# reads (the first, one served in a later round of its XCD, the last and partial one), against the generator's own rows.
def _cells_equal(window, rec, n, n_cols, win_begin=0, min_qv=0):
    import records_expand
    packed = window.download_columns()
    slab = 1 << 15
    for r0 in range(0, n, slab):                      # (by slabs of reads: the expected matrix never exists whole)
        r1 = min(n, r0 + slab)
        exp = records_expand.expand(rec, n_cols, win_begin, min_qv, read_begin=r0, read_end=r1)
        by = packed[:, r0 // 2:(r1 + 1) // 2]
        got = np.empty((n_cols, 2 * by.shape[1]), dtype=np.uint8)
        got[:, 0::2] = by & 15
        got[:, 1::2] = by >> 4
        if not (got[:, :r1 - r0].T == exp).all():
            bad = np.argwhere(got[:, :r1 - r0].T != exp)[0]
            return "read %d column %d: got %d, expected %d" % (r0 + bad[0], bad[1], got[bad[1], bad[0]], exp[bad[0], bad[1]])
    return None


@pytest.mark.parametrize("n,l", [(100_000, 3000), (333_333, 1000), (2000, 600), (5000, 640)])
def test_device_ingest_at_size(jl, n, l):
    """100k x 3000 (the bench's window: 98 groups of 1024 reads x 12 sweeps) and 333 333 x 1000 (326 groups: they do not divide
    among the eight XCDs, an XCD serves forty of them in turn): jl_records_window of the resident records == every cell.
    2000 x 600 and 5000 x 640: windows whose LAST sweep every read ends in — each read has several entries there, a workgroup's
    entry list is longer than its thread count (a second pass of the table fill: round 5 shipped a slot mapping that was
    wrong exactly there, for a few hours)."""
    rec = synth.raw_records(5, n, l)
    jl.records_upload(rec["pos"], rec["cigar"], rec["cig_off"], rec["seq4"], rec["seq_off"])
    w = capi.Juliet(0)
    try:
        w.records_window(jl, l, 0, 0)
        assert _cells_equal(w, rec, n, l) is None
        got = None
        sp, ref = synth.SynthParams(seed=5), synth.reference(5, l)
        packed = w.download_columns()
        last = (n - 1) // 1024
        for g in sorted({0, min((n // 1024) // 2 + 3, last), last}):
            r0, r1 = 1024 * g, min(n, 1024 * g + 1024)
            got = msa.unpack_columns(np.ascontiguousarray(packed[:, r0 // 2:(r1 + 1) // 2]), r1 - r0)
            assert (got == synth.rows(sp, l, r0, r1, ref)).all(), g
    finally:
        w.close()
        jl.records_drop()


def test_device_ingest_rich_qv_at_size(jl):
    """The documented input shape at the bench's size (bench.py once_through_qv; doc/JULIET.md:256-259, 273-276): 100k x 3000
    `ccs --richQVs`-style records — a filtered base keeps its letter and carries a low quality, ten cigar ops a read, one quality
    byte per base — through the QV instantiations of both ingest kernels with min_qv = 20: every cell against the numpy
    statement, three groups of 1024 reads against the generator's own rows (where the filtered bases are N), and the same
    records without a threshold (min_qv = 0) keep their letters."""
    n, l = 100_000, 3000
    rec = synth.raw_records(5, n, l, extra=("--rich-qv",))
    assert len(rec["cigar"]) / n < 16 and len(rec["qual"]) >= n * (l - 50)
    jl.records_upload(rec["pos"], rec["cigar"], rec["cig_off"], rec["seq4"], rec["seq_off"], rec["qual"], rec["qual_off"])
    w = capi.Juliet(0)
    try:
        w.records_window(jl, l, 0, 20)
        assert _cells_equal(w, rec, n, l, 0, 20) is None
        sp, ref = synth.SynthParams(seed=5), synth.reference(5, l)
        packed = w.download_columns()
        last = (n - 1) // 1024
        for g in sorted({0, 52, last}):
            r0, r1 = 1024 * g, min(n, 1024 * g + 1024)
            got = msa.unpack_columns(np.ascontiguousarray(packed[:, r0 // 2:(r1 + 1) // 2]), r1 - r0)
            assert (got == synth.rows(sp, l, r0, r1, ref)).all(), g
        # a window that begins and ends inside the reads, on no sweep boundary
        w.records_window(jl, 1777, 611, 20)
        assert _cells_equal(w, rec, n, 1777, 611, 20) is None
        w.records_window(jl, l, 0, 0)
        assert _cells_equal(w, rec, n, l, 0, 0) is None
    finally:
        w.close()
        jl.records_drop()


@pytest.mark.parametrize("n,l,win,min_qv", [(24_000, 1100, (37, 1039), 20), (24_000, 1100, (0, 1100), 0), (21_000, 700, (300, 693), 20)])
def test_device_ingest_qv_and_ragged_window_at_size(jl, n, l, win, min_qv):
    """Tens of groups through the QV path and through windows that begin inside the reads and end on no sweep boundary
    (n_cols % 256 != 0), on records with what an ingest must ignore: insertions of 1-4 bases, soft and hard clips, and 2 % of
    the aligned bases below the threshold."""
    rec = synth.raw_records(9, n, l, extra=("--ins-ppm", "2500", "--clips", "--low-qv-ppm", "20000", "--partial", "0.3"))
    b, e = win
    assert (e - b) % 256 != 0
    jl.records_upload(rec["pos"], rec["cigar"], rec["cig_off"], rec["seq4"], rec["seq_off"], rec["qual"], rec["qual_off"])
    w = capi.Juliet(0)
    try:
        w.records_window(jl, e - b, b, min_qv)
        assert _cells_equal(w, rec, n, e - b, b, min_qv) is None
    finally:
        w.close()
        jl.records_drop()


@pytest.mark.parametrize("n,l,ins_ppm,del_rate,min_qv", [(8_000, 3000, 5000, 0.005, 0), (8_000, 3000, 10000, 0.01, 20), (6_000, 3000, 20000, 0.02, 0),
                                                         (1_200, 24000, 10000, 0.01, 0)])
def test_device_ingest_indel_rich_reads(jl, n, l, ins_ppm, del_rate, min_qv):
    """Reads with an indel every 100, 50, 25 columns: (tile, sweep) units with more entries than the planes kernel's first size
    holds (handed on to its second), reads with more entries than cigar_runs' first launch keeps in LDS (left to its second; at
    24 000 columns more than the second keeps, too), sweeps with more inserted bases than a staging row has room for and units
    that overflow even the second size (the column-by-column kernel) — every cell against the records' expansion."""
    rec = synth.raw_records(11, n, l, extra=("--ins-ppm", str(ins_ppm), "--del", str(del_rate), "--low-qv-ppm", "20000"))
    assert len(rec["cigar"]) / n > 64
    if min_qv:
        jl.records_upload(rec["pos"], rec["cigar"], rec["cig_off"], rec["seq4"], rec["seq_off"], rec["qual"], rec["qual_off"])
    else:
        jl.records_upload(rec["pos"], rec["cigar"], rec["cig_off"], rec["seq4"], rec["seq_off"])
    w = capi.Juliet(0)
    try:
        # (the third window ends in the MIDDLE of the reads: all of a read's runs behind it are clamped to its last column — ADVICE
        # r05: counted as entries of the last sweep they sent every unit to the second size or to slow_pair)
        for b, e in ((0, l), (131, l - 77), (200, l // 2 + 13)):
            w.records_window(jl, e - b, b, min_qv)
            assert _cells_equal(w, rec, n, e - b, b, min_qv) is None, (b, e)
    finally:
        w.close()
        jl.records_drop()


def _records_from_cigars(cigars, rng, pos=None):
    """records (the dict of synth.raw_records) from lists of (op letter, length); bases drawn at random, qualities 93"""
    OPS = {"M": 0, "I": 1, "D": 2, "N": 3, "S": 4, "H": 5, "P": 6, "=": 7, "X": 8}
    rec = {"pos": [], "cigar": [], "cig_off": [0], "seq4": [], "seq_off": [0], "qual": [], "qual_off": [0]}
    for i, ops in enumerate(cigars):
        rec["pos"].append(0 if pos is None else pos[i])
        rec["cigar"] += [(ln << 4) | OPS[op] for op, ln in ops]
        rec["cig_off"].append(len(rec["cigar"]))
        nq = sum(ln for op, ln in ops if op in "=XIS")
        codes = [int(x) for x in rng.choice([1, 2, 4, 8], nq)] + [0]
        rec["seq4"] += [(codes[k] << 4) | codes[k + 1] for k in range(0, nq, 2)]
        rec["seq_off"].append(len(rec["seq4"]))
        rec["qual"] += [93] * nq
        rec["qual_off"].append(len(rec["qual"]))
    kinds = {"pos": np.int32, "cigar": np.uint32, "cig_off": np.uint64, "seq4": np.uint8, "seq_off": np.uint64, "qual": np.uint8, "qual_off": np.uint64}
    return {k: np.array(v, dtype=kinds[k]) for k, v in rec.items()}


@pytest.mark.parametrize("with_long", [False, True])
def test_device_ingest_reads_at_the_long_read_boundary(jl, with_long):
    """cigar_walk_kernel leaves a read of more than 192 ops or more than 37 runs to the launch for long reads, and the upload looks
    at the cigars (jl_ingest_read_is_long) to find out whether that launch is needed at all: reads of exactly 35..37 runs, of 192
    ops that are one run ('=' and 'X' in turns), with insertions and zero-length ops that end runs — none long (the launch is not
    made), then the same with reads of 38, 39 and 60 runs and of 193 ops among them (it is) — every cell."""
    rng = np.random.default_rng(29 + with_long)
    l = 700

    def alternating(n_runs, gap="D"):      # n_runs runs: aligned, gap, aligned, ... over about 600 columns
        n_al = (n_runs + 1) // 2
        ops = []
        for k in range(n_runs):
            ops.append(("=", 600 // n_runs) if k % 2 == 0 else (gap, 1 + k % 3))
        assert sum(1 for op, _ in ops if op == "=") == n_al
        return ops

    cigars = []
    for n_runs in (1, 5, 35, 36, 37):
        cigars += [alternating(n_runs), alternating(n_runs, "N"), [("S", 3)] + alternating(n_runs) + [("H", 2)]]
    cigars.append([("=", 2) if k % 2 == 0 else ("X", 1) for k in range(192)])                       # 192 ops, one run
    cigars.append(sum(([("=", 20), ("I", 2)] for _ in range(14)), []) + [("=", 20)])                # insertions end runs: 15 runs, 29 ops
    cigars.append(sum(([("=", 9), ("D", 0), ("X", 1)] for _ in range(36)), []) + [("=", 5)])        # zero-length ops end runs: 37 runs, 109 ops
    cigars.append(sum(([("=", 9), ("P", 1), ("X", 1)] for _ in range(14)), []))                     # so do pads: 15 runs
    if with_long:
        for n_runs in (38, 39, 60):
            cigars += [alternating(n_runs), alternating(n_runs, "N")]
        cigars.append([("=", 2) if k % 2 == 0 else ("X", 1) for k in range(193)])                   # one run, but 193 ops
        cigars.append(sum(([("=", 9), ("D", 0), ("X", 1)] for _ in range(37)), []) + [("=", 5)])    # 38 runs
    order = rng.permutation(len(cigars))
    cigars = [cigars[i] for i in order] * 3             # (several tiles' worth would need thousands: three copies, different starts)
    pos = [int(p) for p in rng.integers(0, 60, len(cigars))]
    rec = _records_from_cigars(cigars, rng, pos)
    n = len(cigars)
    jl.records_upload(rec["pos"], rec["cigar"], rec["cig_off"], rec["seq4"], rec["seq_off"], rec["qual"], rec["qual_off"])
    w = capi.Juliet(0)
    try:
        for b, e, min_qv in ((0, l, 0), (17, 640, 20), (300, 301, 0)):
            w.records_window(jl, e - b, b, min_qv)
            assert _cells_equal(w, rec, n, e - b, b, min_qv) is None, (b, e, min_qv)
    finally:
        w.close()
        jl.records_drop()


def test_device_ingest_random_qv_shapes(jl):
    """A seeded slice of tools_tuning/ingest_stress_qv.py: rich-QV and plain records with insertions, clips, poor qualities and
    deletion rates drawn at random, windows that begin and end inside the reads, thresholds from 1 to 127 — every cell against
    the numpy statement (the quality path: unaligned 16-byte quality loads, masks per nibble, two tiles a workgroup)."""
    import records_expand
    rng = np.random.default_rng(606)
    w = capi.Juliet(0)
    try:
        for k in range(8):
            n = int(rng.integers(1, 9000))
            l = int(rng.integers(40, 2600))
            extra = ["--rich-qv"] if rng.random() < 0.7 else []
            extra += ["--ins-ppm", str(int(rng.choice([0, 800, 5000, 30000]))), "--low-qv-ppm", str(int(rng.choice([0, 20000, 300000]))),
                      "--del", str(float(rng.choice([0.0, 0.0013, 0.01, 0.08]))), "--partial", str(float(rng.choice([0.0, 0.3, 0.9])))]
            if rng.random() < 0.5:
                extra += ["--clips"]
            rec = synth.raw_records(3000 + k, n, l, extra=tuple(extra))
            jl.records_upload(rec["pos"], rec["cigar"], rec["cig_off"], rec["seq4"], rec["seq_off"], rec["qual"], rec["qual_off"])
            for _ in range(2):
                b = int(rng.integers(0, max(1, l // 2)))
                e = int(rng.integers(b + 1, l + 1))
                min_qv = int(rng.choice([1, 5, 13, 20, 60, 94, 127]))
                w.records_window(jl, e - b, b, min_qv)
                got = msa.unpack_columns(w.download_columns(), n)
                assert (got == records_expand.expand(rec, e - b, b, min_qv)).all(), (k, n, l, b, e, min_qv, extra)
            jl.records_drop()
    finally:
        w.close()


def test_device_ingest_random_shapes(jl):
    """A seeded slice of tools_tuning/ingest_stress.py: read counts, widths, indel and mask rates and windows drawn at random,
    with and without qualities."""
    rng = np.random.default_rng(int(os.environ.get("JL_TEST_SEED", "2025")))     # (JL_TEST_SEED: a soak with other draws)
    for k in range(10):
        n = int(rng.integers(1, 6000))
        l = int(rng.integers(30, 1500))
        sp = synth.SynthParams(seed=2000 + k, partial_rate=float(rng.uniform(0, 0.6)), del_rate=float(rng.choice([0.0, 0.002, 0.05, 0.3])),
                               mask_rate=float(rng.choice([0.0, 0.02, 0.4])), sub_rate=0.01)
        ref = synth.reference(sp.seed, l)
        rows = synth.rows(sp, l, 0, n, ref)
        if n > 8:
            rows[5] = 6
            rows[3, : l // 2] = 6
        pos, cigar, cig_off, seq4, seq_off, qual, qual_off = rows_to_records(rows, ref, rng)
        b = int(rng.integers(0, max(1, l // 2)))
        e = int(rng.integers(b + 1, l + 1))
        jl.ingest_records(e - b, b, pos, cigar, cig_off, seq4, seq_off)
        assert (msa.unpack_columns(jl.download_columns(), n) == rows[:, b:e]).all(), (k, n, l, b, e)
        jl.ingest_records(e - b, b, pos, cigar, cig_off, seq4, seq_off, qual, qual_off, min_qv=50)
        assert (msa.unpack_columns(jl.download_columns(), n) == rows[:, b:e]).all(), ("qv", k, n, l, b, e)


@pytest.mark.parametrize("chunk,hints", [(1, (0, 0, 0, 0)), (37, (0, 0, 0, 0)), (256, (5000, 100000, 1 << 20, 1 << 21)), (10000, (0, 0, 0, 0))])
def test_device_ingest_in_chunks(jl, chunk, hints):
    """jl_records_begin / _append / _finish: any chunking (one read per append, ragged chunks, one chunk; device arrays
    that grow from nothing or are sized up front) gives the matrix of the one-call form, with and without qualities,
    and the insertion counters as well."""
    n, l = 1500, 333
    rng = np.random.default_rng(chunk)
    sp = synth.SynthParams(seed=77, partial_rate=0.3, del_rate=0.02, mask_rate=0.03, sub_rate=0.01)
    ref = synth.reference(sp.seed, l)
    rows = synth.rows(sp, l, 0, n, ref)
    rows[11] = 6
    pos, cigar, cig_off, seq4, seq_off, qual, qual_off = rows_to_records(rows, ref, rng)
    jl.track_insertions(True)
    jl.ingest_records(l - 20, 13, pos, cigar, cig_off, seq4, seq_off)
    want = jl.download_columns().copy()
    want_ins = jl.insertions_fetch()
    jl.ingest_records_chunked(l - 20, 13, pos, cigar, cig_off, seq4, seq_off, chunk_reads=chunk, hints=hints)
    assert (jl.download_columns() == want).all()
    assert (msa.unpack_columns(jl.download_columns(), n) == rows[:, 13:l - 7]).all()
    got_ins = jl.insertions_fetch()
    assert (got_ins[0] == want_ins[0]).all() and (got_ins[1] == want_ins[1]).all()
    jl.track_insertions(False)
    jl.ingest_records_chunked(l - 20, 13, pos, cigar, cig_off, seq4, seq_off, qual, qual_off, min_qv=94, chunk_reads=chunk, hints=hints)
    exp = rows[:, 13:l - 7].copy()
    exp[exp < 4] = 5
    assert (msa.unpack_columns(jl.download_columns(), n) == exp).all()
    # errors are those of the one-call form, and a failed append leaves nothing behind
    bad = cigar.copy()
    bad[int(cig_off[n // 2])] &= ~np.uint32(15)
    with pytest.raises(capi.JulietError) as err:
        jl.ingest_records_chunked(l, 0, pos, bad, cig_off, seq4, seq_off, chunk_reads=chunk)
    assert "cigar M" in str(err.value)
    with pytest.raises(capi.JulietError) as err:      # a failed append ends the stream: nothing to finish
        jl._chk(jl.lib.jl_records_finish(jl.h, l, 0, 0))
    assert "before jl_records_begin" in str(err.value)
    jl.ingest_records_chunked(l, 0, pos, cigar, cig_off, seq4, seq_off, chunk_reads=chunk)
    assert (msa.unpack_columns(jl.download_columns(), n) == rows).all()


def test_device_ingest_refuses_cigars_whose_lengths_wrap(jl):
    """Untrusted records through the C ABI: four insertions of 2^28 - 1 bases, a hundred matches and twelve more
    insertions sum to 84 modulo 2^32 — below the hundred bases the record holds — while the run of matches sits at query offset
    2^30; the lengths add up in 64 bits on the device and the record is refused.  Likewise a reference span of 2^30 or more."""
    OPS = {"I": 1, "D": 2, "=": 7}
    big = 268435455
    good = [(100 << 4) | OPS["="]]
    wrap = [(big << 4) | OPS["I"]] * 4 + [(100 << 4) | OPS["="]] + [(big << 4) | OPS["I"]] * 12
    assert sum(w >> 4 for w in wrap) % (1 << 32) == 84
    span = [(50 << 4) | OPS["="]] + [(big << 4) | OPS["D"]] * 5 + [(50 << 4) | OPS["="]]
    seq50 = np.full(50, 0x12, dtype=np.uint8)

    def build(cigars):
        cig = np.array([w for c in cigars for w in c], dtype=np.uint32)
        co = np.cumsum([0] + [len(c) for c in cigars]).astype(np.uint64)
        n = len(cigars)
        return (np.zeros(n, dtype=np.int32), cig, co, np.tile(seq50, n), (50 * np.arange(n + 1)).astype(np.uint64))

    jl.ingest_records(120, 0, *build([good, good, good]))
    assert (msa.unpack_columns(jl.download_columns(), 3)[:, :100] < 4).all()
    with pytest.raises(capi.JulietError) as err:
        jl.ingest_records(120, 0, *build([good, good, wrap, good]))
    assert "record 2" in str(err.value) and "more bases" in str(err.value)
    with pytest.raises(capi.JulietError) as err:
        jl.ingest_records(120, 0, *build([good, span, wrap]))
    assert "record 1" in str(err.value) and "2^30" in str(err.value)
    # the same through a second step of 128 ops (the carries between steps)
    long_wrap = [(1 << 4) | OPS["="], (1 << 4) | OPS["I"]] * 70 + wrap
    with pytest.raises(capi.JulietError) as err:
        jl.ingest_records(120, 0, *build([good, long_wrap]))
    assert "record 1" in str(err.value) and "more bases" in str(err.value)


def test_device_ingest_dense_runs_take_the_slow_path(jl):
    """A deletion at every other column: hundreds of runs per read and sweep, far more than a workgroup's entry area holds
    in either size — most (read, sweep) pairs are written column by column (slow_pair: runs looked up in HBM, bits flipped
    with atomics behind the workgroup's own stores), the first few reads of every tile through the tile path; both with and
    without QV masking, and as windows of one upload."""
    n, l = 700, 1000
    rng = np.random.default_rng(5)
    sp = synth.SynthParams(seed=41, partial_rate=0.2, mask_rate=0.02, sub_rate=0.01)
    ref = synth.reference(sp.seed, l)
    rows = synth.rows(sp, l, 0, n, ref)
    dense = rows[:, 1::2]
    dense[dense < 6] = 4                        # every odd column of every covered stretch is a deletion
    rows[3] = 6
    pos, cigar, cig_off, seq4, seq_off, qual, qual_off = rows_to_records(rows, ref, rng)
    assert (cig_off[1:] - cig_off[:-1]).max() > 600
    jl.ingest_records(l, 0, pos, cigar, cig_off, seq4, seq_off)
    assert (msa.unpack_columns(jl.download_columns(), n) == rows).all()
    jl.ingest_records(l - 133, 77, pos, cigar, cig_off, seq4, seq_off, qual, qual_off, min_qv=94)
    exp = rows[:, 77:l - 56].copy()
    exp[exp < 4] = 5
    assert (msa.unpack_columns(jl.download_columns(), n) == exp).all()
    # two windows of ONE upload, built without waiting in between (jl_records_window_async), then a run behind each
    rec, w0, w1 = capi.Juliet(0), capi.Juliet(0), capi.Juliet(0)
    rec.records_upload(pos, cigar, cig_off, seq4, seq_off)
    w0.records_window(rec, 500, 0, wait=False)
    w1.records_window(rec, 502, 498, wait=False)
    assert (msa.unpack_columns(w0.download_columns(), n) == rows[:, :500]).all()
    assert (msa.unpack_columns(w1.download_columns(), n) == rows[:, 498:]).all()
    for c in (w1, w0, rec):
        c.close()


def test_device_ingest_long_cigar(jl):
    """One op per base: 5000 ops per read, every lane fetches a cigar word at every column."""
    l = 5000
    rng = np.random.default_rng(1)
    ref = synth.reference(3, l)
    rows = np.tile(ref, (3, 1)).astype(np.uint8)
    rows[:, ::2] = (rows[:, ::2] + 1) % 4       # alternate match / mismatch: one op per base => 5000 ops
    rows[1, 1000:1100] = 4
    pos, cigar, cig_off, seq4, seq_off, _, _ = rows_to_records(rows, ref, rng, with_noise_ops=False)
    assert cig_off[1] - cig_off[0] > 4000
    jl.ingest_records(l, 0, pos, cigar, cig_off, seq4, seq_off)
    assert (msa.unpack_columns(jl.download_columns(), 3) == rows).all()


# --------------------------------------------------------------------------------------------- cross-window phasing
def test_phase_across_windows_equals_unsharded(oracle):
    """the same reads span several windows (config 4/5 as one long reference).  Each window is called
    on its own context with the global Bonferroni factor; phasing then runs on the compact matrix of all variant
    columns and must equal the oracle's phasing of the unsharded matrix."""
    from minorseq_amd import sharding
    n, l = 7000, 900
    sp = synth.SynthParams(seed=29, minor_permille=(70, 60, 50, 40), partial_rate=0.15)
    ref = synth.reference(sp.seed, l)
    rows = synth.rows(sp, l, 0, n, ref)
    # move two of the planted edits into the last third so that haplotypes really span windows
    genes = np.array([(1, l + 1)], dtype=capi.GENE)
    rows[: n // 20, 700:703] = (rows[: n // 20, 700:703] + 1) % 4
    rows[n // 40: n // 16, 820:823] = (rows[n // 40: n // 16, 820:823] + 2) % 4
    full_v = oracle.call(rows, genes, refseq=ref)
    assert len(np.unique(full_v["col"] // 300)) >= 2                         # variants in more than one window
    exp = oracle.phase(rows, full_v)
    for world in (2, 3):
        wb = sharding.window_bounds(l, world)
        ctxs, tables = [], []
        prm = capi.default_params(n_tests=sharding.default_n_tests(genes))
        for b, e in wb:
            c = capi.Juliet(0)
            c.upload_columns(msa.pack_columns(rows[:, b:e]), n, win_begin=b)
            c.pileup_async(genes, ref)
            c.call_async(prm)
            tables.append(c.call_fetch())
            ctxs.append(c)
        merged = sharding.merge_tables(tables, [b for b, _ in wb])
        assert_variants_equal(merged, full_v)
        ph, pos_global = capi.phase_across_windows(ctxs, merged)
        assert (pos_global == exp["pos_cols"]).all()
        got = dict(ph, pos_cols=pos_global)
        assert_phase_equal(got, exp, len(full_v))
        for c in ctxs:
            c.close()


def test_phase_across_windows_rccl_single_rank(jl, oracle):
    """The RCCL form of the column exchange (owner broadcasts 3 columns per position) with world = 1."""
    import ctypes as C
    n, l = 4000, 300
    sp = synth.SynthParams(seed=37, minor_permille=(70, 60, 50, 40))
    ref = synth.reference(sp.seed, l)
    rows = synth.rows(sp, l, 0, n, ref)
    genes = np.array([(1, l + 1)], dtype=capi.GENE)
    jl.upload_columns(msa.pack_columns(rows), n)
    jl.pileup_async(genes, ref)
    jl.call_async(capi.default_params())
    table = jl.call_fetch()
    idbuf = np.zeros(128, dtype=np.uint8)
    assert jl.lib.jl_comm_unique_id(idbuf.ctypes.data_as(C.c_void_p)) == 0
    comm = C.c_void_p()
    jl._chk(jl.lib.jl_comm_create(jl.h, idbuf.ctypes.data_as(C.c_void_p), 0, 1, C.byref(comm)))
    try:
        ph, pos_global = capi.phase_across_windows([jl], table, comm=comm, win_begins=[0], win_ncols=[l])
    finally:
        jl.lib.jl_comm_destroy(comm)
    exp = oracle.phase(rows, table)
    assert (pos_global == exp["pos_cols"]).all()
    assert_phase_equal(dict(ph, pos_cols=pos_global), exp, len(table))


# --------------------------------------------------------------------------------------------- deep coverage
def test_ten_million_reads_against_oracle(oracle):
    """Config-5 depth (1e7 reads) on a narrow window: exercises the multi-flush path of the pileup kernel
    (611 tiles per column), 2^25-slot grouping table, p-values that underflow, u32 counts near 1e7."""
    n, l = 10_000_000, 60
    sp = synth.SynthParams(seed=5, minor_permille=(10, 10, 10, 10))
    ref = synth.reference(sp.seed, l)
    genes = np.array([(1, l + 1), (2, l)], dtype=capi.GENE)
    j = capi.Juliet(0)
    j.alloc(n, l)
    j.synth_fill(sp, ref)
    out = j.run(genes, ref)
    pf = j.pileup_fetch()
    rows = msa.unpack_columns(j.download_columns(), n)
    j.close()
    assert (pf["col_counts"].sum(axis=1) == n).all()
    assert (pf["col_counts"] == oracle.pileup(rows)).all()
    hist, cov = oracle.codon_hist(rows, pf["pos_col"])
    assert (pf["hist"] == hist).all() and (pf["coverage"] == cov).all()
    exp_v = oracle.call(rows, genes, refseq=ref)
    assert len(exp_v) >= 4
    assert_variants_equal(out["variants"], exp_v)
    assert (exp_v["p_value"] == 0.0).any() and np.isfinite(exp_v["log_p"]).all()     # underflow carried as log-p
    ph = oracle.phase(rows, exp_v)
    assert_phase_equal(out["phase"], ph, len(exp_v))
    s = out["phase"]["summary"]
    assert s["reported_reads"] + s["insufficient_reads"] + s["damaged_reads"] == n


# --------------------------------------------------------------------------------------------- randomized sweep
def test_random_shapes_and_plans_against_oracle(oracle):
    """Forty random (reads, columns, gene layout, parameter) combinations through jl_run_async vs the oracle."""
    rng = np.random.default_rng(int(os.environ.get("JL_TEST_SEED", "20260101")))     # (JL_TEST_SEED: a soak with other draws)
    j = capi.Juliet(0)
    for case in range(40):
        n = int(rng.integers(1, 4000))
        l = int(rng.integers(3, 260))
        sp = synth.SynthParams(seed=int(rng.integers(1, 1 << 30)), partial_rate=float(rng.choice([0.0, 0.3])),
                               mask_rate=float(rng.choice([0.0, 0.02, 0.1])), del_rate=float(rng.choice([0.0, 0.01])),
                               sub_rate=float(rng.choice([1e-4, 0.01])),
                               minor_permille=tuple(int(x) for x in rng.integers(0, 120, 4)))
        ref = synth.reference(sp.seed, l)
        rows = synth.rows(sp, l, 0, n, ref)
        ng = int(rng.integers(1, 4))
        genes = []
        for _ in range(ng):
            b = int(rng.integers(1, max(2, l - 2)))
            e = int(rng.integers(b + 1, l + 8))
            genes.append((b, e))
        genes = np.array(genes, dtype=capi.GENE)
        use_ref = bool(rng.integers(0, 2))
        win_b = int(rng.integers(0, max(1, l // 4)))
        win = rows[:, win_b:]
        prm = capi.default_params(alpha=float(rng.choice([0.01, 0.2])), n_tests=float(rng.choice([0.0, 1.0, 50.0])),
                                  chemistry=str(rng.choice(["sequel", "permissive"])),
                                  expected_round=int(rng.integers(0, 3)))
        min_reads = int(rng.choice([1, 3, 10]))
        j.upload_columns(msa.pack_columns(win), n, win_begin=win_b)
        out = j.run(genes, ref if use_ref else None, prm, phasing=True, min_reads=min_reads)
        exp_v = oracle.call(win, genes, win_begin=win_b, refseq=ref if use_ref else None, params=oracle_params(prm))
        assert_variants_equal(out["variants"], exp_v)
        assert_phase_equal(out["phase"], oracle.phase(win, exp_v, min_reads=min_reads), len(exp_v))
    j.close()


def test_variant_table_overflow_is_reported(oracle):
    """More called codons than the fixed-stride table (4096 rows): JL_ERR_OVERFLOW, the rows that fit are the
    first rows of the ordered table."""
    n, l = 2000, 3000
    sp = synth.SynthParams(seed=3, sub_rate=0.05, minor_permille=(10, 10, 10, 10))
    ref = synth.reference(sp.seed, l)
    rows = synth.rows(sp, l, 0, n, ref)
    genes = np.array([(1, l + 1)], dtype=capi.GENE)
    loose = capi.default_params(alpha=0.9, n_tests=1.0)
    exp = oracle.call(rows, genes, refseq=ref, params=oracle_params(loose))
    assert len(exp) > capi.VARIANT_CAP
    j = capi.Juliet(0)
    j.upload_columns(msa.pack_columns(rows), n)
    j.pileup_async(genes, ref)
    j.call_async(loose)
    out = np.zeros(capi.VARIANT_CAP, dtype=capi.VARIANT)
    import ctypes as C
    cnt = C.c_uint32()
    rc = j.lib.jl_call_fetch(j.h, out.ctypes.data_as(C.c_void_p), capi.VARIANT_CAP, C.byref(cnt))
    assert rc == -5 and cnt.value == len(exp)
    assert_variants_equal(out, exp[: capi.VARIANT_CAP])
    # the whole-path entry point reports the same condition
    with pytest.raises(capi.JulietError) as e:
        j.run(genes, ref, loose, phasing=False)
    assert e.value.status == -5
    j.close()


def test_fused_grouping_with_thousands_of_distinct_patterns(jl, oracle):
    """Vp = 10 (single-word keys, fused kernel) on reads so noisy that a 2048-read block holds more distinct
    patterns than its 1024-slot LDS table: the overflow path inserts straight into the global table, and the
    global table itself sees long probe sequences.  Grouping must stay exact."""
    n, l = 9000, 60
    sp = synth.SynthParams(seed=77, sub_rate=0.12, del_rate=0.0, mask_rate=0.0, minor_permille=(100, 100, 100, 100))
    ref = synth.reference(sp.seed, l)
    rows = synth.rows(sp, l, 0, n, ref)
    genes = np.array([(1, l + 1)], dtype=capi.GENE)
    loose = capi.default_params(alpha=0.9, n_tests=1.0)
    jl.upload_columns(msa.pack_columns(rows), n)
    jl.pileup_async(genes, ref)
    jl.call_async(loose)
    table = jl.call_fetch()
    cols = np.unique(table["col"])[:10]
    sub = table[np.isin(table["col"], cols)]
    assert len(cols) == 10
    for min_reads in (1, 2, 10):
        jl.phase_async(sub, min_reads=min_reads)
        try:
            got = jl.phase_fetch(cap_var=len(sub))
        except capi.JulietError as e:
            # more than 4096 groups reach min_reads = 1: documented capacity of the selector
            assert e.status == -5 and min_reads == 1
            continue
        exp = oracle.phase(rows, sub, min_reads=min_reads)
        assert exp["summary"]["n_positions"] == 10
        assert_phase_equal(got, exp, len(sub))
    exp1 = oracle.phase(rows, sub, min_reads=1)
    assert exp1["summary"]["reported_reads"] + exp1["summary"]["insufficient_reads"] == n   # no damaged reads here
    # distinct patterns per 2048-read block exceed the LDS table
    pat = np.stack([rows[:2048, c] * 16 + rows[:2048, c + 1] * 4 + rows[:2048, c + 2] for c in cols], axis=1)
    assert len(np.unique(pat, axis=0)) > 1024


# --------------------------------------------------------------------------------------------- chunk table
@pytest.mark.parametrize("n,l", [(5000, 700), (8200, 301), (300, 40)])
def test_pileup_genes_in_different_frames(jl, oracle, n, l):
    """HIV-like layout: consecutive genes in different reading frames plus a short overlap (p6/protease style,
    SURVEY A.5).  The chunk table keeps every single-frame stretch halo-free; results must not depend on it."""
    sp = synth.SynthParams(seed=n + l, partial_rate=0.2, mask_rate=0.03, del_rate=0.01, sub_rate=0.01,
                           minor_permille=(60, 50, 40, 30))
    ref = synth.reference(sp.seed, l)
    rows = synth.rows(sp, l, 0, n, ref)
    third = l // 3
    def begin_in_frame(near, frame):          # smallest 1-based begin >= near whose 0-based start is in `frame`
        b = near
        while (b - 1) % 3 != frame:
            b += 1
        return b
    genes = np.array([(begin_in_frame(2, 1), third + 1), (begin_in_frame(third + 2, 2), 2 * third + 2),
                      (begin_in_frame(max(1, 2 * third - 20), 0), l - 1), (5, 9), (l - 4, l + 5)], dtype=capi.GENE)
    assert len({(int(g["begin"]) - 1) % 3 for g in genes[:3]}) >= 2          # really different frames
    jl.upload_columns(msa.pack_columns(rows), n)
    for refseq in (ref, None):
        jl.pileup_async(genes, refseq)
        got = jl.pileup_fetch()
        assert (got["col_counts"] == oracle.pileup(rows)).all()
        hist, cov = oracle.codon_hist(rows, got["pos_col"])
        assert (got["hist"] == hist).all() and (got["coverage"] == cov).all()
    out = jl.run(genes, ref, capi.default_params(alpha=0.3, n_tests=1.0))
    exp_v = oracle.call(rows, genes, refseq=ref, params=oracle_lib.default_params(alpha=0.3, n_tests=1.0))
    assert_variants_equal(out["variants"], exp_v)
    assert_phase_equal(out["phase"], oracle.phase(rows, exp_v), len(exp_v))


def test_group_run_with_a_window_without_variants(oracle):
    """One window of the group carries no variant at all (every read is the reference) and one is empty of coverage
    in half of its columns: the grouped call / phase / id launches must leave exactly the oracle's (empty) answer for
    them and the full answer for their neighbour."""
    l = 240
    genes = np.array([(1, l + 1)], dtype=capi.GENE)
    ref = synth.reference(33, l)
    n = 3000
    rows_ref = np.tile(ref, (n, 1)).astype(np.uint8)                       # no variant
    rows_mix = synth.rows(synth.SynthParams(seed=34, minor_permille=(60, 50, 40, 30)), l, 0, n, ref)
    rows_half = rows_mix.copy()
    rows_half[:, l // 2:] = 6                                              # right half not covered by any read
    ctxs, rows_all = [], [rows_ref, rows_mix, rows_half]
    for rows in rows_all:
        j = capi.Juliet(0)
        j.upload_columns(msa.pack_columns(rows), n)
        j.sync()
        ctxs.append(j)
    grp = capi.Group(ctxs)
    try:
        for rep in range(2):
            grp.run_async(genes, ref, capi.default_params(), True, 10, True)
        for k, (j, rows) in enumerate(zip(ctxs, rows_all)):
            exp_v = oracle.call(rows, genes, refseq=ref)
            v = j.run_view()
            assert v is not None
            assert_variants_equal(v["variants"], exp_v)
            assert_phase_equal(v["phase"], oracle.phase(rows, exp_v), len(exp_v))
            if k == 0:
                assert len(exp_v) == 0 and v["phase"]["summary"]["n_haplotypes"] == 0
    finally:
        grp.close()
        for j in ctxs:
            j.close()


def test_group_run_with_genes_in_different_frames(oracle):
    """The grouped launches on the HIV-like layout above: chunks with and without halo columns in one pileup launch
    (the general path of pileup_group_kernel), three windows of different depth, majority-codon mode included."""
    l = 420
    third = l // 3
    def begin_in_frame(near, frame):
        b = near
        while (b - 1) % 3 != frame:
            b += 1
        return b
    genes = np.array([(begin_in_frame(2, 1), third + 1), (begin_in_frame(third + 2, 2), 2 * third + 2),
                      (begin_in_frame(max(1, 2 * third - 20), 0), l - 1), (5, 9), (l - 4, l + 5)], dtype=capi.GENE)
    assert len({(int(g["begin"]) - 1) % 3 for g in genes[:3]}) >= 2
    ref = synth.reference(515, l)
    prm = capi.default_params()
    ctxs, rows_all = [], []
    for k, n in enumerate((2500, 7000, 4100)):
        sp = synth.SynthParams(seed=515 + k, partial_rate=0.1 * k, mask_rate=0.03, del_rate=0.01,
                               minor_permille=(60, 50, 40, 30))
        rows = synth.rows(sp, l, 0, n, ref)
        j = capi.Juliet(0)
        j.upload_columns(msa.pack_columns(rows), n)
        j.sync()
        ctxs.append(j)
        rows_all.append(rows)
    grp = capi.Group(ctxs)
    try:
        for refseq in (ref, None):
            for rep in range(2):   # the second launch replays the captured graph
                grp.run_async(genes, refseq, prm, True, 10, True)
            for j, rows in zip(ctxs, rows_all):
                exp_v = oracle.call(rows, genes, refseq=refseq)
                assert 0 < len(np.unique(exp_v["col"])) <= 10
                v = j.run_view()
                assert v is not None
                assert_variants_equal(v["variants"], exp_v)
                assert_phase_equal(v["phase"], oracle.phase(rows, exp_v), len(exp_v))
                pf = j.pileup_fetch()
                assert (pf["col_counts"] == oracle.pileup(rows)).all()
                hist, cov = oracle.codon_hist(rows, pf["pos_col"])
                assert (pf["hist"] == hist).all() and (pf["coverage"] == cov).all()
    finally:
        grp.close()
        for j in ctxs:
            j.close()


# --------------------------------------------------------------------------------------------- torch plumbing
def test_adopted_torch_tensor_and_caller_stream(oracle):
    """PyTorch as plumbing only: the resident matrix lives in a torch tensor (jl_msa_adopt) and all work of the
    context is enqueued on a torch stream the caller owns."""
    import torch
    n, l = 5300, 300
    sp = synth.SynthParams(seed=19, minor_permille=(70, 60, 50, 40), partial_rate=0.1)
    ref = synth.reference(sp.seed, l)
    rows = synth.rows(sp, l, 0, n, ref)
    # the resident format itself (three bit planes per column), with a caller-chosen stride: any multiple of 16 bytes
    # that holds the reads (the library's own matrices use whole 128-byte lines)
    stride = (n + 7) // 8 + 15 & ~15
    assert stride != msa.plane_stride(n)
    packed = msa.pack_planes(rows, stride)
    assert (msa.unpack_planes(packed, n) == rows).all()
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        t = torch.from_numpy(packed).cuda(non_blocking=False)
    stream.synchronize()
    j = capi.Juliet(0, stream=stream.cuda_stream)
    j.adopt(t.data_ptr(), n, l, stride, keep_alive=t)
    assert (msa.unpack_columns(j.download_columns(), n) == rows).all()   # (the interchange format out of an adopted matrix)
    genes = np.array([(1, l + 1)], dtype=capi.GENE)
    out = j.run(genes, ref)
    exp_v = oracle.call(rows, genes, refseq=ref)
    assert_variants_equal(out["variants"], exp_v)
    assert_phase_equal(out["phase"], oracle.phase(rows, exp_v), len(exp_v))
    # work really was ordered on the caller's stream: a torch op enqueued behind it sees the finished table
    j.run_async(genes, ref)
    rows_ptr, cnt_ptr, cap = j.variant_table_device()
    assert cap == capi.VARIANT_CAP and rows_ptr and cnt_ptr
    stream.synchronize()
    # The caller owns an adopted matrix and may rewrite it at any time: the library keeps no derived copy of anything, so
    # new reads in the same buffer give the new answer.  And an adopted matrix (caller's stride) counts like the same
    # reads uploaded through the interchange format (the library's stride).
    rows2 = synth.rows(synth.SynthParams(seed=23, minor_permille=(90, 30, 20, 10), partial_rate=0.3), l, 0, n, ref)
    with torch.cuda.stream(stream):
        t.copy_(torch.from_numpy(msa.pack_planes(rows2, stride)))
    stream.synchronize()
    out2 = j.run(genes, ref)
    exp2 = oracle.call(rows2, genes, refseq=ref)
    assert_variants_equal(out2["variants"], exp2)
    assert_phase_equal(out2["phase"], oracle.phase(rows2, exp2), len(exp2))
    j.pileup_async(genes, ref)
    from_adopted = j.pileup_fetch()
    k = capi.Juliet(0)
    k.upload_columns(msa.pack_columns(rows2), n)
    k.pileup_async(genes, ref)
    from_upload = k.pileup_fetch()
    for key in from_adopted:
        assert (from_adopted[key] == from_upload[key]).all(), key
    k.close()
    j.close()
    del t
