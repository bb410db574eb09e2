"""Oracle vs the known-answer fragments of the reference's screenshots and the
planted truth of the synthetic mixture.  CPU only."""
import numpy as np
import pytest

import oracle_lib
from minorseq_amd import msa, synth


def _a1_matrix():
    """2998 reads x 9 columns reproducing the K65 context table of doc/img/juliet_hiv-context.png (A.1)."""
    n = 2998
    m = np.zeros((n, 9), dtype=np.uint8)  # all 'A'
    m[:, 2] = msa.SYM_G                    # rel -1: 2952 G
    # rel -3 .. -1 (columns 0..2)
    m[2947:, 0] = msa.SYM_MASK             # 51 N
    m[2923:2925, 1] = msa.SYM_G            # 2 G
    m[2925:, 1] = msa.SYM_MASK             # 73 N
    m[2952:2956, 2] = msa.SYM_A            # 4 A
    m[2956:, 2] = msa.SYM_MASK             # 42 N
    # codon columns 3,4,5 (rel 0,1,2)
    m[0:339, 3] = msa.SYM_GAP              # 339 '-'
    m[339:392, 3] = msa.SYM_MASK           # 53 N
    m[392:456, 4] = msa.SYM_MASK           # 64 N
    m[456:469, 5] = msa.SYM_MASK           # 60 N: 13 new reads ...
    m[0:47, 5] = msa.SYM_MASK              # ... + 47 that are already excluded  => union 469
    m[469:498, 4] = msa.SYM_G              # 29 reads AGA (K65R)
    # rel 3,4,5
    m[0:60, 6] = msa.SYM_MASK
    m[0:56, 7] = msa.SYM_MASK
    m[0:247, 8] = msa.SYM_MASK
    return m


def test_a1_context_table(oracle):
    m = _a1_matrix()
    col = oracle.pileup(m)
    expect = np.array([[2947, 0, 0, 0, 0, 51], [2923, 0, 2, 0, 0, 73], [4, 0, 2952, 0, 0, 42],
                       [2606, 0, 0, 0, 339, 53], [2905, 0, 29, 0, 0, 64], [2938, 0, 0, 0, 0, 60],
                       [2938, 0, 0, 0, 0, 60], [2942, 0, 0, 0, 0, 56], [2751, 0, 0, 0, 0, 247]], dtype=np.uint32)
    assert (col == expect).all()
    assert (col.sum(axis=1) == 2998).all()          # every row of the context table sums to 2998
    hist, cov = oracle.codon_hist(m, [3])
    assert cov[0] == 2529                            # coverage printed next to K65R
    assert cov[0] <= col[3:6, :4].sum(axis=1).min()  # <= min real-base depth (2606)
    assert hist[0, msa.codon_index("AGA")] == 29
    assert hist[0, msa.codon_index("AAA")] == 2500
    assert f"{100.0 * 29 / 2529:.2g}" == "1.1"       # frequency shown with 2 significant digits


def test_a1_call_k65r(oracle):
    """With the example config semantics (ref AAA), AGA at 29/2529 is called; nothing else is."""
    m = _a1_matrix()
    genes = np.array([(4, 7)], dtype=oracle_lib.GENE)  # one codon at columns 3..5, 1-based [4,7)
    ref = np.zeros(9, dtype=np.uint8)
    ref[2] = 2
    v = oracle.call(m, genes, refseq=ref, params=oracle_lib.default_params(n_tests=1000))
    assert len(v) == 1
    assert v[0]["codon"] == msa.codon_index("AGA") and v[0]["ref_codon"] == 0
    assert v[0]["count"] == 29 and v[0]["coverage"] == 2529 and v[0]["expected"] == 1
    assert v[0]["codon_pos"] == 1 and v[0]["col"] == 3
    # majority mode gives the same call (J:133-134)
    v2 = oracle.call(m, genes, params=oracle_lib.default_params(n_tests=1000))
    assert (v2 == v).all()


def test_reference_vs_majority_mode(oracle):
    """A.2 juliet_hiv-own.png: against a supplied reference, a 99 % codon IS a variant (S3S AGC->AGT)."""
    n = 3000
    m = np.zeros((n, 3), dtype=np.uint8)
    m[:, 0], m[:, 1], m[:, 2] = 0, 2, 3          # AGT in 99 %
    m[:30, 2] = 1                                 # AGC in 1 %
    genes = np.array([(1, 4)], dtype=oracle_lib.GENE)
    prm = oracle_lib.default_params(n_tests=1000)
    ref = np.array([0, 2, 1], dtype=np.uint8)     # reference AGC
    v = oracle.call(m, genes, refseq=ref, params=prm)
    assert len(v) == 1 and v[0]["codon"] == msa.codon_index("AGT") and v[0]["count"] == 2970
    assert v[0]["p_value"] == 0.0 or v[0]["p_value"] < 1e-300
    assert v[0]["log_p"] < -1000                   # log-p survives the underflow
    v = oracle.call(m, genes, params=prm)          # majority mode: AGT is the reference, AGC the 1 % minor
    assert len(v) == 1 and v[0]["codon"] == msa.codon_index("AGC") and v[0]["ref_codon"] == msa.codon_index("AGT")


def test_gene_frames_and_window(oracle):
    sp = synth.SynthParams(seed=7)
    rows = synth.rows(sp, 300, 0, 400)
    # overlapping genes in different frames are independent (J:261-264); window offset shifts columns
    genes = np.array([(101, 161), (102, 165), (1, 50)], dtype=oracle_lib.GENE)
    full = oracle.call(rows, genes, win_begin=0, params=oracle_lib.default_params(alpha=0.5, n_tests=1))
    sub = oracle.call(rows[:, 90:200], genes, win_begin=90, params=oracle_lib.default_params(alpha=0.5, n_tests=1))
    keep = full[full["gene"] != 2]
    assert len(sub) == len(keep)
    for a, b in zip(sub, keep):
        assert a["col"] + 90 == b["col"]
        for k in ("gene", "codon_pos", "codon", "ref_codon", "count", "coverage", "p_value"):
            assert a[k] == b[k]


@pytest.fixture(scope="module")
def c1():
    """BASELINE.json configs[0]: 1k synthetic CCS reads x 3 kb."""
    sp = synth.SynthParams(seed=1)
    ref = synth.reference(sp.seed, 3000)
    return sp, ref, synth.rows(sp, 3000, 0, 1000, ref)


def test_c1_calls_planted_variants(oracle, c1):
    sp, ref, rows = c1
    plan = synth.make_plan(sp, 3000, ref)
    genes = np.array([(1, 3001)], dtype=oracle_lib.GENE)
    # 1k reads cannot carry 1 % minors at the default threshold (J:233-237: 2500x minimal) -> raise the minors
    sp5 = synth.SynthParams(seed=1, minor_permille=(60, 60, 60, 60))
    rows5 = synth.rows(sp5, 3000, 0, 1000, ref)
    v = oracle.call(rows5, genes, refseq=ref)
    called = {(int(r["col"]) // 3, int(r["codon"])) for r in v}
    for k in range(synth.N_EDITS):
        c = plan.edit_col[k]
        cod = [int(x) for x in ref[c - c % 3: c - c % 3 + 3]]
        cod[c % 3] = plan.edit_base[k]
        assert (c // 3, 16 * cod[0] + 4 * cod[1] + cod[2]) in called
    assert len(v) == 5
    ph = oracle.phase(rows5, v, min_reads=10)
    s = ph["summary"]
    assert s["reported_reads"] + s["insufficient_reads"] + s["damaged_reads"] == 1000   # J:378-379, A.4
    assert s["n_positions"] == 5 and s["n_haplotypes"] == 5
    assert (np.diff(ph["hap_count"].astype(np.int64)) <= 0).all()                       # descending (A.3)
    assert ph["hit"][:, 0].sum() == 0                                                    # A = wild type
    assert sorted(ph["hit"].sum(axis=0).tolist()) == [0, 1, 1, 1, 2]                     # one haplotype co-carries two
    both = [h for h in range(5) if ph["hit"][:, h].sum() == 2][0]
    pair = np.nonzero(ph["hit"][:, both])[0]
    assert ph["cooc"][pair[0], pair[1]] == ph["hap_count"][both]
    pct = 100.0 * ph["hap_count"] / s["reported_reads"]
    assert abs(pct.sum() - 100.0) < 1e-9                                                 # A.3 sums to 100
    # read_hap agrees with the counts
    for h in range(5):
        assert (ph["read_hap"] == h).sum() == ph["hap_count"][h]
    assert (ph["read_hap"] == oracle_lib.HAP_DAMAGED).sum() == s["damaged_reads"]
    assert (ph["read_hap"] == oracle_lib.HAP_INSUFFICIENT).sum() == s["insufficient_reads"]


def test_category_arithmetic_a4(oracle):
    """A.4: the three exclusive categories sum to the read count; marginals overlap."""
    sp = synth.SynthParams(seed=11, partial_rate=0.2, mask_rate=0.05, del_rate=0.02, minor_permille=(80, 80, 80, 80))
    ref = synth.reference(sp.seed, 600)
    rows = synth.rows(sp, 600, 0, 1500, ref)
    genes = np.array([(1, 601)], dtype=oracle_lib.GENE)
    v = oracle.call(rows, genes, refseq=ref)
    assert len(v) >= 3
    ph = oracle.phase(rows, v)
    s = ph["summary"]
    assert s["reported_reads"] + s["insufficient_reads"] + s["damaged_reads"] == 1500
    assert s["marginal_gap"] + s["marginal_heteroduplex"] + s["marginal_partial"] >= s["damaged_reads"]
    assert max(s["marginal_gap"], s["marginal_heteroduplex"], s["marginal_partial"]) <= s["damaged_reads"]
    assert s["marginal_partial"] > 0 and s["marginal_gap"] > 0 and s["marginal_heteroduplex"] > 0


def test_permutation_invariance(oracle, c1):
    """Permuting reads never changes counts, calls or haplotypes (SURVEY §4 property)."""
    sp, ref, _ = c1
    sp5 = synth.SynthParams(seed=3, minor_permille=(60, 60, 60, 60))
    rows = synth.rows(sp5, 3000, 0, 800, ref)
    perm = np.random.default_rng(5).permutation(len(rows))
    genes = np.array([(1, 3001)], dtype=oracle_lib.GENE)
    v1, v2 = oracle.call(rows, genes, refseq=ref), oracle.call(rows[perm], genes, refseq=ref)
    assert (v1 == v2).all()
    p1, p2 = oracle.phase(rows, v1), oracle.phase(rows[perm], v2)
    assert p1["summary"] == p2["summary"]
    assert (p1["hap_count"] == p2["hap_count"]).all() and (p1["hap_pattern"] == p2["hap_pattern"]).all()
    assert (p1["read_hap"][perm] == p2["read_hap"]).all()


def test_min_reads_threshold(oracle):
    """J:253-254: a haplotype needs >= 10 reads to be reported."""
    n = 200
    m = np.zeros((n, 3), dtype=np.uint8)
    m[:10, 0] = 1   # CAA x10
    m[10:19, 0] = 2  # GAA x9
    var = np.zeros(2, dtype=oracle_lib.VARIANT)
    var["col"] = 0
    var["codon"] = [msa.codon_index("CAA"), msa.codon_index("GAA")]
    ph = oracle.phase(m, var, min_reads=10)
    assert ph["summary"]["n_haplotypes"] == 2            # wild type (181) + CAA (10)
    assert ph["hap_count"].tolist() == [181, 10]
    assert ph["summary"]["insufficient_reads"] == 9
    assert ph["hit"].tolist() == [[0, 1], [0, 0]]


def test_pack_roundtrip():
    rng = np.random.default_rng(0)
    for n in (1, 2, 255, 256, 257, 1000):
        rows = rng.integers(0, 7, size=(n, 17), dtype=np.uint8)
        p = msa.pack_columns(rows)
        assert p.shape == (17, msa.col_stride(n)) and p.shape[1] % 128 == 0
        assert (msa.unpack_columns(p, n) == rows).all()
        # pad nibbles are 6 (uncovered)
        full = msa.unpack_columns(p, p.shape[1] * 2)
        assert (full[n:] == 6).all()


def test_openmp_sweeps_give_identical_results(oracle, c1):
    """The all-cores CPU baseline (OpenMP over reads) is the same restatement: identical counts and calls."""
    sp, ref, rows = c1
    genes = np.array([(1, 3001), (2, 2999)], dtype=oracle_lib.GENE)
    a = oracle.call(rows, genes, refseq=ref, params=oracle_lib.default_params(alpha=0.3, n_tests=1))
    col_a = oracle.pileup(rows)
    oracle.set_threads(min(4, oracle.max_threads()))
    try:
        b = oracle.call(rows, genes, refseq=ref, params=oracle_lib.default_params(alpha=0.3, n_tests=1))
        col_b = oracle.pileup(rows)
    finally:
        oracle.set_threads(1)
    assert (a == b).all() and (col_a == col_b).all() and len(a) > 0


def test_appendix_a_fixture_relationships():
    """The transcribed screenshot numbers themselves satisfy the relationships the SPEC is built on."""
    import json
    import os
    a = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "appendix_a.json")))
    rows = a["a1_context_k65"]["rows"]
    assert all(sum(r) == 2998 for r in rows.values())                       # one symbol per read and column
    real = [sum(rows[k][:4]) for k in ("0", "1", "2")]
    bad = [sum(rows[k][4:]) for k in ("0", "1", "2")]
    assert 2998 - sum(bad) <= a["a1_context_k65"]["coverage"] <= min(real)  # coverage = reads with three real bases
    assert f"{100.0 * rows['1'][2] / a['a1_context_k65']['coverage']:.2g}" == a["a1_context_k65"]["percent_shown"]
    assert abs(sum(a["a3_phasing_percent"]) - 100.0) < 1e-9                 # haplotype percentages sum to 100
    w = a["a3_weakest_variant"]
    assert round(w["percent"] / 100 * w["coverage"]) == 21                   # the SPEC §5 threshold anchor
    c = a["a4_categories"]
    assert c["reported"] + c["insufficient"] + c["damaged"] == c["total"]   # doc/JULIET.md:378-379
    assert c["marginal_gaps"] + c["marginal_heteroduplexes"] + c["marginal_partial"] > c["damaged"]   # marginals overlap


def test_fuse_consensus_rule(oracle):
    """doc/FUSE.md:17-20 as docs/SPEC.md §11 fixes it: majority base per column, majority-deletion columns removed,
    an IN-FRAME insertion carried by more than half of the covering reads is included, out-of-frame and minority ones
    are not, and of two majority insertions closer than the minimal distance only the first is."""
    l, n = 40, 1000
    col = np.zeros((l, 6), dtype=np.uint32)
    ref = (np.arange(l) * 7 + 3) % 4
    col[np.arange(l), ref] = n
    col[5] = [100, 0, 0, 0, 900, 0]              # majority deletion: the column disappears
    col[30] = 0                                   # nobody covers it: N
    lh = np.zeros((l, 32), dtype=np.uint32)
    bc = np.zeros((l, 30, 4), dtype=np.uint32)

    def plant(c, length, count, seq):
        lh[c, min(length, 31)] += count
        for j, b in enumerate(seq[:30]):
            bc[c, j, "ACGT".index(b)] += count

    plant(8, 3, 700, "GGT")                       # included
    plant(12, 6, 400, "ACGTAC")                   # minority: no
    plant(14, 4, 900, "TTTT")                     # out of frame: no
    plant(15, 3, 900, "CCC")                      # 7 columns after the insertion at 8: too close at distance 10 ...
    plant(20, 9, 800, "AAACCCGGG")                # ... 12 columns after: included
    plant(20, 3, 100, "TTT")                      # a rarer length at the same place does not win, its bases do not either
    def expect(included):
        return "".join(included.get(c, "") + ("" if c == 5 else "N" if c == 30 else "ACGT"[ref[c]]) for c in range(l))

    assert oracle.fuse(col, lh, bc, 0.5, 10) == expect({8: "GGT", 20: "AAACCCGGG"})
    # at distance 7 the insertion at 15 is far enough from the one at 8 — and the one at 20 then too close to it
    assert oracle.fuse(col, lh, bc, 0.5, 7) == expect({8: "GGT", 15: "CCC"})
    assert oracle.fuse(col, lh, bc, 0.5, 5) == expect({8: "GGT", 15: "CCC", 20: "AAACCCGGG"})
    assert oracle.fuse(col, None, None) == "".join("" if c == 5 else "N" if c == 30 else "ACGT"[ref[c]] for c in range(l))
    assert oracle.fuse(col, lh, bc, 0.95, 1) == oracle.fuse(col, None, None)           # nothing reaches 95 %


def test_two_formulations_of_the_consensus_agree(oracle):
    """orc_fuse decides from counters (what the front end's fuse.hpp does with the device's), orc_fuse_records from explicit
    insertion records and a sweep over the rows.  Random records with planted insertions: an in-frame insertion in most
    reads, and at the SAME column an out-of-frame one with other bases in a fifth of them — it must not vote on the bases."""
    rng = np.random.default_rng(21)
    n, l = 600, 90
    ref = rng.integers(0, 4, l).astype(np.uint8)
    rows = np.tile(ref, (n, 1))
    noise = rng.random((n, l))
    rows[noise < 0.02] = 4                                   # deletions
    rows[(noise >= 0.02) & (noise < 0.03)] = 5               # masked bases
    code = [1, 2, 4, 8]                                      # BAM nt16 of A C G T
    pos = np.zeros(n, dtype=np.int32)
    cig, co, s4, so = [], [0], [], [0]
    plan = {30: ("ACGTTG", 0.7, "TTAC", 0.2), 60: ("GGA", 0.6, None, 0.0), 66: ("CCC", 0.9, None, 0.0), 12: ("AC", 0.8, None, 0.0)}
    for r in range(n):
        ops, bases, c = [], [], 0
        u = rng.random(len(plan) * 2)
        k = 0
        while c < l:
            if c in plan:
                inf, pf, outf, po = plan[c]
                x = u[k]; k += 1
                ins = inf if x < pf else (outf if outf and x < pf + po else None)
                if ins:
                    ops.append((len(ins) << 4) | 1)
                    bases += [code["ACGT".index(b)] for b in ins]
            sym = int(rows[r, c])
            if sym == 4:
                ops.append((1 << 4) | 2)                     # D
            else:
                ops.append((1 << 4) | 7)                     # =
                bases.append(code[ref[c]] if sym < 4 else 15)   # N where masked: still a query base
                if sym == 5:
                    rows[r, c] = 5
            c += 1
        cig += ops
        co.append(len(cig))
        if len(bases) % 2:
            bases.append(0)
        s4 += [(bases[i] << 4) | bases[i + 1] for i in range(0, len(bases), 2)]
        so.append(len(s4))
    cigar, cig_off = np.array(cig, dtype=np.uint32), np.array(co, dtype=np.uint64)
    seq4, seq_off = np.array(s4, dtype=np.uint8), np.array(so, dtype=np.uint64)
    col = oracle.pileup(rows)
    lh, bc = oracle.insertions(l, 0, pos, cigar, cig_off, seq4, seq_off)
    assert bc[30].sum() == lh[30, 6] * 6 and lh[30, 4] > 50          # the 4-base insertion is counted by length, not by base
    for frac, dist in ((0.5, 10), (0.5, 5), (0.1, 1)):
        a = oracle.fuse(col, lh, bc, frac, dist)
        b = oracle.fuse_records(rows, 0, pos, cigar, cig_off, seq4, seq_off, frac, dist)
        assert a == b
    a = oracle.fuse(col, lh, bc, 0.5, 5)
    want = "".join(("ACGTTG" if c == 30 else "GGA" if c == 60 else "CCC" if c == 66 else "") + "ACGT"[ref[c]] for c in range(l))
    assert a == want                                                   # in frame and in most reads; the 2-base one never
    assert "CCC" + "ACGT"[ref[66]] not in oracle.fuse(col, lh, bc, 0.5, 10)[60:80]     # 6 columns behind the one at 60: too close at distance 10


def test_all_cores_form_equals_the_plain_restatement(oracle):
    """bench.py's cpu_baseline_all_cores leg: OpenMP over columns / codon positions of a registered by-column copy (and over
    reads for the patterns of the phasing stage) gives exactly what the single-threaded loops give; so does the older
    split over reads that is used when no copy is registered."""
    from minorseq_amd import capi, synth
    n, l = 3000, 240
    sp = synth.SynthParams(seed=5, minor_permille=(70, 60, 50, 40), partial_rate=0.2)
    ref = synth.reference(sp.seed, l)
    rows = synth.rows(sp, l, 0, n, ref)
    genes = np.array([(1, l + 1), (20, 200)], dtype=capi.GENE)
    want_v = oracle.call(rows, genes, refseq=ref)
    want_c = oracle.pileup(rows)
    want_p = oracle.phase(rows, want_v)
    assert len(want_v) >= 4
    try:
        for with_cols in (True, False):
            oracle.set_threads(4)
            oracle.set_columns(rows if with_cols else None)
            v = oracle.call(rows, genes, refseq=ref)
            assert v.tobytes() == want_v.tobytes()
            assert (oracle.pileup(rows) == want_c).all()
            p = oracle.phase(rows, v)
            assert p["summary"] == want_p["summary"] and (p["read_hap"] == want_p["read_hap"]).all()
            assert (p["hap_count"] == want_p["hap_count"]).all() and (p["hit"] == want_p["hit"]).all()
            m = oracle.call(rows, genes)      # majority mode as well
            oracle.set_threads(1)
            oracle.set_columns(None)
            assert m.tobytes() == oracle.call(rows, genes).tobytes()
    finally:
        oracle.set_threads(1)
        oracle.set_columns(None)
