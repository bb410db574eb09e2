"""The oracle against EVERY known answer the reference holds for this path: the 66 variant rows printed on the
screenshots of real juliet output (tests/golden/appendix_a.json, transcribed from /root/reference/doc/img/*.png) and
the three phasing scenarios of the FAQ (doc/JULIET.md:278-288, 356-366).  CPU only; tests/test_gpu_golden.py sends
the same matrices through the HIP path."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib
import scenarios as sc
from minorseq_amd import msa

HERE = os.path.dirname(os.path.abspath(__file__))
FX = sc.load_fixture()
# docs/SPEC.md §5: the defaults call every printed row for any Bonferroni factor in this range (the HIV config's own
# factor is not recoverable from the reference: its gene list is cut off on juliet_target.png)
N_TESTS = (982.0, 1884.0)


@pytest.fixture(scope="module")
def fmt(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("fmt") / "libformat_shim.so")
    subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-o", out, os.path.join(HERE, "csrc", "format_shim.cpp")])
    lib = C.CDLL(out)
    lib.shim_format_percent.argtypes = [C.c_double, C.c_char_p]
    lib.shim_format_hap_percent.argtypes = [C.c_double, C.c_char_p]

    def call(fn, x):
        buf = C.create_string_buffer(64)
        fn(float(x), buf)
        return buf.value.decode()
    return (lambda x: call(lib.shim_format_percent, x)), (lambda x: call(lib.shim_format_hap_percent, x))


def all_rows():
    for name, t in FX["tables"].items():
        for g in t["genes"]:
            for r in g["rows"]:
                yield name, g["name"], r


def test_display_rule_is_pinned_by_the_printed_rows(fmt):
    """Two significant digits, truncated: reproduces all 66 printed percentages at their printed coverage; rounding
    to nearest cannot produce 12 of them.  The front end's formatter (host/format.hpp) and the test mirror agree."""
    fmt_percent, fmt_hap = fmt
    impossible_by_rounding = 0
    n = 0
    for name, gene, r in all_rows():
        pct, cov = r[5], r[6]
        c = sc.count_for(pct, cov)                       # raises when no count displays as printed
        assert fmt_percent(100.0 * c / cov) == pct == sc.fmt_percent(100.0 * c / cov), (name, gene, r)
        lo, hi = max(1, c - 60), min(cov, c + 60)
        if not any(f"{float(f'{100.0 * k / cov:.2g}'):g}" == pct for k in range(lo, hi + 1)):
            impossible_by_rounding += 1
        n += 1
    assert n == 66 and impossible_by_rounding == 12, (n, impossible_by_rounding)
    # haplotype percentages: one decimal, rounded (printed columns sum to 100.0); 27 reads print as 1.2 %
    t = FX["a4_perc_tooltip"]
    assert any(fmt_hap(100.0 * t["reads"] / d) == t["percent_shown"] for d in range(2160, 2349))
    for key in ("hiv_phasing", "major_after"):
        assert abs(sum(float(x) for x in FX["tables"][key]["haplotype_percent"]) - 100.0) < 1e-9
    for x, s in ((92.52, "92.5"), (1.04, "1"), (0.696, "0.7"), (100.0, "100"), (95.78, "95.8")):
        assert fmt_hap(x) == s == sc.fmt_hap_percent(x)


@pytest.mark.parametrize("name", list(FX["tables"]))
def test_every_printed_row_is_called_at_spec_defaults(oracle, name):
    """A matrix with exactly the printed (count, coverage) per row: the oracle at docs/SPEC.md defaults calls every
    row and nothing else, with the printed reference codon, at both ends of the SPEC's Bonferroni range."""
    t = FX["tables"][name]
    rows, ref, pos = sc.table_msa(t)
    genes = np.array([(1, 3 * len(pos) + 1)], dtype=oracle_lib.GENE)
    exp = sorted((i + 1, msa.codon_index(vc), c, p[3], msa.codon_index(p[2])) for i, p in enumerate(pos) for vc, c, _ in p[4])
    for nt in N_TESTS:
        v = oracle.call(rows, genes, refseq=ref, params=oracle_lib.default_params(n_tests=nt))
        got = [(int(r["codon_pos"]), int(r["codon"]), int(r["count"]), int(r["coverage"]), int(r["ref_codon"])) for r in v]
        assert got == exp, (name, nt)
        assert (v["p_value"] < 0.01).all() and (v["count"] > v["expected"]).all()
        for r in v:   # frequency = count / coverage displays as printed
            want = [x[2] for x in pos[r["codon_pos"] - 1][4] if msa.codon_index(x[0]) == r["codon"]][0]
            assert sc.fmt_percent(100.0 * r["count"] / r["coverage"]) == want
    if t.get("mode") == "majority" or all(float(r[5]) < 50 for g in t["genes"] for r in g["rows"]):
        # minors only: "tested against the major codon" gives the same table (doc/JULIET.md:133-134, hiv-unknown)
        v2 = oracle.call(rows, genes, params=oracle_lib.default_params(n_tests=N_TESTS[0]))
        v1 = oracle.call(rows, genes, refseq=ref, params=oracle_lib.default_params(n_tests=N_TESTS[0]))
        assert (v1 == v2).all()


def test_weakest_printed_call_sits_at_the_threshold(oracle):
    """G99G, 0.72 % of 2907 = 21 reads, is the weakest printed call: 20 reads are NOT called at the SPEC defaults."""
    prm = oracle_lib.default_params()
    e = oracle.expected(prm, 2907, msa.codon_index("GGG"), msa.codon_index("GGT"))
    assert e == 1
    for nt in N_TESTS:
        p21, _ = oracle.fisher(21, 2907 - 21, e, 2907 - e)
        p20, _ = oracle.fisher(20, 2907 - 20, e, 2907 - e)
        assert p21 * nt < 0.01 <= p20 * N_TESTS[1]


def test_hiv_phasing_table(oracle):
    """juliet_hiv-phasing.png in full: nine calls over three genes, haplotypes A..I global across the genes, the printed
    percentages, which haplotype carries which variant, Y181C + G190A together in C, wild type = A."""
    rows, ref, genes, pos, haps = sc.hiv_phasing()
    t = FX["tables"]["hiv_phasing"]
    g = np.array(genes, dtype=oracle_lib.GENE)
    v = oracle.call(rows, g, refseq=ref, params=oracle_lib.default_params(n_tests=1500))
    assert [(int(r["count"]), int(r["coverage"])) for r in v] == [(p[4][0][1], p[3]) for p in pos]
    assert [int(r["gene"]) for r in v] == [0, 1, 1, 1, 1, 1, 1, 1, 2]
    ph = oracle.phase(rows, v)
    s = ph["summary"]
    assert s["n_haplotypes"] == 9 and s["reported_reads"] + s["insufficient_reads"] + s["damaged_reads"] == len(rows)
    assert [sc.fmt_hap_percent(100.0 * c / s["reported_reads"]) for c in ph["hap_count"]] == t["haplotype_percent"]
    printed = [r[8] for gg in t["genes"] for r in gg["rows"]]
    for vi, letters in enumerate(printed):
        assert [t["haplotype_names"][h] for h in np.nonzero(ph["hit"][vi])[0]] == letters
    assert ph["hit"][:, 0].sum() == 0                                   # A: "Wild type, no variant" (doc/JULIET.md:201)
    assert ph["cooc"][5, 6] == ph["hap_count"][2] == 27                  # Y181C x G190A co-occur in C
    assert [int(c) for c in ph["hap_count"]] == [h[0] for h in haps]


def test_faq_variant_without_haplotype(oracle):
    """doc/JULIET.md:278-283, juliet_abl-nohaplotype.png: the reads of the three minor codons all carry a frame-shift
    deletion in another variant codon -> four calls (two codons at position 223 sharing one coverage), ONE haplotype
    (100 %) that carries only A217A."""
    rows, ref, e = sc.abl_nohaplotype()
    genes = np.array([(1, 10)], dtype=oracle_lib.GENE)
    v = oracle.call(rows, genes, refseq=ref, params=oracle_lib.default_params(n_tests=1130))   # (3585 - 193) / 3 codons
    got = sorted((msa.codon_string(r["codon"]), int(r["count"]), int(r["coverage"])) for r in v)
    assert got == sorted(e["calls"])
    assert v[1]["coverage"] == v[2]["coverage"] == 2401 and v[1]["codon_pos"] == v[2]["codon_pos"] == 2
    ph = oracle.phase(rows, v)
    assert ph["summary"]["n_haplotypes"] == 1 and ph["summary"]["reported_reads"] == e["reported"]
    assert sc.fmt_hap_percent(100.0 * ph["hap_count"][0] / ph["summary"]["reported_reads"]) == "100"
    hit_by_codon = {msa.codon_string(r["codon"]): int(ph["hit"][i, 0]) for i, r in enumerate(v)}
    assert hit_by_codon == {"GCG": 1, "GCC": 0, "CCA": 0, "TTC": 0}
    assert ph["summary"]["marginal_gap"] >= 22 + 23 + 34                 # the minor carriers are damaged by deletions


def test_faq_no_haplotype_columns(oracle):
    """doc/JULIET.md:285-288: every read has a deletion in one of the variant codons -> calls, but no haplotype."""
    rows, ref, e = sc.no_haplotype_columns()
    genes = np.array([(1, 7)], dtype=oracle_lib.GENE)
    v = oracle.call(rows, genes, refseq=ref, params=oracle_lib.default_params(n_tests=1000))
    assert [(msa.codon_string(r["codon"]), int(r["count"]), int(r["coverage"])) for r in v] == e["calls"]
    ph = oracle.phase(rows, v)
    s = ph["summary"]
    assert s["n_haplotypes"] == 0 and s["reported_reads"] == 0 and s["insufficient_reads"] == 0
    assert s["damaged_reads"] == s["marginal_gap"] == len(rows)
    assert ph["hit"].shape == (2, 0) and (ph["read_hap"] == oracle_lib.HAP_DAMAGED).all()


def test_faq_major_calls_dilute_minor_haplotypes(oracle):
    """doc/JULIET.md:356-366, juliet_major-before.png / -after.png.  Before: one haplotype (100 %) carrying every
    >= 99 % call, M41L and K65R unassigned.  With --max-perc 90: A 95.8 (wild type), B 1.1 {Y181C, G190A},
    C 1.1 {K65R}, D 1 {T215Y}, E 1 {M41L}."""
    rows, ref, pos, minor, printed_after = sc.major_dilution()
    genes = np.array([(1, 3 * len(pos) + 1)], dtype=oracle_lib.GENE)
    v = oracle.call(rows, genes, refseq=ref, params=oracle_lib.default_params(n_tests=1500))
    assert [(msa.codon_string(r["codon"]), int(r["count"]), int(r["coverage"])) for r in v] == [(p[3], p[4], p[5]) for p in pos]
    before = oracle.phase(rows, v)
    assert before["summary"]["n_haplotypes"] == 1
    is_minor = np.isin(np.arange(len(pos)), list(minor.values()))
    assert (before["hit"][~is_minor, 0] == 1).all() and (before["hit"][is_minor, 0] == 0).all()
    assert before["summary"]["insufficient_reads"] == 32                 # 4 minor haplotypes x 8 clean reads: below 10
    keep = v[100.0 * v["count"] / v["coverage"] < 90.0]                  # --max-perc 90 (doc/JULIET.md:352-354)
    assert len(keep) == 5
    after = oracle.phase(rows, keep)
    s = after["summary"]
    assert [sc.fmt_hap_percent(100.0 * c / s["reported_reads"]) for c in after["hap_count"]] == printed_after
    aa_of = [pos[i][1] for i in sorted(minor.values())]
    members = {h: tuple(aa_of[k] for k in np.nonzero(after["hit"][:, h])[0]) for h in range(5)}
    assert members == {0: (), 1: (181, 190), 2: (65,), 3: (215,), 4: (41,)}
