"""The reference's printed known answers through the HIP path (C ABI and the `juliet` CLI): the same matrices as
tests/test_oracle_golden_rows.py — every variant row printed on the screenshots under /root/reference/doc/img and the
three phasing scenarios of the FAQ (doc/JULIET.md:278-288, 356-366).  Each result is compared with the oracle
(bit-exact integers, p within 1e-10) AND with what the screenshot prints."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

import oracle_lib
import scenarios as sc
from minorseq_amd import capi, msa
from test_gpu_parity import assert_phase_equal, assert_variants_equal, oracle_params

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JULIET = os.path.join(ROOT, "minorseq_amd", "bin", "juliet")
SYNTH = os.path.join(ROOT, "minorseq_amd", "bin", "juliet-synth")
FX = sc.load_fixture()


@pytest.fixture(scope="module")
def jl():
    j = capi.Juliet(0)
    yield j
    j.close()


def run_both(jl, oracle, rows, genes, ref, prm, min_reads=10):
    """The whole path on the device (one graph) and on the oracle; asserts parity and returns the device result."""
    genes = np.array(genes, dtype=capi.GENE)
    jl.upload_rows(rows)
    out = jl.run(genes, ref, prm, phasing=True, min_reads=min_reads)
    exp_v = oracle.call(rows, genes, refseq=ref, params=oracle_params(prm))
    perc = 100.0 * exp_v["count"] / np.maximum(exp_v["coverage"], 1)
    if prm.min_perc >= 0:
        exp_v = exp_v[perc > prm.min_perc]
        perc = 100.0 * exp_v["count"] / np.maximum(exp_v["coverage"], 1)
    if prm.max_perc >= 0:
        exp_v = exp_v[perc < prm.max_perc]
    assert_variants_equal(out["variants"], exp_v)
    assert_phase_equal(out["phase"], oracle.phase(rows, exp_v, min_reads), len(exp_v))
    return out


@pytest.mark.parametrize("name", list(FX["tables"]))
def test_printed_rows_are_called_on_the_device(jl, oracle, name):
    t = FX["tables"][name]
    rows, ref, pos = sc.table_msa(t)
    exp = sorted((i + 1, msa.codon_index(vc), c, p[3]) for i, p in enumerate(pos) for vc, c, _ in p[4])
    for nt in (982.0, 1884.0):
        out = run_both(jl, oracle, rows, [(1, 3 * len(pos) + 1)], ref, capi.default_params(n_tests=nt))
        v = out["variants"]
        assert [(int(r["codon_pos"]), int(r["codon"]), int(r["count"]), int(r["coverage"])) for r in v] == exp
        for r in v:
            want = [x[2] for x in pos[r["codon_pos"] - 1][4] if msa.codon_index(x[0]) == r["codon"]][0]
            assert sc.fmt_percent(100.0 * r["count"] / r["coverage"]) == want


def test_hiv_phasing_table_on_the_device(jl, oracle):
    rows, ref, genes, pos, haps = sc.hiv_phasing()
    t = FX["tables"]["hiv_phasing"]
    out = run_both(jl, oracle, rows, genes, ref, capi.default_params(n_tests=1500))
    ph = out["phase"]
    s = ph["summary"]
    assert [sc.fmt_hap_percent(100.0 * c / s["reported_reads"]) for c in ph["hap_count"]] == t["haplotype_percent"]
    printed = [r[8] for g in t["genes"] for r in g["rows"]]
    for vi, letters in enumerate(printed):
        assert [capi.haplotype_name(h) for h in np.nonzero(ph["hit"][vi, :9])[0]] == letters
    assert [int(r["gene"]) for r in out["variants"]] == [0, 1, 1, 1, 1, 1, 1, 1, 2]     # columns are global across genes


def test_faq_scenarios_on_the_device(jl, oracle):
    # a variant without a haplotype (doc/JULIET.md:278-283)
    rows, ref, e = sc.abl_nohaplotype()
    out = run_both(jl, oracle, rows, [(1, 10)], ref, capi.default_params(n_tests=1130))
    assert out["phase"]["summary"]["n_haplotypes"] == 1
    assert {msa.codon_string(r["codon"]): int(out["phase"]["hit"][i, 0]) for i, r in enumerate(out["variants"])} == \
        {"GCG": 1, "GCC": 0, "CCA": 0, "TTC": 0}
    # no haplotype columns at all (doc/JULIET.md:285-288)
    rows, ref, e = sc.no_haplotype_columns()
    out = run_both(jl, oracle, rows, [(1, 7)], ref, capi.default_params(n_tests=1000))
    s = out["phase"]["summary"]
    assert len(out["variants"]) == 2 and s["n_haplotypes"] == 0 and s["reported_reads"] == 0 and s["damaged_reads"] == len(rows)
    assert (out["phase"]["read_hap"] == capi.HAP_DAMAGED).all()
    # major calls dilute the minor haplotypes; --max-perc 90 brings them back (doc/JULIET.md:356-366)
    rows, ref, pos, minor, printed_after = sc.major_dilution()
    genes = [(1, 3 * len(pos) + 1)]
    before = run_both(jl, oracle, rows, genes, ref, capi.default_params(n_tests=1500))       # 16 positions: multi-word keys
    assert before["phase"]["summary"]["n_haplotypes"] == 1 and len(before["variants"]) == 16
    is_minor = np.isin(np.arange(len(pos)), list(minor.values()))
    assert (before["phase"]["hit"][:16, 0][~is_minor] == 1).all() and (before["phase"]["hit"][:16, 0][is_minor] == 0).all()
    after = run_both(jl, oracle, rows, genes, ref, capi.default_params(n_tests=1500, max_perc=90.0))
    s = after["phase"]["summary"]
    assert len(after["variants"]) == 5
    assert [sc.fmt_hap_percent(100.0 * c / s["reported_reads"]) for c in after["phase"]["hap_count"]] == printed_after
    aa_of = [pos[i][1] for i in sorted(minor.values())]
    members = {h: tuple(aa_of[k] for k in np.nonzero(after["phase"]["hit"][:5, h])[0]) for h in range(5)}
    assert members == {0: (), 1: (181, 190), 2: (65,), 3: (215,), 4: (41,)}


def test_faq_scenarios_through_a_group_run(oracle):
    """The same three matrices as windows of ONE group launch (different shapes, one of them needing the multi-word
    pipeline): results per window equal the oracle's."""
    mats = [sc.abl_nohaplotype()[:2], sc.no_haplotype_columns()[:2]]
    rows3, ref3, pos3, _, _ = sc.major_dilution()
    # one gene / reference for all windows of a group: give every window the widest layout, padded with uncovered columns
    l = 3 * len(pos3)
    ctxs, exp = [], []
    genes = np.array([(1, l + 1)], dtype=capi.GENE)
    prm = capi.default_params(n_tests=1500)
    for rows, ref in mats + [(rows3, ref3)]:
        full = np.full((rows.shape[0], l), msa.SYM_NONE, dtype=np.uint8)
        full[:, : rows.shape[1]] = rows
        c = capi.Juliet(0)
        c.upload_rows(full)
        ctxs.append(c)
        exp.append(full)
    grp = capi.Group(ctxs)
    grp.run_async(genes, None, prm, True, 10, True)          # majority-codon mode: no common reference
    for c, full in zip(ctxs, exp):
        got = c.run_view() or c.run_fetch(True, True, cap_var=64)
        ev = oracle.call(full, genes, params=oracle_params(prm))
        assert_variants_equal(got["variants"], ev)
        ep = oracle.phase(full, ev)
        if ep["summary"]["n_positions"] > 10:
            got = c.run_fetch(True, True, cap_var=64)        # flagged: the fetch re-runs the multi-word pipeline
        assert_phase_equal(got["phase"], ep, len(ev))
    grp.close()
    for c in ctxs:
        c.close()


# ---------------------------------------------------------------------------------------------- through the CLI
def to_bam(tmp, rows, ref, genes, name, drms=None):
    mpath, bam, cfg = (str(tmp / f"{name}.{x}") for x in ("msa", "bam", "json"))
    with open(mpath, "wb") as f:
        f.write(np.array([rows.shape[0], rows.shape[1], 0], dtype=np.uint64).tobytes())
        f.write(np.ascontiguousarray(rows, dtype=np.uint8).tobytes())
    refs = "".join("ACGT"[b] for b in ref)
    subprocess.check_call([SYNTH, "--from-rows", mpath, "--ref", refs, "-o", bam])
    json.dump({"genes": [dict(name=n, begin=b, end=e, drms=drms or []) for n, b, e in genes], "referenceName": "printed",
               "referenceSequence": refs, "version": "tests/scenarios.py", "databaseVersion": "none"}, open(cfg, "w"))
    return bam, cfg


def run_cli(tmp, bam, cfg, *args):
    out, html = str(tmp / "o.json"), str(tmp / "o.html")
    subprocess.check_call([JULIET, "-c", cfg, "--mode-phasing", *args, bam, out, html])
    return json.load(open(out)), open(html).read()


def test_cli_reproduces_the_phasing_screenshot(tmp_path):
    """juliet_hiv-phasing.png as `juliet -c cfg --mode-phasing in.bam out.json out.html` prints it: gene tables, percentages
    as displayed, haplotype names A..I with their percentages, haplotype_hit per variant."""
    rows, ref, genes, pos, haps = sc.hiv_phasing()
    t = FX["tables"]["hiv_phasing"]
    named = [(g["name"], b, e) for g, (b, e) in zip(t["genes"], genes)]
    bam, cfg = to_bam(tmp_path, rows, ref, named, "phasing")
    j, html = run_cli(tmp_path, bam, cfg, "--n-tests", "1500")
    assert [g["name"] for g in j["genes"]] == [g["name"] for g in t["genes"]]
    hb = j["haplotype"]
    assert [h["name"] for h in hb["haplotypes"]] == t["haplotype_names"]
    assert [sc.fmt_hap_percent(100.0 * h["frequency"]) for h in hb["haplotypes"]] == t["haplotype_percent"]
    k = 0
    for g, tg in zip(j["genes"], t["genes"]):
        assert len(g["variant_positions"]) == len(tg["rows"])
        for vp, r in zip(g["variant_positions"], tg["rows"]):
            vc = vp["variant_amino_acids"][0]["variant_codons"][0]
            assert (vp["ref_codon"], vp["ref_amino_acid"], vp["variant_amino_acids"][0]["amino_acid"], vc["codon"],
                    vp["coverage"]) == (r[0], r[1], r[3], r[4], r[6])
            assert sc.fmt_percent(100.0 * vc["frequency"]) == r[5]
            assert [t["haplotype_names"][h] for h, x in enumerate(vc["haplotype_hit"]) if x] == r[8]
            # the HTML row prints the same strings as the screenshot (codon, AA, AA, codon, %, coverage)
            assert re.search(rf"<td>{r[0]}</td><td>{r[1]}</td><td>\d+</td><td>{r[3]}</td><td>{r[4]}</td><td>{re.escape(r[5])}</td><td>{r[6]}</td>", html)
            k += 1
    assert k == 9 and hb["reported_reads"] + hb["insufficient_coverage_reads"] + hb["damaged_reads"] == len(rows)


def test_cli_faq_scenarios(tmp_path):
    # ABL1 223: two codons at one position, amino acids in printed order (A above P), one haplotype without the minors
    rows, ref, e = sc.abl_nohaplotype()
    bam, cfg = to_bam(tmp_path, rows, ref, [("ABL1", 1, 10)], "abl")
    j, html = run_cli(tmp_path, bam, cfg, "--n-tests", "1130")
    vps = j["genes"][0]["variant_positions"]
    assert [vp["coverage"] for vp in vps] == [2289, 2401, 2077]
    assert [(a["amino_acid"], a["variant_codons"][0]["codon"]) for a in vps[1]["variant_amino_acids"]] == [("A", "GCC"), ("P", "CCA")]
    hits = [vc["haplotype_hit"] for vp in vps for a in vp["variant_amino_acids"] for vc in a["variant_codons"]]
    assert hits == [[True], [False], [False], [False]]
    assert [sc.fmt_hap_percent(100.0 * h["frequency"]) for h in j["haplotype"]["haplotypes"]] == ["100"]
    # no haplotype columns
    rows, ref, e = sc.no_haplotype_columns()
    bam, cfg = to_bam(tmp_path, rows, ref, [("g", 1, 7)], "nohap")
    j, _ = run_cli(tmp_path, bam, cfg, "--n-tests", "1000")
    assert j["haplotype"]["haplotypes"] == [] and j["haplotype"]["damaged_reads"] == len(rows)
    assert all(vc["haplotype_hit"] == [] for g in j["genes"] for vp in g["variant_positions"]
               for a in vp["variant_amino_acids"] for vc in a["variant_codons"])
    # --max-perc 90 (doc/JULIET.md:352-366)
    rows, ref, pos, minor, printed_after = sc.major_dilution()
    bam, cfg = to_bam(tmp_path, rows, ref, [("Reverse Transcriptase", 1, 3 * len(pos) + 1)], "major")
    j, _ = run_cli(tmp_path, bam, cfg, "--n-tests", "1500")
    assert len(j["haplotype"]["haplotypes"]) == 1
    j, _ = run_cli(tmp_path, bam, cfg, "--n-tests", "1500", "--max-perc", "90")
    assert [sc.fmt_hap_percent(100.0 * h["frequency"]) for h in j["haplotype"]["haplotypes"]] == printed_after
    assert sum(len(g["variant_positions"]) for g in j["genes"]) == 5
