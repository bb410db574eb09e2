"""End to end on the GPU: synthetic PacBio-style BAM -> `juliet` (C++ front end over the C ABI) -> JSON,
checked against the oracle run on the same reads.  Exercises the documented command-line surface
(doc/JULIET.md:62-66, 121, 195, 270-271, 342-370)."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

import oracle_lib
from minorseq_amd import capi, msa, synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JULIET = os.path.join(ROOT, "minorseq_amd", "bin", "juliet")
SYNTH = os.path.join(ROOT, "minorseq_amd", "bin", "juliet-synth")
N, L, SEED = 6000, 900, 4
MINOR = (60, 50, 40, 30)


@pytest.fixture(scope="module", autouse=True)
def built():
    """The binaries normally travel with the tree; build them only if they are missing (never under a loaded .so)."""
    if not os.path.exists(os.path.join(ROOT, "minorseq_amd", "libjuliet_hip.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "minorseq_amd", "csrc")])
    if not (os.path.exists(JULIET) and os.path.exists(SYNTH) and os.path.exists(os.path.join(os.path.dirname(JULIET), "fuse"))):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "minorseq_amd", "host")])


@pytest.fixture(scope="module")
def sample(tmp_path_factory):
    d = tmp_path_factory.mktemp("cli")
    bam, cfg = str(d / "s.align.bam"), str(d / "cfg.json")
    subprocess.check_call([SYNTH, "--reads", str(N), "--cols", str(L), "--seed", str(SEED), "--partial", "0.1",
                           "--minor-permille", *map(str, MINOR), "-o", bam, "--config-out", cfg])
    sp = synth.SynthParams(seed=SEED, partial_rate=0.1, minor_permille=MINOR)
    ref = synth.reference(SEED, L)
    return d, bam, cfg, synth.rows(sp, L, 0, N, ref), ref


def run_juliet(d, bam, *args, out="o.json"):
    o = str(d / out)
    subprocess.check_call([JULIET, *args, bam, o])
    return json.load(open(o)) if o.endswith(".json") else open(o).read()


def flat_variants(j):
    rows = []
    for gi, g in enumerate(j["genes"]):
        for vp in g["variant_positions"]:
            for aa in vp["variant_amino_acids"]:
                for vc in aa["variant_codons"]:
                    rows.append((gi, vp["ref_position"], msa.codon_index(vc["codon"]), vc, vp, aa))
    return sorted(rows, key=lambda r: r[:3])


def test_json_matches_oracle_with_phasing(sample, oracle):
    d, bam, cfg, rows, ref = sample
    j = run_juliet(d, bam, "-c", cfg, "--mode-phasing")
    genes = np.array([(1, L + 1)], dtype=capi.GENE)
    exp = oracle.call(rows, genes, refseq=ref)
    got = flat_variants(j)
    assert len(got) == len(exp) >= 4
    for (gi, pos, cod, vc, vp, aa), e in zip(got, exp):
        assert (gi, pos, cod) == (e["gene"], e["codon_pos"], e["codon"])
        assert vc["count"] == e["count"] and vp["coverage"] == e["coverage"] and vc["expected"] == e["expected"]
        assert vc["frequency"] == e["count"] / e["coverage"]
        assert abs(vc["pValue"] - e["p_value"]) <= 1e-10
        assert vp["ref_codon"] == msa.codon_string(e["ref_codon"])
        assert vc["known_drm"] == "synthetic drug"        # the planted edits are the config's DRMs
        # MSA context rows are the column pileup (doc/JULIET.md:99-100)
        col = oracle.pileup(rows)
        for m in vp["msa"]:
            c = m["abs_pos"] - 1
            assert [m[s] for s in "ACGT-N"] == col[c].tolist() and m["wt"] == "ACGT"[ref[c]]
        assert [m["rel_pos"] for m in vp["msa"]] == list(range(max(-3, -int(e["col"])), 6))
    ph = oracle.phase(rows, exp)
    hb = j["haplotype"]
    s = ph["summary"]
    assert (hb["reported_reads"], hb["insufficient_coverage_reads"], hb["damaged_reads"]) == \
        (s["reported_reads"], s["insufficient_reads"], s["damaged_reads"])
    assert hb["reported_reads"] + hb["insufficient_coverage_reads"] + hb["damaged_reads"] == N   # doc/JULIET.md:378-379
    assert (hb["marginal_gaps"], hb["marginal_heteroduplexes"], hb["marginal_partial"]) == \
        (s["marginal_gap"], s["marginal_heteroduplex"], s["marginal_partial"])
    assert [h["reads"] for h in hb["haplotypes"]] == ph["hap_count"].tolist()
    assert all(re.fullmatch(r"[A-Z][a-z]?", h["name"]) for h in hb["haplotypes"])             # doc/JULIET.md:198
    assert [h["name"] for h in hb["haplotypes"]] == [capi.haplotype_name(i) for i in range(len(hb["haplotypes"]))]
    assert abs(sum(h["frequency"] for h in hb["haplotypes"]) - 1.0) < 1e-12
    for hi, h in enumerate(hb["haplotypes"]):
        assert [msa.codon_index(c) for c in h["codons"]] == ph["hap_pattern"][hi].tolist()
        idx = sorted(int(n.split("/")[1]) for n in h["read_names"])
        assert idx == np.nonzero(ph["read_hap"] == hi)[0].tolist()
    for k, (gi, pos, cod, vc, vp, aa) in enumerate(got):
        assert vc["haplotype_hit"] == [bool(x) for x in ph["hit"][k]]                           # doc/JULIET.md:207-211
    # drug summaries (doc/JULIET.md:104-107): the made-up drug lists all five planted mutations
    assert [d["drug"] for d in j["drug_summaries"]] == ["synthetic drug"]
    muts = j["drug_summaries"][0]["variants"]
    assert len(muts) == len(exp) and all(re.fullmatch(r"[A-Z]\d+[A-Z]", m["mutation"]) for m in muts)
    assert [m["frequency"] for m in muts] == [e["count"] / e["coverage"] for e in exp]
    # traceability block (doc/JULIET.md:75-79)
    assert j["input"]["input_file"] == bam and "--mode-phasing" in j["input"]["command_line"]
    assert re.fullmatch(r"\d{4}-\d\d-\d\dT\d\d:\d\d:\d\d\.\d{3}Z", j["input"]["timestamp"])
    assert j["target_config"]["referenceName"] == "synthetic_ref" and j["target_config"]["chemistry_model"] == "sequel"


def test_filters_and_region(sample, oracle):
    d, bam, cfg, rows, ref = sample
    genes = np.array([(1, L + 1)], dtype=capi.GENE)
    exp = oracle.call(rows, genes, refseq=ref)
    perc = 100.0 * exp["count"] / exp["coverage"]
    j = run_juliet(d, bam, "-c", cfg, "--min-perc", "4.5")
    assert [(r[1], r[2]) for r in flat_variants(j)] == [(e["codon_pos"], e["codon"]) for e in exp[perc > 4.5]]
    j = run_juliet(d, bam, "-c", cfg, "--max-perc", "4.5")
    assert [(r[1], r[2]) for r in flat_variants(j)] == [(e["codon_pos"], e["codon"]) for e in exp[perc < 4.5]]
    assert "haplotype" not in j                                   # phasing is opt-in (doc/JULIET.md:194-195)
    # --region: AA numbering stays relative to the gene (first_codon offset)
    b, e_ = 31, 400
    j = run_juliet(d, bam, "-c", cfg, "--region", f"{b}-{e_}", "--n-tests", "300")
    sub = oracle.call(rows, np.array([(b, e_)], dtype=capi.GENE), refseq=ref, params=oracle_lib.default_params(n_tests=300))
    assert [(r[1], r[2]) for r in flat_variants(j)] == [(x["codon_pos"] + (b - 1) // 3, x["codon"]) for x in sub]
    assert len(sub) >= 1


def test_fisher_tail_flag(sample, oracle):
    """--fisher-tail greater|two-sided (doc/JULIET.md:38-42 leaves the sidedness open): the CLI hands
    jl_params.tail to the device; the JSON matches the oracle's two-sided table, p-values within 1e-10, and the two-sided
    calls are a subset of the one-sided ones (p doubles)."""
    d, bam, cfg, rows, ref = sample
    genes = np.array([(1, L + 1)], dtype=capi.GENE)
    one = oracle.call(rows, genes, refseq=ref)
    two = oracle.call(rows, genes, refseq=ref, params=oracle_lib.default_params(tail=1))
    j2 = flat_variants(run_juliet(d, bam, "-c", cfg, "--fisher-tail", "two-sided", out="two.json"))
    j1 = flat_variants(run_juliet(d, bam, "-c", cfg, "--fisher-tail", "greater", out="one.json"))
    for got, exp in ((j1, one), (j2, two)):
        assert len(got) == len(exp) >= 4
        for (gi, pos, cod, vc, vp, aa), e in zip(got, exp):
            assert (gi, pos, cod, vc["count"], vp["coverage"]) == (e["gene"], e["codon_pos"], e["codon"], e["count"], e["coverage"])
            assert abs(vc["pValue"] - e["p_value"]) <= 1e-10
    k1 = {(r[0], r[1], r[2]): r[3]["pValue"] for r in j1}
    for r in j2:
        assert (r[0], r[1], r[2]) in k1
        assert abs(r[3]["pValue"] - min(1.0, 2.0 * k1[(r[0], r[1], r[2])])) <= 1e-10 or r[3]["pValue"] >= k1[(r[0], r[1], r[2])]
    bad = subprocess.run([JULIET, "-c", cfg, "--fisher-tail", "less", bam, str(d / "bad.json")], capture_output=True, text=True)
    assert bad.returncode == 1 and "greater or two-sided" in bad.stderr


def test_drm_only_keeps_only_config_mutations(tmp_path, oracle):
    """--drm-only (doc/JULIET.md:370) on a noisy sample where plenty of non-DRM codons are significant too."""
    n, l, seed = 3000, 300, 8
    bam, cfg = str(tmp_path / "noisy.bam"), str(tmp_path / "noisy.json")
    subprocess.check_call([SYNTH, "--reads", str(n), "--cols", str(l), "--seed", str(seed), "--sub", "0.012",
                           "--minor-permille", "60", "50", "40", "30", "-o", bam, "--config-out", cfg])
    sp = synth.SynthParams(seed=seed, sub_rate=0.012, minor_permille=(60, 50, 40, 30))
    ref = synth.reference(seed, l)
    rows = synth.rows(sp, l, 0, n, ref)
    genes = np.array([(1, l + 1)], dtype=capi.GENE)
    every = flat_variants(run_juliet(tmp_path, bam, "-c", cfg))
    only = flat_variants(run_juliet(tmp_path, bam, "-c", cfg, "--drm-only"))
    assert len(every) > len(only) >= 4
    assert all(r[3]["known_drm"] == "synthetic drug" for r in only)
    assert [(r[1], r[2]) for r in only] == [(r[1], r[2]) for r in every if r[3]["known_drm"]]
    exp = oracle.call(rows, genes, refseq=ref)
    assert [(r[1], r[2]) for r in every] == [(e["codon_pos"], e["codon"]) for e in exp]


def test_no_config_majority_mode_and_html(sample, oracle):
    d, bam, cfg, rows, ref = sample
    j = run_juliet(d, bam)                                        # doc/JULIET.md:182-188
    assert [g["name"] for g in j["genes"]] == ["unknown"]
    lo = int((rows != 6).any(axis=0).argmax())
    hi = L - int((rows[:, ::-1] != 6).any(axis=0).argmax())
    exp = oracle.call(rows[:, lo:hi], np.array([(lo + 1, hi + 1)], dtype=capi.GENE), win_begin=lo)
    assert [(r[1], r[2]) for r in flat_variants(j)] == [(e["codon_pos"], e["codon"]) for e in exp]
    html = run_juliet(d, bam, "-c", cfg, out="o.html")
    assert html.startswith("<!DOCTYPE html>") and "Variant Discovery" in html and "synthetic drug" in html
    # two outputs at once (doc/JULIET.md:65)
    subprocess.check_call([JULIET, "-c", cfg, bam, str(d / "both.html"), str(d / "both.json")])
    assert os.path.getsize(d / "both.html") > 0 and json.load(open(d / "both.json"))["genes"]


def test_the_long_way_out_writes_the_same_files(sample):
    """The front end ends its process as soon as its output is written and closed (`_exit`); with JL_SLOW_EXIT=1 it returns from main
    — destructors, the runtime's own teardown: what a profiler or a leak checker needs.  Same JSON, same HTML, exit code 0 both ways."""
    d, bam, cfg, rows, ref = sample
    outs = {}
    for tag, env in (("fast", {}), ("slow", {"JL_SLOW_EXIT": "1"})):
        j, h = str(d / f"exit_{tag}.json"), str(d / f"exit_{tag}.html")
        r = subprocess.run([JULIET, "-c", cfg, "--mode-phasing", bam, j, h], env=dict(os.environ, **env), capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        outs[tag] = (open(j).read(), open(h).read())
    fj, sj = (json.loads(outs[t][0]) for t in ("fast", "slow"))
    for doc in (fj, sj):      # (the run's own stamp and command line)
        doc["input"].pop("timestamp")
        doc["input"].pop("command_line")
    assert fj == sj and abs(len(outs["fast"][1]) - len(outs["slow"][1])) < 64 and fj["genes"]


def test_consensus_by_product(sample, oracle):
    """--consensus: majority base per column, majority-deletion columns dropped (doc/FUSE.md:17-24, without insertions)."""
    d, bam, cfg, rows, ref = sample
    fa = str(d / "cons.fasta")
    subprocess.check_call([JULIET, "-c", cfg, "--consensus", fa, bam, str(d / "c.json")])
    lines = open(fa).read().splitlines()
    assert lines[0].startswith(">consensus window=1-")
    seq = "".join(lines[1:])
    col = oracle.pileup(rows)
    exp = ""
    for c in range(L):
        k = col[c, :5]
        if k.max() == 0:
            exp += "N"
        elif int(k.argmax()) < 4:
            exp += "ACGT"[int(k.argmax())]
    assert seq[: len(exp)] == exp[: len(seq)] and abs(len(seq) - len(exp)) <= 5   # window may carry context columns
    assert seq.startswith("".join("ACGT"[b] for b in ref[:50]))                 # the 96 % major clone is the reference


def test_rich_qv_filter_end_to_end(tmp_path, oracle):
    """Filtered bases arrive as letters + a poor sq track (ccs --richQVs); `--min-qv` masks them on the device
    and the calls equal those on the N-encoded reads."""
    n, l, seed = 4000, 300, 12
    bam, cfg = str(tmp_path / "rich.bam"), str(tmp_path / "rich.json")
    subprocess.check_call([SYNTH, "--reads", str(n), "--cols", str(l), "--seed", str(seed), "--rich-qv",
                           "--minor-permille", "60", "50", "40", "30", "-o", bam, "--config-out", cfg])
    sp = synth.SynthParams(seed=seed, minor_permille=(60, 50, 40, 30))
    ref = synth.reference(seed, l)
    rows = synth.rows(sp, l, 0, n, ref)
    j = run_juliet(tmp_path, bam, "-c", cfg, "--min-qv", "10")
    exp = oracle.call(rows, np.array([(1, l + 1)], dtype=capi.GENE), refseq=ref)
    got = flat_variants(j)
    assert [(r[1], r[2], r[3]["count"], r[4]["coverage"]) for r in got] == \
        [(e["codon_pos"], e["codon"], e["count"], e["coverage"]) for e in exp]
    # unfiltered: coverage is higher because nothing is masked
    j0 = run_juliet(tmp_path, bam, "-c", cfg)
    assert all(a[4]["coverage"] > b[4]["coverage"] for a, b in zip(flat_variants(j0), got))


@pytest.mark.parametrize("rich", [False, True])
def test_records_in_several_chunks_keep_their_order(tmp_path, oracle, rich):
    """The front end hands the records to the device 8192 at a time while it parses the next chunk (jl_records_append):
    20 000 reads = three chunks.  Calls, read categories and — the part that depends on chunk ORDER — the read names of
    every haplotype must be those of the reads in file order; with --min-qv the qualities travel in the same chunks."""
    n, l, seed = 20_000, 300, 31
    bam, cfg = str(tmp_path / "c.bam"), str(tmp_path / "c.json")
    subprocess.check_call([SYNTH, "--reads", str(n), "--cols", str(l), "--seed", str(seed), "--partial", "0.15",
                           *( ["--rich-qv"] if rich else []), "--minor-permille", "60", "50", "40", "30", "-o", bam, "--config-out", cfg])
    sp = synth.SynthParams(seed=seed, partial_rate=0.15, minor_permille=(60, 50, 40, 30))
    ref = synth.reference(seed, l)
    rows = synth.rows(sp, l, 0, n, ref)
    j = run_juliet(tmp_path, bam, "-c", cfg, "--mode-phasing", *(["--min-qv", "10"] if rich else []))
    exp = oracle.call(rows, np.array([(1, l + 1)], dtype=capi.GENE), refseq=ref)
    got = flat_variants(j)
    assert [(r[1], r[2], r[3]["count"], r[4]["coverage"]) for r in got] == \
        [(e["codon_pos"], e["codon"], e["count"], e["coverage"]) for e in exp] and len(exp) >= 4
    ph = oracle.phase(rows, exp)
    hb = j["haplotype"]
    assert [h["reads"] for h in hb["haplotypes"]] == ph["hap_count"].tolist()
    for hi, h in enumerate(hb["haplotypes"]):
        idx = [int(nm.split("/")[1]) for nm in h["read_names"]]
        assert idx == np.nonzero(ph["read_hap"] == hi)[0].tolist()      # file order, across the chunk boundaries at 8192, 16384
    assert hb["reported_reads"] + hb["insufficient_coverage_reads"] + hb["damaged_reads"] == n


class _Dom:
    """Minimal HTML tree (tag, attrs, children, text) — enough to read the page back cell by cell."""

    def __init__(self, html):
        from html.parser import HTMLParser

        root = dict(tag="root", attrs={}, children=[], text="")
        stack = [root]

        class P(HTMLParser):
            def handle_starttag(self, tag, attrs):
                node = dict(tag=tag, attrs=dict(attrs), children=[], text="")
                stack[-1]["children"].append(node)
                if tag not in ("meta", "br"):
                    stack.append(node)

            def handle_endtag(self, tag):
                while len(stack) > 1 and stack[-1]["tag"] != tag:
                    stack.pop()
                if len(stack) > 1:
                    stack.pop()

            def handle_data(self, data):
                stack[-1]["text"] += data

        P(convert_charrefs=True).feed(html)
        self.root = root

    def find(self, node=None, **want):
        node = node or self.root
        out = []
        for c in node["children"]:
            ok = all((c["tag"] == v) if k == "tag" else (c["attrs"].get(k.rstrip("_")) == v) for k, v in want.items())
            if ok:
                out.append(c)
            out += self.find(c, **want)
        return out

    @staticmethod
    def text(node):
        return node["text"] + "".join(_Dom.text(c) for c in node["children"])

    def rows(self, table):
        return [[self.text(c).strip() for c in tr["children"] if c["tag"] in ("td", "th")] for tr in table["children"] if tr["tag"] == "tr"]


def test_html_is_a_one_to_one_rendering_of_the_json(sample):
    """doc/JULIET.md:68-69: "The HTML page is a 1:1 conversion of the JSON file and contains the identical information".
    Both outputs of ONE run; the page is parsed back and every cell compared with the JSON value it renders (numbers in
    the display format the reference's screenshots print), every JSON leaf must be on the page, and no table has rows
    the JSON does not."""
    import scenarios as sc
    d, bam, cfg, rows, ref = sample
    subprocess.check_call([JULIET, "-c", cfg, "--mode-phasing", bam, str(d / "one.json"), str(d / "one.html")])
    j = json.load(open(d / "one.json"))
    dom = _Dom(open(d / "one.html").read())
    # section 1 and 2: key / value tables
    (t,) = dom.find(tag="table", id="input-table")
    assert dict(dom.rows(t)) == {k: str(v) for k, v in j["input"].items()}
    (t,) = dom.find(tag="table", id="target-table")
    want = {k: (str(v) if not isinstance(v, float) else repr(v)) for k, v in j["target_config"].items() if k != "genes"}
    assert dict(dom.rows(t)) == want
    (ul,) = dom.find(tag="ul", id="target-genes")
    lis = [c for c in ul["children"] if c["tag"] == "li"]
    assert len(lis) == len(j["target_config"]["genes"])
    for li, g in zip(lis, j["target_config"]["genes"]):
        assert (li["attrs"]["data-begin"], li["attrs"]["data-end"]) == (str(g["begin"]), str(g["end"]))
        assert dom.text(dom.find(li, tag="b")[0]) == g["name"]
        drms = dom.find(li, tag="li", class_="drm")
        assert [dom.text(dom.find(x, tag="span", class_="drm-name")[0]) for x in drms] == [q["name"] for q in g["drms"]]
        assert [[dom.text(s) for s in dom.find(x, tag="span", class_="drm-pos")] for x in drms] == [q["positions"] for q in g["drms"]]
    # section 3: one table per gene, one row per variant codon, haplotype columns, context tables
    haps = j["haplotype"]["haplotypes"]
    tables = dom.find(tag="table", class_="gene")
    assert [t["attrs"]["data-gene"] for t in tables] == [g["name"] for g in j["genes"]]
    n_rows = 0
    for t, g in zip(tables, j["genes"]):
        head = dom.rows(t)
        assert head[0][2:] == [h["name"] for h in haps]
        assert head[1][8:] == [sc.fmt_hap_percent(100.0 * h["frequency"]) for h in haps]
        vrows = [tr for tr in t["children"] if tr["attrs"].get("class") == "variant"]
        crows = [tr for tr in t["children"] if tr["attrs"].get("class") == "context"]
        flat = [(vp, aa, vc) for vp in g["variant_positions"] for aa in vp["variant_amino_acids"] for vc in aa["variant_codons"]]
        assert len(vrows) == len(flat) and len(crows) == len(g["variant_positions"])
        for tr, (vp, aa, vc) in zip(vrows, flat):
            cells = [dom.text(c).strip() for c in tr["children"]]
            assert cells[:8] == [vp["ref_codon"], vp["ref_amino_acid"], str(vp["ref_position"]), aa["amino_acid"], vc["codon"],
                                 sc.fmt_percent(100.0 * vc["frequency"]), str(vp["coverage"]), vc["known_drm"]]
            assert [c == "x" for c in cells[8:]] == vc["haplotype_hit"]
            title = dict(kv.split("=") for kv in tr["attrs"]["title"].split())
            assert int(title["count"]) == vc["count"] and int(title["expected"]) == vc["expected"]
            assert float(title["pValue"]) == vc["pValue"] and float(title["log_pValue"]) == vc["log_pValue"]
            n_rows += 1
        for tr, vp in zip(crows, g["variant_positions"]):
            (mt,) = dom.find(tr, tag="table", class_="msa")
            got = dom.rows(mt)[1:]
            assert got == [[str(m["rel_pos"]), str(m["abs_pos"])] + [str(m[s]) for s in "ACGT-N"] + [m["wt"]] for m in vp["msa"]]
    assert n_rows >= 4
    # section 4
    (t,) = dom.find(tag="table", id="drug-table")
    want = [[dd["drug"], v["gene"], v["mutation"], v["codon"], sc.fmt_percent(100.0 * v["frequency"])] for dd in j["drug_summaries"] for v in dd["variants"]]
    assert dom.rows(t)[1:] == want
    # the haplotype block: read categories, positions, counts, codons, read names
    hb = j["haplotype"]
    (t,) = dom.find(tag="table", id="hap-categories")
    got = {tr["attrs"]["data-key"]: dom.text([c for c in tr["children"] if c["tag"] == "td"][0]) for tr in t["children"] if "data-key" in tr["attrs"]}
    assert got == {k: str(hb[k]) for k in ("reported_reads", "insufficient_coverage_reads", "damaged_reads", "marginal_gaps",
                                           "marginal_heteroduplexes", "marginal_partial")}
    (p,) = dom.find(tag="p", id="hap-positions")
    assert [dom.text(s) for s in dom.find(p, tag="span")] == [str(x) for x in hb["variant_positions_abs"]]
    (t,) = dom.find(tag="table", id="hap-table")
    trs = [tr for tr in t["children"] if tr["tag"] == "tr"][1:]
    assert len(trs) == len(haps)
    for tr, hp in zip(trs, haps):
        tds = [c for c in tr["children"] if c["tag"] == "td"]
        assert [dom.text(c).strip() for c in tds[:3]] == [hp["name"], sc.fmt_hap_percent(100.0 * hp["frequency"]), str(hp["reads"])]
        assert dom.text(tds[3]).split() == hp["codons"]
        assert [dom.text(s) for s in dom.find(tds[4], tag="span", class_="rn")] == hp["read_names"]
    # nothing of the JSON is left out: every top-level key has its section
    assert set(j) == {"input", "target_config", "genes", "drug_summaries", "haplotype"}


def test_consensus_keeps_in_frame_majority_insertions(tmp_path, oracle):
    """`juliet --consensus` = the documented scope of `fuse` (doc/FUSE.md:17-24): planted insertions — an in-frame one in
    80 % of the reads, an in-frame one in 30 %, a 4-base one in 90 %, and two in-frame majority ones 6 columns apart —
    against the known answer (the 96 % major clone is the reference) and against the oracle's rule on the same counts."""
    n, l, seed = 3000, 300, 6
    bam, cfg, fa = (str(tmp_path / x) for x in ("ins.bam", "ins.json", "ins.fasta"))
    plants = [(60, 3, 800), (90, 6, 300), (120, 4, 900), (150, 6, 850), (156, 3, 900), (240, 9, 700)]
    args = []
    for c, k, pm in plants:
        args += ["--insert", f"{c}:{k}:{pm}"]
    subprocess.check_call([SYNTH, "--reads", str(n), "--cols", str(l), "--seed", str(seed), *args, "-o", bam, "--config-out", cfg])
    subprocess.check_call([JULIET, "-c", cfg, "--consensus", fa, bam, str(tmp_path / "o.json")])
    seq = "".join(open(fa).read().splitlines()[1:])
    ref = synth.reference(seed, l)
    ins_seq = lambda c, k: "".join("ACGT"[(c * 7 + j * 3 + 1) & 3] for j in range(k))   # noqa: E731  juliet-synth's bases
    want = ""
    for c in range(l):
        if c in (60, 150, 240):            # in frame and in most reads; 156 is within 10 columns of 150
            want += ins_seq(c, dict((p[0], p[1]) for p in plants)[c])
        want += "ACGT"[ref[c]]
    assert seq == want
    # a smaller distance lets the second of the close pair in; the calls themselves ignore insertions (doc/JULIET.md:26-27)
    subprocess.check_call([JULIET, "-c", cfg, "--consensus", fa, "--ins-min-distance", "6", bam, str(tmp_path / "o2.json")])
    seq6 = "".join(open(fa).read().splitlines()[1:])
    assert len(seq6) == len(seq) + 3 and ins_seq(156, 3) in seq6
    # `fuse in.bam out.fasta` (doc/FUSE.md:26-31) is the same front end under its own name: no config, the covered window
    fuse_bin = os.path.join(os.path.dirname(JULIET), "fuse")
    fa2 = str(tmp_path / "fuse.fasta")
    subprocess.check_call([fuse_bin, bam, fa2])
    head, *body = open(fa2).read().splitlines()
    assert head.startswith(">consensus") and "".join(body) == want
    assert subprocess.run([fuse_bin, bam], capture_output=True).returncode == 1      # usage
    j1, j2 = json.load(open(tmp_path / "o.json")), json.load(open(tmp_path / "o2.json"))
    assert j1["genes"] == j2["genes"]
    rows = synth.rows(synth.SynthParams(seed=seed), l, 0, n, ref)
    exp = oracle.call(rows, np.array([(1, l + 1)], dtype=capi.GENE), refseq=ref)
    assert [(r[1], r[2]) for r in flat_variants(j1)] == [(e["codon_pos"], e["codon"]) for e in exp]


def _strip(j):
    """A JSON document without what legitimately differs between two runs of the same input."""
    j = json.loads(json.dumps(j))
    j["input"].pop("timestamp")
    j["input"].pop("command_line")
    return j


@pytest.mark.parametrize("extra", [("--mode-phasing",), (), ("--mode-phasing", "--drm-only"), ("--mode-phasing", "--max-perc", "90")])
def test_windows_give_the_json_of_one_window(sample, oracle, extra):
    """`juliet --windows K` (doc/JULIET.md:261-264: each gene is treated separately, so cutting the reference into column
    windows never shows): the BAM is decoded and uploaded once, K windows are called with the global Bonferroni factor,
    phasing runs across the windows — and the JSON is the one-window JSON, byte for byte outside the input block."""
    d, bam, cfg, rows, ref = sample
    one = run_juliet(d, bam, "-c", cfg, *extra, out="w1.json")
    for k in (2, 3, 8):
        many = run_juliet(d, bam, "-c", cfg, "--windows", str(k), *extra, out=f"w{k}.json")
        assert _strip(many) == _strip(one), k
    if extra == ("--mode-phasing",):       # ... and it is the oracle's answer
        genes = np.array([(1, L + 1)], dtype=capi.GENE)
        table = oracle.call(rows, genes, refseq=ref)
        exp = oracle.phase(rows, table)
        hb = one["haplotype"]
        assert [h["reads"] for h in hb["haplotypes"]] == exp["hap_count"].tolist()
        assert hb["reported_reads"] == exp["summary"]["reported_reads"] and hb["damaged_reads"] == exp["summary"]["damaged_reads"]


def test_windows_on_a_long_reference_with_overlapping_genes(tmp_path, oracle):
    """A reduced configs[3]: 20 000 reads x 10 kb, three genes in different frames (two overlap), eight windows on one GPU
    against the one-window run and the oracle."""
    n, l, seed = 20000, 10000, 11
    bam, cfg = str(tmp_path / "long.bam"), str(tmp_path / "cfg.json")
    subprocess.check_call([SYNTH, "--reads", str(n), "--cols", str(l), "--seed", str(seed), "--minor-permille", "60", "50", "40", "30",
                           "-o", bam, "--config-out", cfg])
    c = json.load(open(cfg))
    c["genes"] = [dict(name="g0", begin=1, end=3001, drms=[]), dict(name="g1", begin=2900, end=7100, drms=[]),
                  dict(name="g2", begin=7102, end=10000, drms=[])]
    json.dump(c, open(cfg, "w"))
    one = run_juliet(tmp_path, bam, "-c", cfg, "--mode-phasing", out="one.json")
    many = run_juliet(tmp_path, bam, "-c", cfg, "--mode-phasing", "--windows", "8", out="many.json")
    assert _strip(many) == _strip(one)
    sp = synth.SynthParams(seed=seed, minor_permille=(60, 50, 40, 30))
    ref = synth.reference(seed, l)
    rows = synth.rows(sp, l, 0, n, ref)
    genes = np.array([(1, 3001), (2900, 7100), (7102, 10000)], dtype=capi.GENE)
    table = oracle.call(rows, genes, refseq=ref)
    got = flat_variants(many)
    assert len(got) == len(table) >= 4
    for (gi, pos, cod, vc, vp, aa), e in zip(got, table):
        assert (gi, pos, cod) == (e["gene"], e["codon_pos"], e["codon"]) and vc["count"] == e["count"] and vp["coverage"] == e["coverage"]
    exp = oracle.phase(rows, table)
    assert [h["reads"] for h in many["haplotype"]["haplotypes"]] == exp["hap_count"].tolist()
    assert len(many["haplotype"]["variant_positions_abs"]) == exp["summary"]["n_positions"]


def test_two_ranks_on_one_device(sample, tmp_path):
    """`--devices a,b` starts one rank (thread) per device, each with consecutive windows of its own.  One GPU per box
    here, so the ranks share device 0 — which RCCL refuses, so the ranks exchange in process (device copies between their
    buffers: `--exchange inproc`, the default when a device is named twice).  With and without phasing, two and three ranks:
    the JSON is the one-window JSON.  Asking for RCCL on one device fails loudly, not with a hang.
    (The rank threads cannot run under ThreadSanitizer: a TSan build of the front end dies at start-up once the GPU runtime
    maps its apertures, and the container that runs the sanitizer tests has no GPU, so the ranks never start there.  What
    they share is small: disjoint ranges of one column-count array and a job record each.)"""
    d, bam, cfg, rows, ref = sample
    one = run_juliet(d, bam, "-c", cfg, out="r1.json")
    for k in (2, 5):
        two = run_juliet(d, bam, "-c", cfg, "--windows", str(k), "--devices", "0,0", out=f"r{k}.json")
        assert _strip(two) == _strip(one)
    onep = run_juliet(d, bam, "-c", cfg, "--mode-phasing", out="p1.json")
    assert onep["haplotype"]
    for k, devs in ((2, "0,0"), (5, "0,0"), (7, "0,0,0")):
        twop = run_juliet(d, bam, "-c", cfg, "--mode-phasing", "--windows", str(k), "--devices", devs, out=f"p{k}.json")
        assert _strip(twop) == _strip(onep)
    out = str(tmp_path / "t.json")
    r = subprocess.run([JULIET, "-c", cfg, "--mode-phasing", "--windows", "4", "--devices", "0,0", "--exchange", "rccl", bam, out],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 3 and "communicator" in r.stderr
