"""MSAs built from the reference's printed known answers (tests/golden/appendix_a.json = the screenshots under
/root/reference/doc/img, and the three phasing scenarios of the FAQ, doc/JULIET.md:278-288, 356-366).

The inputs that produced the screenshots are not in the reference snapshot, so these are the smallest inputs that
must reproduce what each screenshot prints: for a printed variant row the codon count is the integer whose displayed
percentage equals the printed one at the printed coverage.  Test infrastructure: the same matrices go through the
oracle (CPU tests) and through the HIP path (GPU tests).
"""
import json
import math
import os

import numpy as np

from minorseq_amd import msa

HERE = os.path.dirname(os.path.abspath(__file__))
N, G, M = msa.SYM_MASK, msa.SYM_GAP, msa.SYM_NONE


def load_fixture():
    with open(os.path.join(HERE, "golden", "appendix_a.json")) as f:
        return json.load(f)


def fmt_percent(x):
    """Percentage as the reference's HTML prints it: two significant digits, TRUNCATED (docs/SPEC.md §6).
    Rounding to nearest cannot produce 12 of the printed rows (e.g. 0.91 % of 2946 reads has no integer count)."""
    if x <= 0:
        return "0"
    e = math.floor(math.log10(x))
    f = 10.0 ** (e - 1)
    return f"{math.floor(x / f + 1e-9) * f:.10g}"


def fmt_hap_percent(x):
    """Haplotype percentage: one decimal, rounded (the printed columns sum to 100.0), trailing zero dropped."""
    return f"{round(x + 1e-12, 1):g}"


def count_for(pct, cov):
    """The count nearest to pct % of cov whose displayed percentage is the printed string."""
    target = float(pct) * cov / 100.0
    best = None
    for c in range(max(1, int(target) - 40), min(cov, int(target) + 40) + 1):
        if fmt_percent(100.0 * c / cov) == pct and (best is None or abs(c - target) < abs(best - target)):
            best = c
    if best is None:
        raise ValueError(f"no count displays as {pct} % of {cov}")
    return best


def codon_cols(codon):
    return ["ACGT".index(ch) for ch in codon]


def table_positions(table):
    """[(gene, aa_pos, ref_codon, coverage, [(var_codon, count, printed_percent)])] in printed order."""
    out = []
    for g in table["genes"]:
        for r in g["rows"]:
            key = (g["name"], r[2])
            if out and out[-1][0] == key[0] and out[-1][1] == key[1]:
                out[-1][4].append((r[4], count_for(r[5], r[6]), r[5]))
            else:
                out.append([g["name"], r[2], r[0], r[6], [(r[4], count_for(r[5], r[6]), r[5])]])
    return out


def table_msa(table):
    """One codon per printed position, laid out as consecutive codons of ONE gene in frame 0 (the printed genes and
    positions only label the rows).  Returns rows uint8[n][3P], reference codes[3P], expected [(count, coverage)]."""
    pos = table_positions(table)
    n = max(p[3] for p in pos) + 37
    rows = np.zeros((n, 3 * len(pos)), dtype=np.uint8)
    ref = np.zeros(3 * len(pos), dtype=np.uint8)
    for i, (_, _, rc, cov, variants) in enumerate(pos):
        ref[3 * i: 3 * i + 3] = codon_cols(rc)
        rows[:, 3 * i: 3 * i + 3] = codon_cols(rc)
        # reads outside the coverage: a filtered base (N) in one of the three codon columns
        shift = (i * 131) % n
        out_cov = (np.arange(n - cov) + shift) % n
        rows[out_cov, 3 * i + (i % 3)] = N
        inside = np.setdiff1d(np.arange(n), out_cov)
        at = min((i * 97) % 11, cov - sum(c for _, c, _ in variants))
        for vc, cnt, _ in variants:
            rows[np.ix_(inside[at: at + cnt], [3 * i, 3 * i + 1, 3 * i + 2])] = codon_cols(vc)
            at += cnt
        assert at <= cov
    return rows, ref, pos


# ---------------------------------------------------------------------------------------------- FAQ scenarios
def abl_nohaplotype():
    """doc/JULIET.md:278-283 + juliet_abl-nohaplotype.png: "all reads associated to that variant contain a
    frame-shift deletion and thus won't be reported".  Four printed rows at ABL1 codons 217, 223 (two codons), 229;
    haplotype A = 100 % carries only A217A.  Every read with one of the three minor codons has a deletion in another
    variant codon; so have the reads that keep the reference codon at 217."""
    t = load_fixture()["tables"]["abl_nohaplotype"]
    pos = table_positions(t)
    n = 2500
    rows = np.zeros((n, 9), dtype=np.uint8)
    ref = np.zeros(9, dtype=np.uint8)
    for i, p in enumerate(pos):
        ref[3 * i: 3 * i + 3] = codon_cols(p[2])
        rows[:, 3 * i: 3 * i + 3] = codon_cols(p[2])
    cov = [p[3] for p in pos]                      # 2289, 2401, 2077
    (c217, n217, _), = pos[0][4]
    (cA, nA, _), (cP, nP, _) = pos[1][4]
    (c229, n229, _), = pos[2][4]
    # major: everybody carries GCG at 217 except `keep` reads that keep the reference codon
    rows[:, 0:3] = codon_cols(c217)
    keep = cov[0] - n217
    used = 0

    def take(k):
        nonlocal used
        idx = np.arange(used, used + k)
        used += k
        return idx

    r_keep = take(keep)
    rows[r_keep, 0:3] = codon_cols(pos[0][2])
    rows[r_keep, 7] = G                            # ... with a deletion in codon 229
    r_a, r_p = take(nA), take(nP)
    rows[np.ix_(r_a, [3, 4, 5])] = codon_cols(cA)
    rows[np.ix_(r_p, [3, 4, 5])] = codon_cols(cP)
    rows[r_a, 6] = G                               # frame-shift deletion in codon 229
    rows[r_p, 8] = G
    r_f = take(n229)
    rows[np.ix_(r_f, [6, 7, 8])] = codon_cols(c229)
    rows[r_f, 1] = G                               # deletion in codon 217
    # remaining damage to meet the printed coverages: deletions / filtered bases in otherwise wild-type reads
    gaps = [int((rows[:, 3 * i: 3 * i + 3] > 3).any(axis=1).sum()) for i in range(3)]
    for i in range(3):
        need = n - cov[i] - gaps[i]
        assert need >= 0
        idx = take(need)
        rows[idx, 3 * i + 1] = N if i == 1 else G
    assert used < n - 100
    expect = dict(calls=[(c217, n217, cov[0]), (cA, nA, cov[1]), (cP, nP, cov[1]), (c229, n229, cov[2])],
                  n_haplotypes=1, hit=[[1], [0], [0], [0]], reported=n - used)
    return rows, ref, expect


def no_haplotype_columns():
    """doc/JULIET.md:285-288: "each and every read has at least one deletion in one of the identified variant codon
    positions" -> phasing is on, variants are called, no haplotype column exists."""
    n = 3000
    rows = np.zeros((n, 6), dtype=np.uint8)
    ref = np.zeros(6, dtype=np.uint8)
    rows[:60, 0] = 2          # GAA in 2 % at codon 1 ...
    rows[1500:1560, 4] = 1    # ... ACA in 2 % at codon 2
    rows[:1500, 5] = G        # first half: deletion in codon 2; second half: deletion in codon 1
    rows[1500:, 2] = G
    expect = dict(calls=[("GAA", 60, 1500), ("ACA", 60, 1500)], n_haplotypes=0, reported=0, damaged=n, marginal_gap=n)
    return rows, ref, expect


MAJOR_MINORS = ("ATG", "AAA", "TAT", "GGA", "ACC")   # reference codons of M41 K65 Y181 G190 T215


def major_dilution():
    """doc/JULIET.md:356-366 + juliet_major-before.png / juliet_major-after.png: "major calls dilute phased minor
    variant haplotypes below the threshold"; with --max-perc 90 the minors phase into A 95.8 / B 1.1 {Y181C+G190A} /
    C 1.1 {K65R} / D 1 {T215Y} / E 1 {M41L}.  Positions (column order): the 13 printed rows of `before` (6 Protease,
    7 RT incl. M41L and K65R) followed by Y181C, G190A, T215Y of `after`; coverages as printed.
    Read classes (3000 reads):
      A      1200  every major, no minor, clean everywhere                       -> the one haplotype of `before`
      B..E   24 + 24 + 22 + 22 minor carriers, clean at the five minor positions; 8 of each clean everywhere (below the
             10-read threshold of `before`), the others lose a base at one major position
      KEEP   reads that keep the reference codon at a 99 % major; damaged at the next major position
      DMG    818 reads damaged at a minor position (so `after` reports 2182 = 2090 + 24 + 24 + 22 + 22 reads)
      TAIL   the rest: wild type at the minors, masked at majors as the printed coverages require
    Returns rows, ref, positions [(gene, aa, ref codon, variant codon, count, coverage)], indices of the minor positions,
    and the printed haplotype percentages of `after`."""
    fx = load_fixture()["tables"]
    before = table_positions(fx["major_before"])
    after = table_positions(fx["major_after"])
    by_aa_after = {p[1]: p for p in after}
    pos = []
    for p in before:
        q = by_aa_after[p[1]] if (p[0] == "Reverse Transcriptase" and p[1] in (41, 65)) else p
        assert q[3] == p[3]                         # M41L / K65R print the same coverage in both screenshots
        pos.append((p[0], p[1], p[2], q[4][0][0], q[4][0][1], q[3]))
    for aa in (181, 190, 215):
        q = by_aa_after[aa]
        pos.append((q[0], aa, q[2], q[4][0][0], q[4][0][1], q[3]))
    n, P = 3000, len(pos)
    minor = {p[1]: i for i, p in enumerate(pos) if p[0] == "Reverse Transcriptase" and p[1] in (41, 65, 181, 190, 215)}
    majors = [i for i in range(P) if i not in minor.values()]
    rows = np.zeros((n, 3 * P), dtype=np.uint8)
    ref = np.zeros(3 * P, dtype=np.uint8)
    for i, p in enumerate(pos):
        ref[3 * i: 3 * i + 3] = codon_cols(p[2])
        rows[:, 3 * i: 3 * i + 3] = codon_cols(p[3] if i in majors else p[2])
    need = [n - p[5] for p in pos]
    cursor = [0]

    def take(k):
        idx = np.arange(cursor[0], cursor[0] + k)
        cursor[0] += k
        return idx

    def put(reads, i, codon):
        rows[np.ix_(reads, [3 * i, 3 * i + 1, 3 * i + 2])] = codon_cols(codon)

    def mask(reads, i):
        assert (rows[reads, 3 * i + 1] < 4).all()
        rows[reads, 3 * i + 1] = N
        need[i] -= len(reads)

    take(1200)                                                           # A
    phased = {}
    for k, (name, aas, cnt) in enumerate((("B", (181, 190), 24), ("C", (65,), 24), ("D", (215,), 22), ("E", (41,), 22))):
        r = take(cnt)
        for aa in aas:
            put(r, minor[aa], pos[minor[aa]][3])
            phased[aa] = cnt
        for j, read in enumerate(r[8:]):
            mask(np.array([read]), majors[(j + 3 * k) % len(majors)])
    for k, i in enumerate(majors):                                       # KEEP
        r = take(pos[i][5] - pos[i][4])
        if len(r):
            put(r, i, pos[i][2])
            mask(r, majors[(k + 1) % len(majors)])
    dmg = take(818)                                                      # DMG: minor-position damage, largest first
    at = 0
    for aa in sorted(minor, key=lambda a: -need[minor[a]]):
        i = minor[aa]
        k = need[i]
        mask(dmg[(at + np.arange(k)) % len(dmg)], i)
        at += k
    assert at >= len(dmg)
    # carriers beyond the phased ones, to reach the printed counts: reads of DMG that are in the coverage of their
    # own position (masked at K65 only / at M41 only)
    only_k65 = dmg[(rows[dmg][:, [3 * minor[a] + 1 for a in (41, 181, 190, 215)]] < 4).all(axis=1)]
    only_m41 = dmg[(rows[dmg][:, [3 * minor[a] + 1 for a in (65, 181, 190, 215)]] < 4).all(axis=1)]
    o = 0
    for aa in (41, 181, 190, 215):
        extra = pos[minor[aa]][4] - phased[aa]
        put(only_k65[o: o + extra], minor[aa], pos[minor[aa]][3])
        o += extra
    put(only_m41[: pos[minor[65]][4] - phased[65]], minor[65], pos[minor[65]][3])
    tail = np.arange(cursor[0], n)                                       # TAIL: the majors' remaining damage
    at = 0
    for i in majors:
        k = need[i]
        assert 0 <= k <= len(tail), (i, k)
        mask(tail[(at + np.arange(k)) % len(tail)], i)
        at += k
    assert all(x == 0 for x in need), need
    return rows, ref, pos, minor, fx["major_after"]["haplotype_percent"]


def hiv_phasing():
    """juliet_hiv-phasing.png (doc/JULIET.md:205): nine variants over three genes, haplotype columns A..I that are
    GLOBAL across the genes, percentages 92.5 1.2 1.2 1 1 0.8 0.8 0.8 0.7 (sum 100.0), A = wild type, C carries
    Y181C and G190A together.  3000 reads; 2300 reported (the tooltip of juliet_haplotype-perc-tooltip.png puts the
    denominator between 2160 and 2348).  Genes are laid out as three ORFs of 1 + 7 + 1 codons in ONE window.
    Returns rows, ref, genes (1-based [begin, end)), positions, expected haplotypes [(count, {variant row indices})]."""
    t = load_fixture()["tables"]["hiv_phasing"]
    pos = table_positions(t)
    P = len(pos)
    names = [h for g in t["genes"] for r in g["rows"] for h in r[8]]          # haplotype letter per row
    by_hap = {}
    for i, h in enumerate(names):
        by_hap.setdefault(h, set()).add(i)
    counts = {"B": 28, "C": 27, "D": 24, "E": 23, "F": 19, "G": 18, "H": 18, "I": 16}
    n, reported = 3000, 2300
    haps = [(reported - sum(counts.values()), set())] + [(counts[h], by_hap[h]) for h in "BCDEFGHI"]
    rows = np.zeros((n, 3 * P), dtype=np.uint8)
    ref = np.zeros(3 * P, dtype=np.uint8)
    for i, p in enumerate(pos):
        ref[3 * i: 3 * i + 3] = codon_cols(p[2])
        rows[:, 3 * i: 3 * i + 3] = codon_cols(p[2])
    need = [n - p[3] for p in pos]
    cur = 0
    carried = [0] * P
    for cnt, members in haps:
        r = np.arange(cur, cur + cnt)
        cur += cnt
        for i in members:
            rows[np.ix_(r, [3 * i, 3 * i + 1, 3 * i + 2])] = codon_cols(pos[i][4][0][0])
            carried[i] += cnt
    assert cur == reported
    # every remaining read is damaged somewhere.  First the carriers beyond the phased ones (a base filtered at
    # the position with the largest remaining need), then wild-type reads, cycling over the positions.
    for i in range(P):
        extra = pos[i][4][0][1] - carried[i]
        assert extra >= 0
        r = np.arange(cur, cur + extra)
        cur += extra
        rows[np.ix_(r, [3 * i, 3 * i + 1, 3 * i + 2])] = codon_cols(pos[i][4][0][0])
        j = max((k for k in range(P) if k != i), key=lambda k: need[k])
        rows[r, 3 * j + 2] = N
        need[j] -= extra
    rest = np.arange(cur, n)
    assert sum(need) >= len(rest) and max(need) <= len(rest)
    at = 0
    for i in sorted(range(P), key=lambda k: -need[k]):
        idx = rest[(at + np.arange(need[i])) % len(rest)]
        rows[idx, 3 * i] = N if i % 2 else G
        at += need[i]
        need[i] = 0
    assert (rows[rest] > 3).any(axis=1).all()
    # three ORFs in one window: Protease = codon 0, RT = codons 1..7, Integrase = codon 8
    genes = [(1, 4), (4, 25), (25, 28)]
    return rows, ref, genes, pos, haps
