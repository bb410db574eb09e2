"""Host-side front end (minorseq_amd/host/): BAM writer + BGZF/BAM reader + CIGAR walk, target-config parsing,
DRM grammar, --region.  CPU only: uses the tool's --dump-msa / --dump-config diagnostics, which never touch a GPU."""
import json
import os
import subprocess

import numpy as np
import pytest

from minorseq_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "minorseq_amd", "bin")
JULIET = os.path.join(BIN, "juliet")
SYNTH = os.path.join(BIN, "juliet-synth")


@pytest.fixture(scope="module", autouse=True)
def built():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "minorseq_amd", "csrc")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "minorseq_amd", "host")])


def read_msa(path):
    raw = open(path, "rb").read()
    n, l, wb = (int(x) for x in np.frombuffer(raw[:24], dtype=np.uint64))
    return np.frombuffer(raw[24:], dtype=np.uint8).reshape(n, l), wb


@pytest.mark.parametrize("n,l,seed,partial,offset", [(300, 90, 3, 0.0, 0), (700, 300, 5, 0.3, 0), (200, 120, 9, 0.5, 250)])
def test_bam_roundtrip_equals_generator(tmp_path, n, l, seed, partial, offset):
    bam, cfg, msa = (str(tmp_path / x) for x in ("s.bam", "s.json", "s.msa"))
    subprocess.check_call([SYNTH, "--reads", str(n), "--cols", str(l), "--seed", str(seed), "--partial", str(partial),
                           "--ref-offset", str(offset), "-o", bam, "--config-out", cfg])
    subprocess.check_call([JULIET, "-c", cfg, "--dump-msa", msa, bam])
    rows, wb = read_msa(msa)
    exp = synth.rows(synth.SynthParams(seed=seed, partial_rate=partial), l, 0, n)
    # the window is the gene plus the -3..+5 context, clipped to the reference
    assert wb == max(0, offset - 3)
    assert (rows[:, offset - wb: offset - wb + l] == exp).all()
    assert (rows[:, : offset - wb] == 6).all()          # nothing aligned before the window start
    # without a config the window is the covered range and the gene is "unknown"
    subprocess.check_call([JULIET, "--dump-msa", msa, bam])
    rows2, wb2 = read_msa(msa)
    assert rows2.shape[0] == n and wb2 <= offset + l


def test_bgzf_multi_block(tmp_path):
    """> 64 KiB of records => several BGZF blocks; the reader must cross member boundaries."""
    bam, msa = str(tmp_path / "big.bam"), str(tmp_path / "big.msa")
    subprocess.check_call([SYNTH, "--reads", "400", "--cols", "3000", "--seed", "2", "-o", bam])
    assert os.path.getsize(bam) > 100_000
    subprocess.check_call([JULIET, "--dump-msa", msa, bam])
    rows, _ = read_msa(msa)
    assert (rows[:, :3000] == synth.rows(synth.SynthParams(seed=2), 3000, 0, 400)).all()


def test_target_config_grammar_and_region(tmp_path):
    cfg = {"genes": [{"begin": 2550, "end": 2700, "name": "Reverse Transcriptase",
                      "drms": [{"name": "fancy drug", "positions": ["M41L"]},
                               {"name": "ATV/r", "positions": ["V32I", "L33", "46IL", "I54VTALM", "V82ATFS", "84"]}]}],
           "referenceName": "my seq", "referenceSequence": "TGGAAGGGCT", "version": "v", "databaseVersion": "DrugDB"}
    p, out = str(tmp_path / "hiv.json"), str(tmp_path / "out.json")
    json.dump(cfg, open(p, "w"))
    subprocess.check_call([JULIET, "-c", p, "--dump-config", out])
    got = json.load(open(out))
    assert got["referenceName"] == "my seq" and got["referenceLength"] == 10 and got["databaseVersion"] == "DrugDB"
    assert got["genes"][0]["drms"][1]["positions"] == ["V32I", "L33", "46IL", "I54VTALM", "V82ATFS", "84"]
    assert got["effective_genes"] == [{"name": "Reverse Transcriptase", "begin": 2550, "end": 2700, "first_codon": 0}]
    # --region snaps inward to the gene's own frame (doc/JULIET.md:270-271)
    subprocess.check_call([JULIET, "-c", p, "-r", "2560-2650", "--dump-config", out])
    eff = json.load(open(out))["effective_genes"][0]
    assert eff == {"name": "Reverse Transcriptase", "begin": 2562, "end": 2650, "first_codon": 4}
    # predefined configs (doc/JULIET.md:118-126)
    subprocess.check_call([JULIET, "-c", "ABL1", "--dump-config", out])
    abl = json.load(open(out))
    assert abl["genes"][0]["begin"] == 193 and abl["genes"][0]["end"] == 3585 and len(abl["genes"][0]["drms"]) == 4
    r = subprocess.run([JULIET, "-c", "HIV", "--dump-config", out], capture_output=True, text=True)
    assert r.returncode == 2 and "not bundled" in r.stderr
    bad = str(tmp_path / "bad.json")
    json.dump({"genes": [{"begin": 1, "end": 10, "name": "g", "drms": [{"name": "d", "positions": ["M4!"]}]}]}, open(bad, "w"))
    assert subprocess.run([JULIET, "-c", bad, "--dump-config", out], capture_output=True).returncode == 2


def test_cli_usage_errors(tmp_path):
    assert subprocess.run([JULIET], capture_output=True).returncode == 1
    assert subprocess.run([JULIET, "--frobnicate", "a.bam", "o.json"], capture_output=True).returncode == 1
    assert subprocess.run([JULIET, "a.bam", "o.txt"], capture_output=True).returncode == 1       # suffix selects the format
    r = subprocess.run([JULIET, str(tmp_path / "missing.bam"), str(tmp_path / "o.json")], capture_output=True, text=True)
    assert r.returncode == 2 and "cannot open" in r.stderr
    assert subprocess.run([JULIET, "--version"], capture_output=True).returncode == 0


def test_rich_qv_tracks_mask_bases(tmp_path):
    """doc/JULIET.md:256-259: bases are filtered on the individual QV tracks (ccs --richQVs) and show up as N.
    The generator writes filtered bases as ordinary letters with a poor `sq` track; --min-qv turns them into N."""
    n, l, seed = 400, 150, 21
    bam, msa_out = str(tmp_path / "rich.bam"), str(tmp_path / "rich.msa")
    subprocess.check_call([SYNTH, "--reads", str(n), "--cols", str(l), "--seed", str(seed), "--rich-qv", "-o", bam])
    exp = synth.rows(synth.SynthParams(seed=seed), l, 0, n)
    subprocess.check_call([JULIET, "--min-qv", "10", "--dump-msa", msa_out, bam])
    rows, _ = read_msa(msa_out)
    assert (rows[:, :l] == exp).all() and (exp == 5).sum() > 100
    # without the filter the masked bases read as the reference base (more false positives: doc/JULIET.md:273-276)
    subprocess.check_call([JULIET, "--dump-msa", msa_out, bam])
    rows0, _ = read_msa(msa_out)
    ref = synth.reference(seed, l)
    assert (rows0[:, :l][exp == 5] == np.broadcast_to(ref, exp.shape)[exp == 5]).all()
    assert (rows0[:, :l][exp != 5] == exp[exp != 5]).all()


def test_quality_tracks_fold_like_the_byte_by_byte_rule(tmp_path):
    """msa_builder.hpp effective_quals (sixteen bases an instruction) against the rule it restates, a base at a time: the lowest of
    QUAL (0xFF = absent) and the dq / iq / sq characters - 33 that the record holds for the base — tracks shorter than the read,
    characters below '!' (they wrap to large values), every length from 0 to 70."""
    import textwrap
    src = tmp_path / "fold.cpp"
    src.write_text(textwrap.dedent(r"""
        #include <cstdio>
        #include <random>
        #include "juliet_hip.h"
        #include "msa_builder.hpp"
        using namespace jlhost;
        int main() {
            std::mt19937 g(5);
            for (int rep = 0; rep < 4000; ++rep) {
                BamRecord r;
                const size_t n = g() % 71;
                r.qual.resize(n);
                for (auto &q : r.qual) q = (g() % 4 == 0) ? 0xFF : (uint8_t)(g() % 94);
                for (std::string *t : {&r.dq, &r.iq, &r.sq}) {
                    const size_t m = (g() % 3 == 0) ? g() % (n + 1) : (g() % 5 == 0 ? 0 : n);
                    t->resize(m);
                    for (auto &c : *t) c = (char)(g() % 7 == 0 ? g() % 256 : 33 + g() % 94);
                }
                std::vector<uint8_t> got, want(n);
                effective_quals(r, got);
                for (size_t i = 0; i < n; ++i) {
                    uint8_t q = r.qual[i];
                    for (const std::string *t : {&r.dq, &r.iq, &r.sq})
                        if (i < t->size()) {
                            const uint8_t v = (uint8_t)((*t)[i] - 33);
                            if (q == 0xFF || v < q) q = v;
                        }
                    want[i] = q;
                }
                if (got != want) { printf("differs at repetition %d\n", rep); return 1; }
            }
            printf("ok\n");
            return 0;
        }
        """))
    exe = str(tmp_path / "fold")
    host = os.path.join(ROOT, "minorseq_amd", "host")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + host, "-I" + os.path.join(ROOT, "include"), "-o", exe, str(src), "-lz", "-lpthread"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr


def test_min_rq_filter_reads_the_rq_tag(tmp_path):
    """doc/JULIET.md:56: filtering on predicted accuracy is left to the user; --min-rq applies it from the rq tag.
    The generator writes rq = 0.999 on every read."""
    bam, msa_out = str(tmp_path / "rq.bam"), str(tmp_path / "rq.msa")
    subprocess.check_call([SYNTH, "--reads", "200", "--cols", "90", "--seed", "5", "-o", bam])
    subprocess.check_call([JULIET, "--min-rq", "0.99", "--dump-msa", msa_out, bam])
    rows, _ = read_msa(msa_out)
    assert rows.shape[0] == 200
    r = subprocess.run([JULIET, "--min-rq", "0.9995", "--dump-msa", msa_out, bam], capture_output=True, text=True)
    assert r.returncode == 2 and "no primary or supplementary alignments" in r.stderr


def test_from_rows_bam_roundtrip(tmp_path):
    """`juliet-synth --from-rows`: any by-row matrix (gaps, filtered bases, ragged ends, interior reference skips) as a
    PacBio-style BAM; `juliet --dump-msa` reads back exactly that matrix.  The scenario tests use this path."""
    rng = np.random.default_rng(3)
    n, l = 300, 60
    rows = rng.integers(0, 4, size=(n, l), dtype=np.uint8)
    rows[rng.random((n, l)) < 0.05] = 4
    rows[rng.random((n, l)) < 0.05] = 5
    for i in range(0, n, 7):
        rows[i, : rng.integers(1, 20)] = 6
        rows[i, l - rng.integers(1, 20):] = 6
    rows[5, 20:30] = 6                       # an interior stretch the read does not cover (cigar N)
    rows[:, 0][rows[:, 0] == 4] = 0          # a record cannot start or end with a deletion
    rows[:, -1][rows[:, -1] == 4] = 0
    for i in range(n):
        cov = np.nonzero(rows[i] != 6)[0]
        if rows[i, cov[0]] == 4:
            rows[i, cov[0]] = 1
        if rows[i, cov[-1]] == 4:
            rows[i, cov[-1]] = 2
    ref = rng.integers(0, 4, size=l, dtype=np.uint8)
    mpath, bam, cfg, back = (str(tmp_path / x) for x in ("m.msa", "m.bam", "m.json", "back.msa"))
    with open(mpath, "wb") as f:
        f.write(np.array([n, l, 0], dtype=np.uint64).tobytes())
        f.write(rows.tobytes())
    refs = "".join("ACGT"[b] for b in ref)
    subprocess.check_call([SYNTH, "--from-rows", mpath, "--ref", refs, "-o", bam])
    json.dump({"genes": [{"name": "g", "begin": 1, "end": l + 1}], "referenceName": "r", "referenceSequence": refs}, open(cfg, "w"))
    subprocess.check_call([JULIET, "-c", cfg, "--dump-msa", back, bam])
    got, wb = read_msa(back)
    assert wb == 0 and got.shape == (n, l) and (got == rows).all()


def _bgzf(raw, level=6, strategy=0, block=60000):
    """Uncompressed bytes -> BGZF (blocks of <= `block` bytes + the empty EOF block)."""
    import struct
    import zlib
    out = b""
    for o in list(range(0, len(raw), block)) + [None]:
        chunk = b"" if o is None else raw[o:o + block]
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
        body = co.compress(chunk) + co.flush()
        bsize = 12 + 6 + len(body) + 8 - 1
        out += struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, bsize) + body
        out += struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk))
    return out


def test_malformed_records_are_rejected(tmp_path):
    """Untrusted input: a cigar that consumes more bases than the record holds, a record whose variable-length
    parts overrun its block, an aux array that overruns it — each ends with exit status 2 and a message, never a crash."""
    import gzip
    import struct
    bam = str(tmp_path / "ok.bam")
    subprocess.check_call([SYNTH, "--reads", "3", "--cols", "60", "--seed", "1", "--sub", "0", "--del", "0", "--mask", "0", "-o", bam])
    raw = bytearray(gzip.open(bam, "rb").read())
    l_text, = struct.unpack_from("<i", raw, 4)
    o = 8 + l_text
    n_ref, = struct.unpack_from("<i", raw, o)
    o += 4
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", raw, o)
        o += 4 + l_name + 4
    rec0 = o                                   # block_size word of the first record
    block, = struct.unpack_from("<i", raw, rec0)
    l_name = raw[rec0 + 4 + 8]
    n_cigar, = struct.unpack_from("<H", raw, rec0 + 4 + 12)
    assert n_cigar == 1
    cig_at = rec0 + 4 + 32 + l_name

    def run(mutated, *flags):
        p = str(tmp_path / "bad.bam")
        open(p, "wb").write(_bgzf(bytes(mutated)))
        seq = subprocess.run([JULIET, *flags, "--dump-msa", str(tmp_path / "bad.msa"), p], capture_output=True, text=True)
        # the same file through the pipelined reader (an output is asked for, so the uploader exists; without a GPU a good
        # file ends with "no usable GPU", exit 3 — a bad one must fail in the decode, exit 2, with the same message)
        pipe = subprocess.run([JULIET, *flags, p, str(tmp_path / "bad.json")], capture_output=True, text=True)
        assert (pipe.returncode == 2) == (seq.returncode == 2), (seq.stderr, pipe.stderr)
        if seq.returncode == 2:
            assert seq.stderr.strip().splitlines()[-1] == pipe.stderr.strip().splitlines()[-1]
        return seq

    assert run(raw).returncode == 0            # the re-compressed original is fine
    m = bytearray(raw)                         # cigar 60= -> 90=: 30 bases more than the record holds
    struct.pack_into("<I", m, cig_at, (90 << 4) | 7)
    r = run(m)
    assert r.returncode == 2 and "cigar consumes 90 bases, the record holds 60" in r.stderr
    m = bytearray(raw)                         # l_seq larger than the block
    struct.pack_into("<I", m, rec0 + 4 + 16, 100000)
    r = run(m)
    assert r.returncode == 2 and "corrupt BAM record" in r.stderr
    m = bytearray(raw)                         # a B-array tag whose count overruns the record, in place of the rq tag
    aux = rec0 + 4 + 32 + l_name + 4 + 30 + 60
    assert bytes(m[aux:aux + 3]) == b"rqf"
    m[aux:aux + 7] = b"zzBc" + struct.pack("<I", 1 << 30)[:3]
    r = run(m, "--min-rq", "0.5")
    assert r.returncode == 2 and "truncated BAM aux" in r.stderr


def test_pipelined_reader_equals_sequential(tmp_path):
    """The pipelined decode (segments of BGZF blocks inflated and parsed on a pool, chunks in file order) against the
    sequential one on a file of many segments with records that straddle them, with and without the filters that read
    qualities and tags: a small C++ driver prints a checksum of everything either reader hands over."""
    import textwrap
    bam = str(tmp_path / "p.bam")
    subprocess.check_call([SYNTH, "--reads", "20000", "--cols", "900", "--seed", "5", "--partial", "0.2", "--rich-qv", "-o", bam])
    src = tmp_path / "cmp.cpp"
    src.write_text(textwrap.dedent(r"""
        #include <cstdio>
        #include "juliet_hip.h"
        #include "decode.hpp"
        using namespace jlhost;
        static uint64_t mix(uint64_t h, const void *p, size_t n) {
            const uint8_t *b = (const uint8_t *)p;
            for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
            return h;
        }
        struct Sum { uint64_t h = 1469598103934665603ull, reads = 0, chunks = 0; };
        static void eat(Sum &s, RecordArrays &c) {
            for (size_t r = 0; r < c.pos.size(); ++r) {
                s.h = mix(s.h, &c.pos[r], 4);
                s.h = mix(s.h, c.cigar.data() + c.cig_off[r], (c.cig_off[r + 1] - c.cig_off[r]) * 4);
                s.h = mix(s.h, c.seq4.data() + c.seq_off[r], c.seq_off[r + 1] - c.seq_off[r]);
                if (c.qual_off.size() > r + 1) s.h = mix(s.h, c.qual.data() + c.qual_off[r], c.qual_off[r + 1] - c.qual_off[r]);
                s.h = mix(s.h, c.names[r].data(), c.names[r].size());
            }
            s.reads += c.pos.size();
            ++s.chunks;
            c.clear();
        }
        int main(int argc, char **argv) {
            IngestOptions io;
            io.min_qv = (uint32_t)atoi(argv[2]);
            io.min_rq = atof(argv[3]);
            const bool want_qual = io.min_qv > 0;
            for (int pass = 0; pass < 2; ++pass) {
                Sum s;
                RecordSink sink;
                sink.give = [&](RecordArrays &c) { eat(s, c); };
                RecordArrays rec;
                std::vector<BamRef> refs;
                std::string text;
                const ReadExtent e = pass ? PipelinedBamReader::run(argv[1], io, -1, want_qual, sink, &refs, &text, 7)
                                          : collect_records(argv[1], io, -1, want_qual, rec, &refs, &text, &sink);
                printf("%llu %llu %lld %lld %d %zu %zu %llu\n", (unsigned long long)s.h, (unsigned long long)e.n_reads, (long long)e.min_pos,
                       (long long)e.max_end, e.ref_id, refs.size(), text.size(), (unsigned long long)s.reads);
            }
            return 0;
        }
        """))
    exe = str(tmp_path / "cmp")
    host = os.path.join(ROOT, "minorseq_amd", "host")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-I" + host, "-I" + os.path.join(ROOT, "include"), "-o", exe, str(src), "-lz", "-lpthread"])
    for qv, rq in (("0", "0"), ("12", "0"), ("0", "0.9985"), ("10", "0.999")):
        out = subprocess.run([exe, bam, qv, rq], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        a, b = out.stdout.strip().splitlines()
        assert a == b, (qv, rq, a, b)
        assert int(a.split()[1]) > 1000
    # the same records in BGZF files of other shapes: stored blocks (level 0), fixed and dynamic Huffman codes, run-length
    # and Huffman-only strategies, blocks of 300 bytes (hundreds of blocks per record: every record straddles) up to the
    # largest a BGZF block may hold — both readers, and the same checksum as the file they were made from
    import gzip
    import zlib
    small = str(tmp_path / "s.bam")
    subprocess.check_call([SYNTH, "--reads", "2500", "--cols", "900", "--seed", "6", "--partial", "0.2", "--rich-qv", "-o", small])
    ref_line = subprocess.run([exe, small, "0", "0"], capture_output=True, text=True).stdout.strip().splitlines()
    assert ref_line[0] == ref_line[1]
    inflated = gzip.decompress(open(small, "rb").read())
    for level, strategy, block in ((0, 0, 60000), (1, 0, 65280), (9, 0, 300), (6, zlib.Z_FIXED, 5000), (6, zlib.Z_RLE, 20000),
                                   (6, zlib.Z_HUFFMAN_ONLY, 65280), (4, zlib.Z_FILTERED, 1111)):
        v = str(tmp_path / "v.bam")
        open(v, "wb").write(_bgzf(inflated, level, strategy, block))
        out = subprocess.run([exe, v, "0", "0"], capture_output=True, text=True)
        assert out.returncode == 0, (level, strategy, block, out.stderr)
        a, b = out.stdout.strip().splitlines()
        assert a == b == ref_line[0], (level, strategy, block)
    # a file cut in the middle of a record, and one cut inside the header
    raw = open(bam, "rb").read()
    for cut, msg in ((len(raw) // 2, "truncated"), (40, "")):
        bad = str(tmp_path / "cut.bam")
        open(bad, "wb").write(raw[:cut])
        out = subprocess.run([exe, bad, "0", "0"], capture_output=True, text=True)
        assert out.returncode != 0


def test_bgzf_block_crc_is_checked(tmp_path):
    """A BGZF block whose inflated bytes do not match its CRC-32 (RFC 1952) ends the run with a message instead of being parsed
    as records; the carry-less-multiply CRC of the front end equals zlib's on every length and alignment
    (tests/cpp/crc_check.cpp)."""
    import struct
    host = os.path.join(ROOT, "minorseq_amd", "host")
    exe = str(tmp_path / "crc_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           os.path.join(ROOT, "tests", "cpp", "crc_check.cpp"), "-lz", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "0 mismatches" in out.stdout, out.stdout + out.stderr
    bam = str(tmp_path / "ok.bam")
    subprocess.check_call([SYNTH, "--reads", "400", "--cols", "300", "--seed", "3", "-o", bam, "--config-out", str(tmp_path / "c.json")])
    raw = bytearray(open(bam, "rb").read())
    # second BGZF block: flip a bit of its stored CRC (the deflate data still inflates to ISIZE bytes)
    bsize0 = struct.unpack_from("<H", raw, 16)[0] + 1
    bsize1 = struct.unpack_from("<H", raw, bsize0 + 16)[0] + 1
    raw[bsize0 + bsize1 - 8] ^= 0x10
    bad = str(tmp_path / "bad.bam")
    open(bad, "wb").write(raw)
    for extra in ([], ["--dump-msa", str(tmp_path / "m.bin")]):     # the pipelined reader (with a device) and the sequential one
        r = subprocess.run([JULIET, "-c", str(tmp_path / "c.json"), *extra, bad, str(tmp_path / "o.json")], capture_output=True, text=True)
        assert r.returncode != 0 and ("CRC" in r.stderr or "no HIP device" in r.stderr or "device" in r.stderr), r.stderr


def test_fast_inflate_equals_zlib(tmp_path):
    """minorseq_amd/host/fast_inflate.hpp (the decoder of the pipelined BAM reader) against zlib on the same raw DEFLATE
    streams, built with AddressSanitizer + UBSan: every level and strategy of zlib's deflate over seven kinds of input
    and thirteen sizes, wrong output sizes refused, then corrupted and truncated copies, which must get zlib's verdict
    and, where both accept, zlib's bytes (tests/cpp/inflate_check.cpp).  Then the blocks of a synthetic BAM."""
    host = os.path.join(ROOT, "minorseq_amd", "host")
    exe = str(tmp_path / "inflate_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-I" + host, os.path.join(ROOT, "tests", "cpp", "inflate_check.cpp"), "-lz", "-o", exe])
    out = subprocess.run([exe, "11", "8"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr + out.stdout
    assert out.stdout.startswith("ok:")
    bam = str(tmp_path / "b.bam")
    subprocess.check_call([os.path.join(ROOT, "minorseq_amd", "bin", "juliet-synth"), "--reads", "2000", "--cols", "900", "--seed", "4",
                           "-o", bam, "--config-out", str(tmp_path / "b.json")])
    out = subprocess.run([exe, "bench", bam], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "bytes agree" in out.stdout, out.stderr + out.stdout
