// juliet-synth — writes the synthetic aligned-CCS mixture of csrc/jl_synth.h as a PacBio-style BAM
// (cigar = X D only, no M: doc/JULIET.md:53) plus a matching target config, so the BAM-in / JSON-out
// surface can be exercised end to end.  Mixture semantics: doc/MIXDATA.md:9-22.
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "../csrc/jl_synth.h"
#include "bam.hpp"
#include "config.hpp"

using namespace jlhost;

int main(int argc, char **argv)
{
    uint64_t n_reads = 1000, seed = 1;
    uint32_t n_cols = 3000, ref_offset = 0;
    double sub = 1.75e-4, del = 1.3e-3, mask = 2.0e-2, partial = 0.0;
    uint32_t minor[4] = {10, 10, 10, 10};
    std::string out, cfg_out, from_rows, ref_string;
    struct Plant { uint32_t col, len, permille; };
    std::vector<Plant> plants;   // --insert col:len:permille — insertions before window column `col` in that share of the reads
    bool rich_qv = false;  // filtered bases keep their letter and get a low QV (BAM: the sq tag; --raw-out: the quality byte) instead of 'N'
    std::string raw_out;   // --raw-out file: the records as the arrays jl_records_append takes, no BAM (bench.py once_through)
    uint64_t ref_seed = 0;
    bool have_ref_seed = false;
    // --raw-out only: what an ingest must ignore or mask, drawn by hashes of (read, column) so that every run gives the same
    // records: insertions of 1-4 bases before a column (parts per million of the cells), soft and hard clips at the ends of
    // half the reads, and aligned bases with a poor quality (ppm) — the cells a QV threshold turns into N
    uint32_t ins_ppm = 0, low_qv_ppm = 0;
    bool clips = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto need = [&]() -> std::string { if (i + 1 >= argc) { std::cerr << a << " needs a value\n"; std::exit(1); } return argv[++i]; };
        if (a == "--reads") n_reads = std::stoull(need());
        else if (a == "--cols") n_cols = (uint32_t)std::stoul(need());
        else if (a == "--seed") seed = std::stoull(need());
        else if (a == "--sub") sub = std::stod(need());
        else if (a == "--del") del = std::stod(need());
        else if (a == "--mask") mask = std::stod(need());
        else if (a == "--partial") partial = std::stod(need());
        else if (a == "--minor-permille") { for (int k = 0; k < 4; ++k) minor[k] = (uint32_t)std::stoul(need()); }
        else if (a == "--ref-offset") ref_offset = (uint32_t)std::stoul(need());  // window starts here in a longer reference
        else if (a == "--rich-qv") rich_qv = true;
        else if (a == "--raw-out") raw_out = need();
        else if (a == "--ins-ppm") ins_ppm = (uint32_t)std::stoul(need());
        else if (a == "--low-qv-ppm") low_qv_ppm = (uint32_t)std::stoul(need());
        else if (a == "--clips") clips = true;
        else if (a == "--ref-seed") { ref_seed = std::stoull(need()); have_ref_seed = true; }   // reference drawn from another seed than the reads
        else if (a == "--insert") {
            const std::string v = need();
            Plant p;
            if (sscanf(v.c_str(), "%u:%u:%u", &p.col, &p.len, &p.permille) != 3) { std::cerr << "--insert wants col:len:permille\n"; return 1; }
            plants.push_back(p);
        }
        else if (a == "--from-rows") from_rows = need();   // a matrix in the --dump-msa format instead of the generator
        else if (a == "--ref") ref_string = need();        // its reference bases (ACGT), one per column
        else if (a == "-o") out = need();
        else if (a == "--config-out") cfg_out = need();
        else { std::cerr << "usage: juliet-synth --reads N --cols L --seed S [--partial p] [--minor-permille a b c d] [--ref-offset k] -o out.bam [--config-out cfg.json]\n"; return 1; }
    }
    if (out.empty() && raw_out.empty()) { std::cerr << "juliet-synth: -o out.bam (or --raw-out file) is required\n"; return 1; }
    if (!from_rows.empty()) {
        // Any by-row matrix (codes 0..6, header {n_reads, n_cols, win_begin} as written by `juliet --dump-msa`) as a
        // PacBio-style BAM: tests send hand-built alignments (the reference's printed scenarios) through the tool.
        std::ifstream f(from_rows, std::ios::binary);
        uint64_t hdr[3];
        if (!f.read((char *)hdr, sizeof hdr)) { std::cerr << "juliet-synth: cannot read " << from_rows << "\n"; return 2; }
        const uint64_t n = hdr[0], l = hdr[1], wb = hdr[2];
        std::vector<uint8_t> m((size_t)n * l);
        if (!f.read((char *)m.data(), (std::streamsize)m.size())) { std::cerr << "juliet-synth: short matrix\n"; return 2; }
        if (ref_string.size() != l) { std::cerr << "juliet-synth: --ref needs one base per column\n"; return 1; }
        const uint32_t rl = (uint32_t)(wb + l);
        const std::string header = "@HD\tVN:1.5\tSO:unknown\tpb:3.0.1\n@SQ\tSN:ref\tLN:" + std::to_string(rl) +
                                   "\n@RG\tID:rows\tPL:PACBIO\tPM:SEQUEL\tDS:READTYPE=CCS\n";
        BamWriter bw(out, header, {{"ref", rl}});
        for (uint64_t i = 0; i < n; ++i) {
            const uint8_t *row = m.data() + (size_t)i * l;
            uint64_t st = 0, en = l;
            while (st < l && row[st] == 6) ++st;
            while (en > st && row[en - 1] == 6) --en;
            if (st == en) continue;   // a read that covers nothing has no record
            BamRecord r;
            r.ref_id = 0;
            r.pos = (int32_t)(wb + st);
            r.mapq = 254;
            r.name = "rows/" + std::to_string(i) + "/ccs";
            r.rq = 0.999f;
            uint32_t run_op = 99, run_len = 0;
            auto flush = [&]() { if (run_len) r.cigar.push_back(run_len << 4 | run_op); run_len = 0; };
            for (uint64_t c = st; c < en; ++c) {
                const uint8_t sy = row[c];
                uint32_t op;
                if (sy == 4) op = CIG_D;
                else if (sy == 6) op = CIG_N;
                else {
                    const uint8_t rb = base_code(ref_string[c]);
                    op = (sy < 4 && sy == rb) ? CIG_EQ : CIG_X;
                    r.seq.push_back(sy < 4 ? sy : (uint8_t)4);
                    r.qual.push_back(93);
                }
                if (op != run_op) { flush(); run_op = op; }
                ++run_len;
            }
            flush();
            bw.write(r);
        }
        bw.close();
        return 0;
    }
    std::vector<uint8_t> ref(n_cols);
    jl_synth_reference(have_ref_seed ? ref_seed : seed, n_cols, ref.data());
    jl_synth_plan pl;
    jl_synth_make_plan(&pl, seed, n_cols, sub, del, mask, partial, minor, ref.data());

    if (!raw_out.empty()) {
        // The same reads as the arrays a BAM decoder hands to jl_records_append (include/juliet_hip.h): pos, cigar words,
        // cigar offsets, BAM's 4-bit packed bases, their byte offsets, one quality byte per base and its offsets.
        // File: 8 u64 {magic, n_reads, n_cigar, n_seq_bytes, n_qual, 0, 0, 0}, then the arrays in that order, each padded to 8 bytes.
        std::vector<int32_t> pos(n_reads);
        std::vector<uint32_t> cigar;
        std::vector<uint64_t> cig_off(n_reads + 1, 0), seq_off(n_reads + 1, 0), qual_off(n_reads + 1, 0);
        std::vector<uint8_t> seq4, qual;
        cigar.reserve(n_reads * 48);
        seq4.reserve(n_reads * (size_t)(n_cols / 2 + 1));
        qual.reserve(n_reads * (size_t)n_cols);
        static const uint8_t nt16[5] = {1, 2, 4, 8, 15};
        for (uint64_t i = 0; i < n_reads; ++i) {
            uint32_t hap, st, en;
            jl_synth_read(&pl, i, &hap, &st, &en);
            pos[i] = (int32_t)(ref_offset + st);
            uint32_t run_op = 99, run_len = 0;
            auto flush = [&]() { if (run_len) cigar.push_back(run_len << 4 | run_op); run_len = 0; };
            uint32_t nb = 0;
            uint8_t half = 0;
            auto push_base = [&](uint8_t code, uint8_t q) {
                if (nb & 1u) seq4.push_back((uint8_t)(half << 4 | code));
                else half = code;
                ++nb;
                qual.push_back(q);
            };
            const uint64_t hr = jl_splitmix64(seed * 0xD1B54A32D192ED03ull + 31ull * i + 5ull);
            if (clips && st < en) {
                if (hr & 1u) cigar.push_back((uint32_t)(1u + (hr >> 8) % 7u) << 4 | CIG_H);
                if (hr & 2u) {
                    const uint32_t k = 1u + (uint32_t)((hr >> 16) % 40u);
                    cigar.push_back(k << 4 | CIG_S);
                    for (uint32_t j = 0; j < k; ++j) push_base(nt16[(hr >> (24 + j % 20)) & 3u], 3);
                }
            }
            for (uint32_t c = st; c < en; ++c) {
                const uint64_t hc = (ins_ppm || low_qv_ppm) ? jl_splitmix64(seed * 0xA24BAED4963EE407ull + (uint64_t)i * n_cols + c) : 0u;
                if (ins_ppm && c > st && hc % 1000000u < ins_ppm) {
                    flush();
                    run_op = 99;
                    const uint32_t k = 1u + (uint32_t)((hc >> 32) & 3u);
                    cigar.push_back(k << 4 | CIG_I);
                    for (uint32_t j = 0; j < k; ++j) push_base(nt16[(hc >> (40 + 2 * j)) & 3u], 2);
                }
                const uint32_t sy = jl_synth_cell(&pl, i, c, hap, st, en, ref[c]);
                uint32_t op;
                if (sy == 4) op = CIG_D;
                else if (rich_qv && sy == 5) {
                    // as `ccs --richQVs` output looks (doc/JULIET.md:256-259, 273-276): the filtered base keeps its letter — a match
                    // in the cigar — and carries a quality below any threshold; the N appears when the ingest applies min_qv
                    op = CIG_EQ;
                    push_base(nt16[ref[c]], (uint8_t)(2u + (jl_splitmix64(seed + 977ull * i + c) & 7u)));
                } else {
                    op = (sy < 4 && sy == ref[c]) ? CIG_EQ : CIG_X;
                    push_base(nt16[sy < 4 ? sy : 4], (low_qv_ppm && (hc >> 20) % 1000000u < low_qv_ppm) ? (uint8_t)(4u + (hc >> 50) % 12u) : (uint8_t)93);
                }
                if (op != run_op) { flush(); run_op = op; }
                ++run_len;
            }
            flush();
            if (clips && st < en && (hr & 4u)) {
                const uint32_t k = 1u + (uint32_t)((hr >> 44) % 25u);
                cigar.push_back(k << 4 | CIG_S);
                for (uint32_t j = 0; j < k; ++j) push_base(nt16[(hr >> (3 + j % 30)) & 3u], 3);
            }
            if (nb & 1u) seq4.push_back((uint8_t)(half << 4));
            cig_off[i + 1] = cigar.size();
            seq_off[i + 1] = seq4.size();
            qual_off[i + 1] = qual.size();
        }
        std::ofstream f(raw_out, std::ios::binary);
        const uint64_t hdr[8] = {0x4A4C524157303031ull, n_reads, cigar.size(), seq4.size(), qual.size(), 0, 0, 0};
        auto put = [&](const void *p, size_t bytes) {
            static const char zero[8] = {0};
            f.write((const char *)p, (std::streamsize)bytes);
            if (bytes % 8) f.write(zero, (std::streamsize)(8 - bytes % 8));
        };
        put(hdr, sizeof hdr);
        put(pos.data(), pos.size() * 4);
        put(cigar.data(), cigar.size() * 4);
        put(cig_off.data(), cig_off.size() * 8);
        put(seq4.data(), seq4.size());
        put(seq_off.data(), seq_off.size() * 8);
        put(qual.data(), qual.size());
        put(qual_off.data(), qual_off.size() * 8);
        if (!f) { std::cerr << "juliet-synth: cannot write " << raw_out << "\n"; return 2; }
        if (out.empty()) return 0;
    }

    const uint32_t ref_len = ref_offset + n_cols;
    const std::string header = "@HD\tVN:1.5\tSO:unknown\tpb:3.0.1\n@SQ\tSN:synthetic_ref\tLN:" + std::to_string(ref_len) +
                               "\n@RG\tID:synth\tPL:PACBIO\tPM:SEQUEL\tDS:READTYPE=CCS\n";
    BamWriter bw(out, header, {{"synthetic_ref", ref_len}});
    BamRecord r;
    for (uint64_t i = 0; i < n_reads; ++i) {
        uint32_t hap, st, en;
        jl_synth_read(&pl, i, &hap, &st, &en);
        r = BamRecord();
        r.ref_id = 0;
        r.pos = (int32_t)(ref_offset + st);
        r.flag = 0;
        r.mapq = 254;
        r.name = "synth/" + std::to_string(i) + "/ccs";
        r.rq = 0.999f;
        uint32_t run_op = 99, run_len = 0;
        auto flush = [&]() { if (run_len) r.cigar.push_back(run_len << 4 | run_op); run_len = 0; };
        for (uint32_t c = st; c < en; ++c) {
            for (size_t pi = 0; pi < plants.size(); ++pi) {
                // an insertion of plants[pi].len bases (a fixed sequence derived from the column) before column c;
                // which reads carry it is a hash of (read, plant)
                const Plant &p = plants[pi];
                if (p.col != c || c == st) continue;
                if (jl_splitmix64(seed * 0x9E3779B9ull + 1000003ull * i + 7919ull * pi) % 1000u >= p.permille) continue;
                flush();
                run_op = 99;
                r.cigar.push_back(p.len << 4 | CIG_I);
                for (uint32_t j = 0; j < p.len; ++j) {
                    r.seq.push_back((uint8_t)((p.col * 7u + j * 3u + 1u) & 3u));
                    r.qual.push_back(93);
                    if (rich_qv) { r.sq.push_back((char)(33 + 60)); r.dq.push_back((char)(33 + 60)); r.iq.push_back((char)(33 + 60)); }
                }
            }
            const uint32_t s = jl_synth_cell(&pl, i, c, hap, st, en, ref[c]);
            uint32_t op;
            if (s == 4) op = CIG_D;
            else {
                op = ((s < 4 && s == ref[c]) || (rich_qv && s == 5)) ? CIG_EQ : CIG_X;
                if (rich_qv) {
                    // as `ccs --richQVs` output would look: a real base letter with a poor per-base QV track
                    r.seq.push_back(s < 4 ? (uint8_t)s : ref[c]);
                    r.sq.push_back(s < 4 ? (char)(33 + 60) : (char)(33 + 3));
                    r.dq.push_back((char)(33 + 60));
                    r.iq.push_back((char)(33 + 60));
                } else {
                    r.seq.push_back(s < 4 ? (uint8_t)s : (uint8_t)4);  // filtered base travels as 'N'
                }
                r.qual.push_back(93);
            }
            if (op != run_op) { flush(); run_op = op; }
            ++run_len;
        }
        flush();
        bw.write(r);
    }
    bw.close();

    if (!cfg_out.empty()) {
        static const char *b = "ACGT";
        std::string seq(ref_offset, 'A');
        for (uint32_t c = 0; c < n_cols; ++c) seq += b[ref[c]];
        Json cfg = Json::object();
        Json g = Json::object();
        g.set("name", Json::of("Synthetic ORF")).set("begin", Json::of(ref_offset + 1)).set("end", Json::of(ref_offset + 3 * (n_cols / 3) + 1));
        Json drms = Json::array();
        // the planted codon edits double as the "known" resistance mutations of a made-up drug
        Json d = Json::object();
        d.set("name", Json::of("synthetic drug"));
        Json ps = Json::array();
        for (int k = 0; k < JL_SYNTH_N_EDITS; ++k) {
            const uint32_t col = pl.edit_col[k], cs = col - col % 3;
            unsigned cod[3] = {ref[cs], ref[cs + 1], ref[cs + 2]};
            const char ra = translate(16 * cod[0] + 4 * cod[1] + cod[2]);
            cod[col % 3] = pl.edit_base[k];
            const char ma = translate(16 * cod[0] + 4 * cod[1] + cod[2]);
            ps.push(Json::of(std::string(1, ra) + std::to_string(cs / 3 + 1) + std::string(1, ma)));
        }
        d.set("positions", ps);
        drms.push(d);
        g.set("drms", drms);
        cfg.set("genes", Json::array().push(g));
        cfg.set("referenceName", Json::of("synthetic_ref")).set("referenceSequence", Json::of(seq));
        cfg.set("version", Json::of("juliet-synth seed " + std::to_string(seed)));
        cfg.set("databaseVersion", Json::of("synthetic DRM list"));
        std::string s;
        cfg.write(s);
        std::ofstream(cfg_out) << s << "\n";
    }
    return 0;
}
