// bam.hpp — BAM records in and out (SAM spec v1 binary layout), the subset juliet needs:
// primary + supplementary alignments of CCS reads with PacBio cigars (= X I D S H N, no M)
// (doc/JULIET.md:50-58).  Reader is streaming; writer exists for the synthetic generator.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "bgzf.hpp"

namespace jlhost {

enum CigarOp : uint32_t { CIG_M = 0, CIG_I = 1, CIG_D = 2, CIG_N = 3, CIG_S = 4, CIG_H = 5, CIG_P = 6, CIG_EQ = 7, CIG_X = 8 };

struct BamRef {
    std::string name;
    uint32_t length;
};

struct BamRecord {
    int32_t ref_id = -1;
    int32_t pos = -1;  // 0-based leftmost
    uint16_t flag = 0;
    uint8_t mapq = 0;
    std::string name;
    std::vector<uint32_t> cigar;  // len << 4 | op
    std::vector<uint8_t> seq;     // base codes 0..3 = ACGT, 4 = N/other
    std::vector<uint8_t> seq4;    // the same bases as stored in BAM (two 4-bit codes per byte, high nibble first)
    std::vector<uint8_t> qual;    // phred, 0xFF when absent
    float rq = -1.f;              // predicted accuracy tag (doc/JULIET.md:56), -1 when absent
    // rich QV tracks of `ccs --richQVs` (doc/JULIET.md:51-52, 256-259): per-base phred+33 strings, empty when absent
    std::string dq, iq, sq;
};

class BamReader {
public:
    explicit BamReader(const std::string &path) : in_(path)
    {
        char magic[4];
        if (!in_.read(magic, 4) || memcmp(magic, "BAM\1", 4) != 0) throw std::runtime_error(path + ": not a BAM file");
        int32_t l_text = rd<int32_t>();
        text_.resize((size_t)l_text);
        if (l_text) in_.read(&text_[0], (size_t)l_text);
        const int32_t n_ref = rd<int32_t>();
        for (int32_t i = 0; i < n_ref; ++i) {
            const int32_t l_name = rd<int32_t>();
            std::string nm((size_t)l_name, '\0');
            in_.read(&nm[0], (size_t)l_name);
            if (!nm.empty() && nm.back() == '\0') nm.pop_back();
            BamRef r{nm, (uint32_t)rd<int32_t>()};
            refs_.push_back(r);
        }
    }
    const std::string &header_text() const { return text_; }
    const std::vector<BamRef> &refs() const { return refs_; }

    // aux fields from offset o: `rq` (float) and the rich-QV strings dq / iq / sq are interpreted; everything else is
    // skipped by type
    // what a record's tags say that this reader uses: rq, and where the rich-QV tracks dq / iq / sq lie in the record (not copied)
    struct AuxViews {
        float rq = -1.f;
        const char *track[3] = {nullptr, nullptr, nullptr};   // dq, iq, sq
        size_t len[3] = {0, 0, 0};
    };
    static void scan_aux(const uint8_t *p, size_t o, size_t block, AuxViews &v)
    {
        v = AuxViews();
        while (o + 3 <= block) {
            const char t0 = (char)p[o], t1 = (char)p[o + 1], ty = (char)p[o + 2];
            o += 3;
            size_t len = 0;
            switch (ty) {
            case 'A': case 'c': case 'C': len = 1; break;
            case 's': case 'S': len = 2; break;
            case 'i': case 'I': case 'f': len = 4; break;
            case 'Z': case 'H': {     // to the terminating NUL (memchr: the rich-QV tracks are a byte per base, three times a read)
                const void *z = memchr(p + o, 0, block - o);
                len = (z ? (size_t)((const uint8_t *)z - (p + o)) : block - o) + 1;
                break;
            }
            case 'B': {
                if (o + 5 > block) throw std::runtime_error("truncated BAM aux array");
                const char sub = (char)p[o];
                uint32_t cnt; memcpy(&cnt, p + o + 1, 4);
                const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
                len = 5 + es * (size_t)cnt;
                break;
            }
            default: throw std::runtime_error("unknown BAM aux type");
            }
            if (o + len > block) throw std::runtime_error("truncated BAM aux field");
            if (t0 == 'r' && t1 == 'q' && ty == 'f') memcpy(&v.rq, p + o, 4);
            if (ty == 'Z' && t1 == 'q' && (t0 == 'd' || t0 == 'i' || t0 == 's')) {
                const int k = t0 == 'd' ? 0 : t0 == 'i' ? 1 : 2;
                v.track[k] = (const char *)p + o;
                v.len[k] = len - 1;
            }
            o += len;
        }
    }
    static void parse_aux(const uint8_t *p, size_t o, size_t block, BamRecord &r)
    {
        AuxViews v;
        scan_aux(p, o, block, v);
        r.rq = v.rq;
        r.dq.assign(v.track[0] ? v.track[0] : "", v.len[0]);
        r.iq.assign(v.track[1] ? v.track[1] : "", v.len[1]);
        r.sq.assign(v.track[2] ? v.track[2] : "", v.len[2]);
    }

    // the bytes of the next record (after its block_size word), in place when they lie inside the reader's inflated
    // batch; valid until the next call
    bool next_raw(const uint8_t *&p, size_t &len)
    {
        int32_t block = 0;
        if (const uint8_t *q = in_.peek(4)) memcpy(&block, q, 4);
        else if (!in_.read(&block, 4)) return false;
        if (block < 32) throw std::runtime_error("corrupt BAM record");
        len = (size_t)block;
        p = in_.peek(len);
        if (!p) {
            buf_.resize(len);
            if (!in_.read(buf_.data(), len)) throw std::runtime_error("truncated BAM record");
            p = buf_.data();
        }
        return true;
    }

    // unpack_seq = false leaves r.seq empty (callers that hand BAM's packed bases to the device need only seq4)
    bool next(BamRecord &r, bool unpack_seq = true)
    {
        const uint8_t *p;
        size_t blen;
        if (!next_raw(p, blen)) return false;
        const int32_t block = (int32_t)blen;
        auto u32 = [&](size_t o) { uint32_t v; memcpy(&v, p + o, 4); return v; };
        auto u16 = [&](size_t o) { uint16_t v; memcpy(&v, p + o, 2); return v; };
        r.ref_id = (int32_t)u32(0);
        r.pos = (int32_t)u32(4);
        const uint8_t l_read_name = p[8];
        r.mapq = p[9];
        const uint16_t n_cigar = u16(12);
        r.flag = u16(14);
        const uint32_t l_seq = u32(16);
        size_t o = 32;
        // untrusted input: the variable-length parts must lie inside the record
        if (32 + (size_t)l_read_name + (size_t)n_cigar * 4 + ((size_t)l_seq + 1) / 2 + (size_t)l_seq > (size_t)block)
            throw std::runtime_error("corrupt BAM record");
        r.name.assign((const char *)p + o, l_read_name ? l_read_name - 1 : 0);
        o += l_read_name;
        r.cigar.resize(n_cigar);
        memcpy(r.cigar.data(), p + o, (size_t)n_cigar * 4);
        o += (size_t)n_cigar * 4;
        static const uint8_t nt16[16] = {4, 0, 1, 4, 2, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4};  // =ACMGRSVTWYHKDBN
        r.seq.resize(unpack_seq ? l_seq : 0);
        if (unpack_seq)
            for (uint32_t i = 0; i < l_seq; ++i) {
                const uint8_t b = p[o + i / 2];
                r.seq[i] = nt16[(i & 1) ? (b & 15) : (b >> 4)];
            }
        r.seq4.assign(p + o, p + o + (l_seq + 1) / 2);
        o += (l_seq + 1) / 2;
        r.qual.assign(p + o, p + o + l_seq);
        o += l_seq;
        parse_aux(p, o, (size_t)block, r);
        return true;
    }

private:
    template <typename T> T rd() { T v; if (!in_.read(&v, sizeof v)) throw std::runtime_error("truncated BAM header"); return v; }
    BgzfReader in_;
    std::string text_;
    std::vector<BamRef> refs_;
    std::vector<uint8_t> buf_;
};

class BamWriter {
public:
    BamWriter(const std::string &path, const std::string &header_text, const std::vector<BamRef> &refs) : out_(path)
    {
        out_.write("BAM\1", 4);
        wr<int32_t>((int32_t)header_text.size());
        out_.write(header_text.data(), header_text.size());
        wr<int32_t>((int32_t)refs.size());
        for (const BamRef &r : refs) {
            wr<int32_t>((int32_t)r.name.size() + 1);
            out_.write(r.name.c_str(), r.name.size() + 1);
            wr<int32_t>((int32_t)r.length);
        }
    }
    // seq codes 0..3 = ACGT, 4 = N
    void write(const BamRecord &r)
    {
        std::vector<uint8_t> b;
        auto put = [&](const void *p, size_t n) { b.insert(b.end(), (const uint8_t *)p, (const uint8_t *)p + n); };
        auto p32 = [&](uint32_t v) { put(&v, 4); };
        auto p16 = [&](uint16_t v) { put(&v, 2); };
        uint32_t ref_len = 0;
        for (uint32_t c : r.cigar) {
            const uint32_t op = c & 15;
            if (op == CIG_M || op == CIG_D || op == CIG_N || op == CIG_EQ || op == CIG_X) ref_len += c >> 4;
        }
        p32((uint32_t)r.ref_id);
        p32((uint32_t)r.pos);
        b.push_back((uint8_t)(r.name.size() + 1));
        b.push_back(r.mapq);
        p16(4680);  // bin: not used by this reader; constant is fine for unindexed files
        p16((uint16_t)r.cigar.size());
        p16(r.flag);
        p32((uint32_t)r.seq.size());
        p32(0xFFFFFFFFu);  // next refID
        p32(0xFFFFFFFFu);  // next pos
        p32(0);            // tlen
        put(r.name.c_str(), r.name.size() + 1);
        put(r.cigar.data(), r.cigar.size() * 4);
        static const uint8_t code16[5] = {1, 2, 4, 8, 15};
        for (size_t i = 0; i < r.seq.size(); i += 2) {
            const uint8_t hi = code16[r.seq[i] > 4 ? 4 : r.seq[i]];
            const uint8_t lo = i + 1 < r.seq.size() ? code16[r.seq[i + 1] > 4 ? 4 : r.seq[i + 1]] : 0;
            b.push_back((uint8_t)(hi << 4 | lo));
        }
        for (size_t i = 0; i < r.seq.size(); ++i) b.push_back(i < r.qual.size() ? r.qual[i] : 0xFF);
        if (r.rq >= 0.f) {
            put("rqf", 3);
            put(&r.rq, 4);
        }
        auto put_z = [&](const char *tag, const std::string &v) {
            if (v.empty()) return;
            put(tag, 2);
            b.push_back('Z');
            put(v.c_str(), v.size() + 1);
        };
        put_z("dq", r.dq);
        put_z("iq", r.iq);
        put_z("sq", r.sq);
        (void)ref_len;
        wr<int32_t>((int32_t)b.size());
        out_.write(b.data(), b.size());
    }
    void close() { out_.close(); }

private:
    template <typename T> void wr(T v) { out_.write(&v, sizeof v); }
    BgzfWriter out_;
};

}  // namespace jlhost
