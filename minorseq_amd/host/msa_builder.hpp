// msa_builder.hpp — aligned BAM records -> by-row symbol matrix of one reference window
// (doc/JULIET.md:50-58 input contract; :26-27 insertions ignored, deletions are '-'; :256-259 filtered base = N).
#pragma once
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <sys/mman.h>

#include <filesystem>
#include <functional>
#include <limits>
#include <system_error>

#include "bam.hpp"

namespace jlhost {

struct IngestOptions {
    int ref_id = -1;        // -1: the reference of the first kept record
    uint32_t min_qv = 0;    // bases whose lowest QV (QUAL and the rich-QV tracks dq/iq/sq when present) is below this
                            // become N (doc/JULIET.md:256-259; threshold UNPINNED); 0 = off
    double min_rq = 0.0;    // skip reads with a lower rq tag; 0 = off (doc/JULIET.md:56 leaves this to the user)
};

// dst[i] = min(dst[i], track[i] - 33) over n bases, as unsigned bytes: a rich-QV track (phred + 33, a character per base) folded into
// the effective qualities — 0xFF, "nothing known", is the largest byte, so the minimum takes whatever the track says.  Sixteen bases
// an instruction: with a byte at a time over QUAL and three tracks this loop was 2.5-3.4 s of the 3.5-4.5 s of CPU time the decode of
// a 100k-read rich-QV BAM took (tools_tuning/decode_stats.cpp).
inline void min_with_track(uint8_t *dst, const char *track, size_t n)
{
    size_t i = 0;
#if defined(__SSE2__)
    const __m128i k33 = _mm_set1_epi8(33);
    for (; i + 16 <= n; i += 16) {
        const __m128i d = _mm_loadu_si128(reinterpret_cast<const __m128i *>(dst + i));
        const __m128i t = _mm_loadu_si128(reinterpret_cast<const __m128i *>(track + i));
        _mm_storeu_si128(reinterpret_cast<__m128i *>(dst + i), _mm_min_epu8(d, _mm_sub_epi8(t, k33)));
    }
#endif
    for (; i < n; ++i) {
        const uint8_t v = (uint8_t)(track[i] - 33);
        if (v < dst[i]) dst[i] = v;
    }
}

// lowest phred over QUAL and whichever rich-QV tracks the record carries, per base (0xFF = nothing known)
// (dst holds QUAL — always l_seq entries, 0xFF = absent — on entry)
inline void fold_tracks(uint8_t *dst, size_t l_seq, const BamRecord &tags)
{
    for (const std::string *t : {&tags.dq, &tags.iq, &tags.sq}) min_with_track(dst, t->data(), std::min(t->size(), l_seq));
}
inline void effective_quals(const BamRecord &r, std::vector<uint8_t> &out)
{
    out.assign(r.qual.begin(), r.qual.end());
    fold_tracks(out.data(), out.size(), r);
}

inline bool keep_record(const BamRecord &r)
{
    // "Reads that are not primary or supplementary alignments, get ignored" (doc/JULIET.md:58)
    return !(r.flag & 0x4) && !(r.flag & 0x100) && r.ref_id >= 0 && r.pos >= 0;
}

inline uint32_t ref_span(const BamRecord &r)
{
    uint32_t n = 0;
    for (uint32_t c : r.cigar) {
        const uint32_t op = c & 15;
        if (op == CIG_D || op == CIG_N || op == CIG_EQ || op == CIG_X || op == CIG_M) n += c >> 4;
    }
    return n;
}

struct ReadExtent {
    uint64_t n_reads = 0;
    int64_t min_pos = std::numeric_limits<int64_t>::max(), max_end = 0;
    int ref_id = -1;
};

inline ReadExtent scan_extent(const std::string &bam, const IngestOptions &opt)
{
    BamReader in(bam);
    BamRecord r;
    ReadExtent e;
    e.ref_id = opt.ref_id;
    while (in.next(r)) {
        if (!keep_record(r)) continue;
        if (opt.min_rq > 0.0 && r.rq >= 0.f && r.rq < opt.min_rq) continue;
        if (e.ref_id < 0) e.ref_id = r.ref_id;
        if (r.ref_id != e.ref_id) continue;
        ++e.n_reads;
        e.min_pos = std::min<int64_t>(e.min_pos, r.pos);
        e.max_end = std::max<int64_t>(e.max_end, (int64_t)r.pos + ref_span(r));
    }
    return e;
}

// rows: uint8[n_reads][n_cols] (codes 0..6), names in record order.  Returns the number of rows filled.
inline uint64_t build_rows(const std::string &bam, const IngestOptions &opt, int ref_id, uint32_t win_begin,
                           uint32_t n_cols, uint64_t cap_reads, std::vector<uint8_t> &rows,
                           std::vector<std::string> *names)
{
    rows.assign((size_t)cap_reads * n_cols, JL_SYM_NONE);
    BamReader in(bam);
    BamRecord r;
    uint64_t n = 0;
    const int64_t wb = win_begin, we = (int64_t)win_begin + n_cols;
    while (in.next(r)) {
        if (!keep_record(r) || r.ref_id != ref_id) continue;
        if (opt.min_rq > 0.0 && r.rq >= 0.f && r.rq < opt.min_rq) continue;
        if (n >= cap_reads) throw std::runtime_error("BAM changed between passes");
        uint8_t *row = rows.data() + (size_t)n * n_cols;
        std::vector<uint8_t> eq;
        if (opt.min_qv) effective_quals(r, eq);
        int64_t rp = r.pos;  // reference cursor
        size_t qp = 0;       // read cursor
        for (uint32_t c : r.cigar) {
            const uint32_t op = c & 15, len = c >> 4;
            switch (op) {
            case CIG_M:
                throw std::runtime_error("read " + r.name + ": cigar M is forbidden in PacBio-compliant BAM (doc/JULIET.md:53)");
            case CIG_EQ:
            case CIG_X:
                for (uint32_t k = 0; k < len; ++k, ++rp, ++qp) {
                    if (rp < wb || rp >= we || qp >= r.seq.size()) continue;
                    uint8_t s = r.seq[qp] < 4 ? r.seq[qp] : (uint8_t)JL_SYM_MASK;
                    if (opt.min_qv && qp < eq.size() && eq[qp] != 0xFF && eq[qp] < opt.min_qv) s = JL_SYM_MASK;
                    row[rp - wb] = s;
                }
                break;
            case CIG_D:
                for (uint32_t k = 0; k < len; ++k, ++rp)
                    if (rp >= wb && rp < we) row[rp - wb] = JL_SYM_GAP;
                break;
            case CIG_N: rp += len; break;            // reference skip: stays uncovered
            case CIG_I: case CIG_S: qp += len; break;  // insertions and clips carry no reference column
            default: break;                           // H, P
            }
        }
        if (names) names->push_back(r.name);
        ++n;
    }
    return n;
}

// Big host arrays on 2 MiB pages (transparent huge pages are opt-in on most hosts): fewer page faults while they fill,
// and the HIP runtime pins a pageable source range page by page before the DMA — 6.5 us per 4 KiB page measured on the
// MI355X box (0.6 GB/s), against 5-40 GB/s once the range is made of huge pages.
template <typename T>
struct HugeAlloc {
    using value_type = T;
    HugeAlloc() = default;
    template <typename U> HugeAlloc(const HugeAlloc<U> &) {}
    T *allocate(size_t n)
    {
        const size_t huge = (size_t)2 << 20, bytes = n * sizeof(T);
        if (bytes < huge) return static_cast<T *>(::operator new(bytes));
        void *p = std::aligned_alloc(huge, (bytes + huge - 1) & ~(huge - 1));
        if (!p) throw std::bad_alloc();
        madvise(p, (bytes + huge - 1) & ~(huge - 1), MADV_HUGEPAGE);
        return static_cast<T *>(p);
    }
    void deallocate(T *p, size_t n)
    {
        if (n * sizeof(T) < ((size_t)2 << 20)) ::operator delete(p);
        else std::free(p);
    }
    // resize() leaves new elements as they are (default-initialised) instead of writing zeros: what grows a vector of this kind is
    // filled by the caller — by several threads at once in the uploader's gather — and the zeros were a first pass over every page
    template <typename U> void construct(U *p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new (static_cast<void *>(p)) U; }
    template <typename U, typename... Args> void construct(U *p, Args &&...args) { ::new (static_cast<void *>(p)) U(std::forward<Args>(args)...); }
    template <typename U> bool operator==(const HugeAlloc<U> &) const { return true; }
    template <typename U> bool operator!=(const HugeAlloc<U> &) const { return false; }
};

// Raw record arrays for jl_msa_ingest_records (cigar expansion happens on the device).
struct RecordArrays {
    std::vector<int32_t> pos;
    std::vector<uint32_t, HugeAlloc<uint32_t>> cigar;
    std::vector<uint64_t> cig_off{0}, seq_off{0}, qual_off{0};
    std::vector<uint8_t, HugeAlloc<uint8_t>> seq4, qual;
    std::vector<std::string> names;
    void clear()   // keeps the capacity: chunks are recycled
    {
        pos.clear(); cigar.clear(); seq4.clear(); qual.clear(); names.clear();
        cig_off.assign(1, 0); seq_off.assign(1, 0); qual_off.assign(1, 0);
    }
};

// Where collect_records hands over its records `chunk_reads` at a time instead of keeping them all: give() receives a
// full chunk (offsets relative to the chunk) and returns an empty one to fill next (recycled, so its pages are warm).
struct RecordSink {
    size_t chunk_reads = 8192;
    std::function<void(RecordArrays &)> give;   // swaps the full chunk for an empty one
};

// One record of the stream -> the arrays the device ingests (shared by the sequential and the pipelined reader).
// `ref_id`: -1 = not known yet (the first kept record decides; sequential reader only).  Returns whether it was kept.
inline bool parse_record(const uint8_t *p, size_t len, const IngestOptions &opt, bool want_qual, int &ref_id, RecordArrays &out,
                         ReadExtent &e)
{
    auto u32 = [&](size_t o) { uint32_t v; memcpy(&v, p + o, 4); return v; };
    auto u16 = [&](size_t o) { uint16_t v; memcpy(&v, p + o, 2); return v; };
    if (len < 32) throw std::runtime_error("corrupt BAM record");
    const int32_t rid = (int32_t)u32(0), pos = (int32_t)u32(4);
    const uint32_t l_name = p[8], n_cigar = u16(12), flag = u16(14), l_seq = u32(16);
    const size_t o_cig = 32 + (size_t)l_name, o_seq = o_cig + (size_t)n_cigar * 4, o_qual = o_seq + ((size_t)l_seq + 1) / 2,
                 o_aux = o_qual + l_seq;
    if (o_aux > len) throw std::runtime_error("corrupt BAM record");
    // "Reads that are not primary or supplementary alignments, get ignored" (doc/JULIET.md:58)
    if ((flag & 0x4) || (flag & 0x100) || rid < 0 || pos < 0) return false;
    const bool need_tags = want_qual || opt.min_rq > 0.0;
    BamReader::AuxViews aux;      // (the tracks stay where they are in the record: folded into the qualities below)
    if (need_tags) {
        BamReader::scan_aux(p, o_aux, len, aux);
        if (opt.min_rq > 0.0 && aux.rq >= 0.f && aux.rq < opt.min_rq) return false;
    }
    if (ref_id < 0) ref_id = rid;
    if (rid != ref_id) return false;
    const size_t c_at = out.cigar.size();
    out.cigar.resize(c_at + n_cigar);
    memcpy(out.cigar.data() + c_at, p + o_cig, (size_t)n_cigar * 4);
    uint32_t span = 0;
    uint64_t query = 0;
    for (size_t k = c_at; k < out.cigar.size(); ++k) {
        const uint32_t op = out.cigar[k] & 15;
        if (op == CIG_M) {
            const std::string name((const char *)p + 32, l_name ? l_name - 1 : 0);
            throw std::runtime_error("read " + name + ": cigar M is forbidden in PacBio-compliant BAM (doc/JULIET.md:53)");
        }
        if (op == CIG_D || op == CIG_N || op == CIG_EQ || op == CIG_X) span += out.cigar[k] >> 4;
        if (op == CIG_I || op == CIG_S || op == CIG_EQ || op == CIG_X) query += out.cigar[k] >> 4;
    }
    // the device walks the cigar into the read's bases and qualities: a cigar that consumes more (or fewer) bases
    // than the record holds would index past them
    if (query != l_seq) {
        const std::string name((const char *)p + 32, l_name ? l_name - 1 : 0);
        throw std::runtime_error("read " + name + ": cigar consumes " + std::to_string(query) + " bases, the record holds " +
                                 std::to_string(l_seq));
    }
    ++e.n_reads;
    e.min_pos = std::min<int64_t>(e.min_pos, pos);
    e.max_end = std::max<int64_t>(e.max_end, (int64_t)pos + span);
    out.pos.push_back(pos);
    out.cig_off.push_back(out.cigar.size());
    // (resize + memcpy: a range insert into a vector with an allocator of its own goes through construct(), element by element)
    const size_t s_at = out.seq4.size();
    out.seq4.resize(s_at + (o_qual - o_seq));
    memcpy(out.seq4.data() + s_at, p + o_seq, o_qual - o_seq);
    out.seq_off.push_back(out.seq4.size());
    if (want_qual) {
        const size_t q_at = out.qual.size();
        out.qual.resize(q_at + l_seq);
        memcpy(out.qual.data() + q_at, p + o_qual, l_seq);
        for (int k = 0; k < 3; ++k) min_with_track(out.qual.data() + q_at, aux.track[k], std::min<size_t>(aux.len[k], l_seq));
        out.qual_off.push_back(out.qual.size());
    }
    out.names.emplace_back((const char *)p + 32, l_name ? l_name - 1 : 0);
    return true;
}

// One pass over the file: every kept record of reference `ref_id` (-1: the reference of the first kept record), plus
// the extent those records cover.  `refs` / `header_text` receive the BAM header when given.  Records are parsed in
// place in the inflated BGZF batch: positions, cigar words and BAM's packed bases are copied once, into the arrays
// the device ingests; qualities and tags are only looked at when a filter needs them.
inline ReadExtent collect_records(const std::string &bam, const IngestOptions &opt, int ref_id, bool want_qual, RecordArrays &out,
                                  std::vector<BamRef> *refs = nullptr, std::string *header_text = nullptr,
                                  const RecordSink *sink = nullptr)
{
    BamReader in(bam);
    if (refs) *refs = in.refs();
    if (header_text) *header_text = in.header_text();
    ReadExtent e;
    e.ref_id = ref_id;
    {   // address space for the big arrays up front (untouched pages cost nothing): growing by doubling would copy them
        // and fault every page in again.  CCS BAMs inflate 5-10x; the bases are about a quarter of that
        std::error_code ec;
        const uintmax_t fsz = std::filesystem::file_size(bam, ec);
        if (!ec && fsz > 0 && !sink) {
            const size_t cap = (size_t)std::min<uintmax_t>(fsz * 4, (uintmax_t)4 << 30);
            try {
                out.seq4.reserve(cap);
                out.cigar.reserve(cap / 8);
                if (want_qual) out.qual.reserve(cap * 2);
            } catch (const std::bad_alloc &) {}   // doubling takes over
        }
    }
    const uint8_t *p;
    size_t len;
    while (in.next_raw(p, len)) {
        parse_record(p, len, opt, want_qual, e.ref_id, out, e);
        if (sink && out.pos.size() >= sink->chunk_reads) sink->give(out);
    }
    if (sink && !out.pos.empty()) sink->give(out);
    return e;
}

}  // namespace jlhost
