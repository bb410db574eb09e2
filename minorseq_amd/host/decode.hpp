// decode.hpp — the BAM decode of the front end as a pipeline over a thread pool (doc/JULIET.md:50-58):
//
//   reader     reads the file's BGZF blocks in segments of kSegBlocks blocks and hands each to the pool to inflate
//   splitter   takes the inflated segments IN ORDER, parses the BAM header once, then walks the records' block_size
//              words to find the record boundaries (a record may straddle two segments: its bytes are joined in a
//              side buffer) and hands each segment's run of whole records to the pool to parse
//   parsers    (pool) one segment's records -> one RecordArrays chunk: positions, cigar words, BAM's packed bases,
//              optionally the effective qualities, the names; plus the extent the records cover
//   consumer   (the calling thread) takes the chunks IN ORDER and gives them to the sink (the uploader thread hands them
//              to the device while the next ones are parsed)
//
// zlib inflates about 370 MB/s per core and a CCS BAM inflates 8x (470 MB for 100k reads x 3 kb).  The reader this
// replaces created its inflating threads anew for every batch, cleared every inflated buffer on the one thread that read
// the file and parsed records on one thread: 0.13-0.15 s.  Here the pool takes every core the process may use (at most
// 64) for both kinds of work: 0.08 s on a 16-core share, where the inflate alone is 470 MB / (16 x 370 MB/s).  Files that are not BGZF (plain concatenated gzip) take the sequential
// reader (collect_records).
#pragma once
#include <sched.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

#include "crc32_fold.hpp"
#include "fast_inflate.hpp"
#include "msa_builder.hpp"

namespace jlhost {

class WorkPool {
public:
    explicit WorkPool(unsigned n)
    {
        for (unsigned i = 0; i < std::max(1u, n); ++i) th_.emplace_back([this] { run(); });
    }
    ~WorkPool()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (std::thread &t : th_) t.join();
    }
    void submit(std::function<void()> f)
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            q_.push_back(std::move(f));
        }
        cv_.notify_one();
    }
    unsigned size() const { return (unsigned)th_.size(); }

private:
    void run()
    {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [this] { return stop_ || !q_.empty(); });
                if (q_.empty()) return;
                f = std::move(q_.front());
                q_.pop_front();
            }
            f();
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<std::function<void()>> q_;
    bool stop_ = false;
    std::vector<std::thread> th_;
};

// bytes without the zero-fill of std::vector (an inflated segment is overwritten whole: clearing 470 MB first, on the one
// thread that reads the file, was a third of the decode)
struct RawBuf {
    std::unique_ptr<uint8_t[]> p;
    size_t n = 0, cap = 0;
    void alloc(size_t bytes)   // keeps what it has when that is enough: buffers are recycled (see PipelinedBamReader)
    {
        if (bytes > cap || !p) {
            cap = std::max<size_t>(bytes, 1);
            p.reset(new uint8_t[cap]);
        }
        n = bytes;
    }
    void assign(const uint8_t *b, const uint8_t *e) { alloc((size_t)(e - b)); if (n) memcpy(p.get(), b, n); }
    void clear() { p.reset(); n = 0; cap = 0; }
    const uint8_t *data() const { return p.get(); }
    uint8_t *data() { return p.get(); }
    size_t size() const { return n; }
};

class PipelinedBamReader {
public:
    static constexpr size_t kSegBlocks = 128;   // BGZF blocks per segment: up to 8 MiB inflated, about 1700 CCS records of 3 kb

    // Cores this process may really use: the smallest of what the hardware has, what its affinity mask allows and what its
    // cgroup's CPU quota gives (a container on a 256-core host often owns 16 of them: 64 inflating threads then run SLOWER
    // than 16 — measured 98 vs 79 ms on this project's GPU boxes).
    static unsigned usable_cpus()
    {
        unsigned n = std::thread::hardware_concurrency();
        if (!n) n = 1;
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min<unsigned>(n, (unsigned)std::max(1, CPU_COUNT(&set)));
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {          // cgroup v2: "<quota> <period>" or "max <period>"
            long long quota = 0, period = 0;
            if (fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0)
                n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
            fclose(f);
        } else {                                                      // cgroup v1
            long long quota = -1, period = 0;
            if (FILE *q = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(q, "%lld", &quota) != 1) quota = -1; fclose(q); }
            if (FILE *q = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(q, "%lld", &period) != 1) period = 0; fclose(q); }
            if (quota > 0 && period > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
        }
        return n;
    }

    // is the file BGZF (block mode)?  Otherwise the caller takes the sequential reader.
    static bool is_bgzf(const std::string &path)
    {
        FILE *f = fopen(path.c_str(), "rb");
        if (!f) throw std::runtime_error("cannot open " + path);
        uint8_t h[18];
        const size_t got = fread(h, 1, sizeof h, f);
        fclose(f);
        return got == sizeof h && bgzf_header(h);
    }

    // Every kept record of reference `ref_id` (-1: the reference of the first kept record) to `sink`, chunk by chunk in
    // file order; returns the extent they cover.
    static ReadExtent run(const std::string &bam, const IngestOptions &opt, int ref_id, bool want_qual, const RecordSink &sink,
                          std::vector<BamRef> *refs, std::string *header_text, unsigned threads = 0)
    {
        unsigned nt = threads ? threads : std::min(64u, usable_cpus());
        if (const char *e = getenv("JL_DECODE_THREADS")) nt = (unsigned)std::max(1, atoi(e));     // (tuning)
        if (const char *env = getenv("JL_BGZF_THREADS")) { const int v = atoi(env); if (v > 0 && v <= 256) nt = (unsigned)v; }
        PipelinedBamReader r(bam, opt, ref_id, want_qual, nt);
        return r.drive(sink, refs, header_text);
    }

private:
    struct Segment {   // one run of BGZF blocks through the pipeline
        std::vector<uint8_t> comp;
        RawBuf out;
        size_t out_total = 0;
        struct Blk { size_t in_off, in_len, out_off; uint32_t isize; };
        std::vector<Blk> blks;
        // set by the splitter: the whole records of this segment
        std::vector<uint8_t> joined;          // a record that began in earlier segments, complete with its head from this one
        size_t rec_begin = 0, rec_end = 0;    // [rec_begin, rec_end) of `out`: whole records
        int ref_id = -1;
        // set by the parser
        RecordArrays chunk;
        ReadExtent extent;
        // hand-offs
        std::mutex m;
        std::condition_variable cv;
        bool inflated = false, parsed = false, last = false;
        std::string error;
    };
    using SegPtr = std::shared_ptr<Segment>;

    static bool bgzf_header(const uint8_t *h)
    {
        return h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4) && h[10] == 6 && h[11] == 0 && h[12] == 'B' && h[13] == 'C';
    }

    PipelinedBamReader(const std::string &path, const IngestOptions &opt, int ref_id, bool want_qual, unsigned threads)
        : path_(path), opt_(opt), ref_id_(ref_id), want_qual_(want_qual), pool_(threads), max_inflight_(std::min<size_t>(2 * (size_t)threads + 4, 48))
    {
    }

    // ---- an ordered, bounded queue of segments between two stages
    struct Channel {
        std::mutex m;
        std::condition_variable cv;
        std::deque<SegPtr> q;
        bool closed = false;
        void push(SegPtr s, size_t bound, const std::atomic<bool> &stop)
        {
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [&] { return q.size() < bound || stop.load(); });
            q.push_back(std::move(s));
            cv.notify_all();
        }
        SegPtr pop()
        {
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [&] { return !q.empty() || closed; });
            if (q.empty()) return nullptr;
            SegPtr s = std::move(q.front());
            q.pop_front();
            cv.notify_all();
            return s;
        }
        void close()
        {
            {
                std::lock_guard<std::mutex> lk(m);
                closed = true;
            }
            cv.notify_all();
        }
        void wake() { cv.notify_all(); }
    };

    static void wait_flag(Segment &s, bool Segment::*flag)
    {
        std::unique_lock<std::mutex> lk(s.m);
        s.cv.wait(lk, [&] { return s.*flag; });
    }
    static void set_flag(Segment &s, bool Segment::*flag)
    {
        {
            std::lock_guard<std::mutex> lk(s.m);
            s.*flag = true;
        }
        s.cv.notify_all();
    }

    // Inflated segments and record chunks are recycled (at most max_inflight_ buffers exist; their pages stay warm instead
    // of sixty mmap / munmap pairs and 115 000 first-touch faults per file).  Measured: no change on the pool's boxes, whose
    // 16-core share the inflate itself fills — kept for hosts with more cores than that.
    RawBuf take_out()
    {
        std::lock_guard<std::mutex> lk(pool_m_);
        if (free_out_.empty()) return RawBuf();
        RawBuf b = std::move(free_out_.back());
        free_out_.pop_back();
        return b;
    }
    void give_out(RawBuf &&b)
    {
        if (!b.p) return;
        std::lock_guard<std::mutex> lk(pool_m_);
        free_out_.push_back(std::move(b));
    }
    void take_chunk(RecordArrays &c)
    {
        std::lock_guard<std::mutex> lk(pool_m_);
        if (free_chunks_.empty()) return;
        c = std::move(free_chunks_.back());
        free_chunks_.pop_back();
    }
    void give_chunk(RecordArrays &&c)
    {
        c.clear();
        std::lock_guard<std::mutex> lk(pool_m_);
        if (free_chunks_.size() < 64) free_chunks_.push_back(std::move(c));
    }

    static uint64_t thread_cpu_ns()
    {
        timespec t;
        clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t);
        return (uint64_t)t.tv_sec * 1000000000ull + (uint64_t)t.tv_nsec;
    }

    void inflate_segment(Segment &s)
    {
        const uint64_t c0 = thread_cpu_ns();
        try {
            s.out = take_out();
            s.out.alloc(s.out_total);   // (on a pool thread: a first touch of its pages is part of the parallel work)
            static thread_local jlz::Inflater inf;   // (26 KB of tables: one per pool thread)
            for (const Segment::Blk &k : s.blks) {
                if (k.isize == 0) continue;
                if (inf.run(s.comp.data() + k.in_off, k.in_len, s.out.data() + k.out_off, k.isize) != 0)
                    throw std::runtime_error("BGZF block failed to inflate");
                // the block's CRC-32 (RFC 1952) sits behind the deflate data: a damaged block that still inflates to ISIZE
                // bytes must not be parsed as records (carry-less-multiply CRC: 0.05 ms per MB)
                uint32_t want;
                memcpy(&want, s.comp.data() + k.in_off + k.in_len, 4);
                if (crc32_of(s.out.data() + k.out_off, k.isize) != want) throw std::runtime_error("BGZF block fails its CRC-32");
            }
            std::vector<uint8_t>().swap(s.comp);
        } catch (const std::exception &ex) {
            s.error = ex.what();
        }
        cpu_inflate_ns_ += thread_cpu_ns() - c0;
        set_flag(s, &Segment::inflated);
    }

    void parse_segment(Segment &s)
    {
        const uint64_t c0 = thread_cpu_ns();
        try {
            if (s.error.empty()) {
                int rid = s.ref_id;
                take_chunk(s.chunk);
                {   // a CCS record is about a third packed bases, a few cigar words per 100 bases; growing by doubling
                    // would copy the arrays three or four times
                    const size_t bytes = s.rec_end - s.rec_begin + s.joined.size();
                    s.chunk.seq4.reserve(bytes / 3 + 64);
                    s.chunk.cigar.reserve(bytes / 32 + 64);
                    if (want_qual_) s.chunk.qual.reserve(bytes * 2 / 3 + 64);
                }
                auto one = [&](const uint8_t *p, size_t n) {   // p at the block_size word of a whole record
                    uint32_t block;
                    memcpy(&block, p, 4);
                    (void)n;
                    if (rid >= 0) parse_record(p + 4, block, opt_, want_qual_, rid, s.chunk, s.extent);
                };
                if (!s.joined.empty()) one(s.joined.data(), s.joined.size());
                size_t o = s.rec_begin;
                while (o < s.rec_end) {
                    uint32_t block;
                    memcpy(&block, s.out.data() + o, 4);
                    one(s.out.data() + o, 4 + (size_t)block);
                    o += 4 + (size_t)block;
                }
            }
            give_out(std::move(s.out));
            s.out = RawBuf();
            std::vector<uint8_t>().swap(s.joined);
        } catch (const std::exception &ex) {
            s.error = ex.what();
        }
        cpu_parse_ns_ += thread_cpu_ns() - c0;
        set_flag(s, &Segment::parsed);
    }

    // reader thread: raw blocks -> segments -> inflate tasks
    void read_loop()
    {
        FILE *f = fopen(path_.c_str(), "rb");
        std::string err;
        try {
            if (!f) throw std::runtime_error("cannot open " + path_);
            bool eof = false;
            while (!eof && !stop_) {
                SegPtr s = std::make_shared<Segment>();
                size_t out_total = 0;
                while (s->blks.size() < kSegBlocks) {
                    uint8_t hdr[18];
                    const size_t got = fread(hdr, 1, sizeof hdr, f);
                    if (got == 0) { eof = true; break; }
                    if (got != sizeof hdr || !bgzf_header(hdr)) throw std::runtime_error("corrupt BGZF block header");
                    const size_t bsize = (size_t)hdr[16] + ((size_t)hdr[17] << 8) + 1;  // whole block
                    if (bsize < 26) throw std::runtime_error("corrupt BGZF block size");
                    const size_t body = bsize - 18;  // deflate data + crc32 + isize
                    const size_t off = s->comp.size();
                    s->comp.resize(off + body);
                    if (fread(s->comp.data() + off, 1, body, f) != body) throw std::runtime_error("truncated BGZF block");
                    uint32_t isize;
                    memcpy(&isize, s->comp.data() + off + body - 4, 4);
                    if (isize > (1u << 16)) throw std::runtime_error("BGZF block larger than 64 KiB");
                    s->blks.push_back({off, body - 8, out_total, isize});
                    out_total += isize;
                }
                if (s->blks.empty()) break;
                s->out_total = out_total;
                Segment *raw = s.get();
                SegPtr keep = s;
                pool_.submit([this, raw, keep] { inflate_segment(*raw); });
                to_split_.push(std::move(s), max_inflight_, stop_);
            }
        } catch (const std::exception &ex) {
            err = ex.what();
        }
        if (f) fclose(f);
        if (!err.empty()) {   // an empty, failed segment carries the error down the pipeline in order
            SegPtr s = std::make_shared<Segment>();
            s->error = err;
            s->inflated = true;
            to_split_.push(std::move(s), max_inflight_, stop_);
        }
        to_split_.close();
    }

    // splitter thread: header once, then record boundaries; segments leave in order with their run of whole records
    void split_loop()
    {
        std::vector<uint8_t> carry;   // bytes of the stream not yet given away (header in progress, or a record's head)
        bool header_done = false;
        std::string err;
        while (SegPtr s = to_split_.pop()) {
            wait_flag(*s, &Segment::inflated);
            if (!s->error.empty() || !err.empty()) {
                if (s->error.empty()) s->error = err;
                err = s->error;
                s->out.clear();
                s->rec_begin = s->rec_end = 0;
            } else {
                try {
                    size_t o = 0;   // first byte of s->out not yet accounted for
                    if (!header_done) {
                        carry.insert(carry.end(), s->out.data(), s->out.data() + s->out.size());
                        o = s->out.size();
                        size_t used = 0;
                        if (parse_header(carry, used)) {
                            header_done = true;
                            // what follows the header in `carry` is record data: make it this segment's `out`
                            s->out.assign(carry.data() + used, carry.data() + carry.size());
                            carry.clear();
                            o = 0;
                        } else {
                            s->out.clear();
                        }
                    }
                    if (header_done) {
                        const RawBuf &b = s->out;
                        if (!carry.empty()) {   // a record that began earlier: complete it from the head of this segment
                            while (carry.size() < 4 && o < b.size()) carry.push_back(b.data()[o++]);
                            if (carry.size() >= 4) {
                                uint32_t block;
                                memcpy(&block, carry.data(), 4);
                                if (block < 32) throw std::runtime_error("corrupt BAM record");
                                const size_t want = 4 + (size_t)block, have = carry.size();
                                const size_t take = std::min(want - have, b.size() - o);
                                carry.insert(carry.end(), b.data() + o, b.data() + o + take);
                                o += take;
                                if (carry.size() == want) {
                                    s->joined.swap(carry);
                                    carry.clear();
                                    note_first_kept(s->joined.data());
                                }
                            }
                        }
                        s->rec_begin = o;
                        while (carry.empty() && o < b.size()) {
                            if (b.size() - o < 4) break;
                            uint32_t block;
                            memcpy(&block, b.data() + o, 4);
                            if (block < 32) throw std::runtime_error("corrupt BAM record");
                            if (b.size() - o < 4 + (size_t)block) break;
                            note_first_kept(b.data() + o);
                            o += 4 + (size_t)block;
                        }
                        s->rec_end = o;
                        if (carry.empty() && o < b.size()) carry.assign(b.data() + o, b.data() + b.size());
                    }
                    s->ref_id = ref_id_;
                } catch (const std::exception &ex) {
                    s->error = err = ex.what();
                }
            }
            Segment *raw = s.get();
            SegPtr keep = s;
            pool_.submit([this, raw, keep] { parse_segment(*raw); });
            to_consume_.push(std::move(s), max_inflight_, stop_);
        }
        if (err.empty() && (!header_done || !carry.empty())) {   // the stream ended inside the header or a record
            SegPtr s = std::make_shared<Segment>();
            s->error = header_done ? "truncated BAM record" : "not a BAM file (truncated header)";
            s->inflated = s->parsed = true;
            to_consume_.push(std::move(s), max_inflight_, stop_);
        }
        to_consume_.close();
    }

    // the reference of the first KEPT record decides which reference is read (ref_id < 0 on entry)
    void note_first_kept(const uint8_t *rec)
    {
        if (ref_id_ >= 0) return;
        uint32_t block;
        memcpy(&block, rec, 4);
        const uint8_t *p = rec + 4;
        auto u32 = [&](size_t o) { uint32_t v; memcpy(&v, p + o, 4); return v; };
        auto u16 = [&](size_t o) { uint16_t v; memcpy(&v, p + o, 2); return v; };
        const int32_t rid = (int32_t)u32(0), pos = (int32_t)u32(4);
        const uint32_t flag = u16(14);
        if ((flag & 0x4) || (flag & 0x100) || rid < 0 || pos < 0) return;
        if (opt_.min_rq > 0.0) {
            const uint32_t l_name = p[8], n_cigar = u16(12), l_seq = u32(16);
            const size_t o_aux = 32 + (size_t)l_name + (size_t)n_cigar * 4 + ((size_t)l_seq + 1) / 2 + l_seq;
            if (o_aux > block) throw std::runtime_error("corrupt BAM record");
            BamRecord r;
            BamReader::parse_aux(p, o_aux, block, r);
            if (r.rq >= 0.f && r.rq < opt_.min_rq) return;
        }
        ref_id_ = rid;
    }

    // magic, text, references: true once `buf` holds all of it (used = bytes of the header)
    bool parse_header(const std::vector<uint8_t> &buf, size_t &used)
    {
        size_t o = 0;
        auto need = [&](size_t n) { return buf.size() - o >= n; };
        auto i32 = [&]() { int32_t v; memcpy(&v, buf.data() + o, 4); o += 4; return v; };
        if (!need(8)) return false;
        if (memcmp(buf.data(), "BAM\1", 4) != 0) throw std::runtime_error(path_ + ": not a BAM file");
        o = 4;
        const int32_t l_text = i32();
        if (l_text < 0) throw std::runtime_error("corrupt BAM header");
        if (!need((size_t)l_text + 4)) return false;
        text_.assign((const char *)buf.data() + o, (size_t)l_text);
        o += (size_t)l_text;
        const int32_t n_ref = i32();
        if (n_ref < 0) throw std::runtime_error("corrupt BAM header");
        refs_.clear();
        for (int32_t i = 0; i < n_ref; ++i) {
            if (!need(4)) return false;
            const int32_t l_name = i32();
            if (l_name < 0) throw std::runtime_error("corrupt BAM header");
            if (!need((size_t)l_name + 4)) return false;
            std::string nm((const char *)buf.data() + o, (size_t)l_name);
            o += (size_t)l_name;
            if (!nm.empty() && nm.back() == '\0') nm.pop_back();
            refs_.push_back(BamRef{nm, (uint32_t)i32()});
        }
        used = o;
        return true;
    }

    ReadExtent drive(const RecordSink &sink, std::vector<BamRef> *refs, std::string *header_text)
    {
        std::thread reader([this] { read_loop(); });
        std::thread splitter([this] { split_loop(); });
        ReadExtent e;
        std::string err;
        while (SegPtr s = to_consume_.pop()) {
            wait_flag(*s, &Segment::parsed);
            if (!err.empty()) continue;   // drain: the stages upstream finish what they hold
            if (!s->error.empty()) {
                err = s->error;
                stop_ = true;
                to_split_.wake();
                to_consume_.wake();
                continue;
            }
            e.n_reads += s->extent.n_reads;
            e.min_pos = std::min(e.min_pos, s->extent.min_pos);
            e.max_end = std::max(e.max_end, s->extent.max_end);
            if (!s->chunk.pos.empty() && sink.give) sink.give(s->chunk);   // the sink trades it for an empty one ...
            give_chunk(std::move(s->chunk));                                  // ... which the next parser fills
        }
        reader.join();
        splitter.join();
        if (!err.empty()) throw std::runtime_error(err);
        if (getenv("JL_DECODE_STATS"))
            fprintf(stderr, "juliet: decode: %u pool threads, cpu inflate %.1f ms, cpu parse %.1f ms\n", pool_.size(),
                    cpu_inflate_ns_.load() * 1e-6, cpu_parse_ns_.load() * 1e-6);
        e.ref_id = ref_id_;
        if (refs) *refs = refs_;
        if (header_text) *header_text = text_;
        return e;
    }

    std::string path_;
    IngestOptions opt_;
    int ref_id_;
    bool want_qual_;
    WorkPool pool_;
    size_t max_inflight_;
    std::atomic<uint64_t> cpu_inflate_ns_{0}, cpu_parse_ns_{0};   // thread CPU time of the two kinds of pool work
    std::atomic<bool> stop_{false};   // set by the consumer on the first error (pushes stop waiting for room; everything still drains)
    Channel to_split_, to_consume_;
    std::mutex pool_m_;
    std::vector<RawBuf> free_out_;
    std::vector<RecordArrays> free_chunks_;
    std::string text_;
    std::vector<BamRef> refs_;
};

}  // namespace jlhost
