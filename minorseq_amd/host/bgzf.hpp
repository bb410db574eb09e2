// bgzf.hpp — minimal BGZF (blocked gzip) reader and writer over zlib, enough for PacBio BAM in/out.
// Input contract: doc/JULIET.md:50-58 (aligned CCS reads in BAM).  No htslib in this image.
#pragma once
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace jlhost {

// Reader.  BGZF is a series of independent gzip members of at most 64 KiB, each announcing its compressed
// size in a 'BC' extra field, so batches of blocks are inflated in parallel; files whose first
// member lacks the field are read as a plain concatenated-gzip stream.
class BgzfReader {
public:
    explicit BgzfReader(const std::string &path, unsigned threads = 0) : f_(fopen(path.c_str(), "rb"))
    {
        if (!f_) throw std::runtime_error("cannot open " + path);
        unsigned hw = std::thread::hardware_concurrency();
        n_threads_ = threads ? threads : std::min(16u, hw ? hw : 1u);
        if (const char *e = getenv("JL_BGZF_THREADS")) { const int v = atoi(e); if (v > 0 && v <= 64) n_threads_ = (unsigned)v; }
        uint8_t hdr[18];
        const size_t got = fread(hdr, 1, sizeof hdr, f_);
        block_mode_ = got == sizeof hdr && is_bgzf_header(hdr);
        fseek(f_, 0, SEEK_SET);
        if (!block_mode_) {
            memset(&z_, 0, sizeof z_);
            if (inflateInit2(&z_, 15 + 32) != Z_OK) throw std::runtime_error("inflateInit2 failed");
            in_.resize(1 << 16);
        }
    }
    ~BgzfReader()
    {
        if (ahead_.valid()) { try { ahead_.get(); } catch (...) {} }   // the background task uses f_
        if (!block_mode_) inflateEnd(&z_);
        if (f_) fclose(f_);
    }
    // read exactly n bytes; returns false on clean EOF at a record boundary (n bytes not started)
    bool read(void *dst, size_t n) { return block_mode_ ? read_blocks(dst, n) : read_stream(dst, n); }
    // the next n bytes in place when they lie inside the inflated batch at hand (valid until the next call), else
    // nullptr: the caller then copies them with read()
    const uint8_t *peek(size_t n)
    {
        if (!block_mode_) return nullptr;
        while (out_pos_ == out_.size())
            if (!refill()) return nullptr;
        if (out_.size() - out_pos_ < n) return nullptr;
        const uint8_t *p = out_.data() + out_pos_;
        out_pos_ += n;
        return p;
    }

private:
    static bool is_bgzf_header(const uint8_t *h)
    {
        return h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4) && h[10] == 6 && h[11] == 0 && h[12] == 'B' && h[13] == 'C';
    }

    // ---- block mode: read a batch of raw blocks, inflate them on n_threads_ threads, serve in order; the NEXT batch is
    // read and inflated by a background task while the caller consumes the current one
    struct Batch {
        std::vector<uint8_t> comp, out;
        bool any = false;
    };
    void load_batch(Batch &b)
    {
        struct Blk { size_t in_off, in_len, out_off; uint32_t isize; };
        std::vector<Blk> blks;
        b.comp.clear();
        b.any = false;
        size_t out_total = 0;
        const size_t kBatch = 256;
        while (blks.size() < kBatch) {
            uint8_t hdr[18];
            const size_t got = fread(hdr, 1, sizeof hdr, f_);
            if (got == 0) break;
            if (got != sizeof hdr || !is_bgzf_header(hdr)) throw std::runtime_error("corrupt BGZF block header");
            const size_t bsize = (size_t)hdr[16] + ((size_t)hdr[17] << 8) + 1;  // whole block
            if (bsize < 26) throw std::runtime_error("corrupt BGZF block size");
            const size_t body = bsize - 18;  // deflate data + crc32 + isize
            const size_t off = b.comp.size();
            b.comp.resize(off + body);
            if (fread(b.comp.data() + off, 1, body, f_) != body) throw std::runtime_error("truncated BGZF block");
            uint32_t isize;
            memcpy(&isize, b.comp.data() + off + body - 4, 4);
            if (isize > (1u << 16)) throw std::runtime_error("BGZF block larger than 64 KiB");
            blks.push_back({off, body - 8, out_total, isize});
            out_total += isize;
        }
        if (blks.empty()) return;
        b.any = true;
        b.out.resize(out_total);
        std::atomic<size_t> next{0};
        std::atomic<bool> bad{false}, bad_crc{false};
        auto work = [&]() {
            z_stream z;
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= blks.size()) return;
                const Blk &k = blks[i];
                if (k.isize == 0) continue;
                memset(&z, 0, sizeof z);
                if (inflateInit2(&z, -15) != Z_OK) { bad = true; return; }
                z.next_in = b.comp.data() + k.in_off;
                z.avail_in = (uInt)k.in_len;
                z.next_out = b.out.data() + k.out_off;
                z.avail_out = k.isize;
                const int rc = inflate(&z, Z_FINISH);
                inflateEnd(&z);
                if (rc != Z_STREAM_END || z.avail_out != 0 || z.avail_in != 0) { bad = true; return; }
                uint32_t want;   // the block's CRC-32 behind the deflate data
                memcpy(&want, b.comp.data() + k.in_off + k.in_len, 4);
                if ((uint32_t)crc32(0L, b.out.data() + k.out_off, k.isize) != want) { bad_crc = true; bad = true; return; }
            }
        };
        const unsigned nt = (unsigned)std::min<size_t>(n_threads_, blks.size());
        if (nt <= 1) work();
        else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; ++t) th.emplace_back(work);
            for (auto &t : th) t.join();
        }
        if (bad_crc) throw std::runtime_error("BGZF block fails its CRC-32");
        if (bad) throw std::runtime_error("BGZF block failed to inflate");
    }
    bool refill()
    {
        if (!ahead_.valid()) {   // first call: nothing in flight yet
            load_batch(batch_[cur_]);
        } else {
            ahead_.get();        // rethrows what the background task threw
            cur_ ^= 1;
        }
        if (!batch_[cur_].any) return false;
        out_.swap(batch_[cur_].out);
        out_pos_ = 0;
        Batch *nxt = &batch_[cur_ ^ 1];
        ahead_ = std::async(std::launch::async, [this, nxt]() { load_batch(*nxt); });
        return true;
    }
    bool read_blocks(void *dst, size_t n)
    {
        uint8_t *out = static_cast<uint8_t *>(dst);
        size_t got = 0;
        while (got < n) {
            if (out_pos_ == out_.size()) {
                if (!refill()) {
                    if (got == 0) return false;
                    throw std::runtime_error("truncated BGZF stream");
                }
                continue;
            }
            const size_t take = std::min(n - got, out_.size() - out_pos_);
            memcpy(out + got, out_.data() + out_pos_, take);
            out_pos_ += take;
            got += take;
        }
        return true;
    }

    // ---- stream mode: plain inflate with member restart
    bool read_stream(void *dst, size_t n)
    {
        uint8_t *out = static_cast<uint8_t *>(dst);
        size_t got = 0;
        while (got < n) {
            if (z_.avail_in == 0 && !eof_) {
                z_.avail_in = (uInt)fread(in_.data(), 1, in_.size(), f_);
                z_.next_in = in_.data();
                if (z_.avail_in == 0) eof_ = true;
            }
            if (z_.avail_in == 0 && eof_) {
                if (got == 0) return false;
                throw std::runtime_error("truncated BGZF stream");
            }
            z_.next_out = out + got;
            z_.avail_out = (uInt)std::min<size_t>(n - got, 1u << 30);
            const size_t before = z_.avail_out;
            const int rc = inflate(&z_, Z_NO_FLUSH);
            got += before - z_.avail_out;
            if (rc == Z_STREAM_END) {
                if (inflateReset(&z_) != Z_OK) throw std::runtime_error("inflateReset failed");
            } else if (rc != Z_OK && rc != Z_BUF_ERROR) {
                throw std::runtime_error(std::string("inflate: ") + (z_.msg ? z_.msg : "error"));
            }
        }
        return true;
    }

    FILE *f_;
    bool block_mode_ = false;
    unsigned n_threads_ = 1;
    std::vector<uint8_t> out_;
    size_t out_pos_ = 0;
    Batch batch_[2];
    int cur_ = 0;
    std::future<void> ahead_;
    z_stream z_;
    std::vector<uint8_t> in_;
    bool eof_ = false;
};

class BgzfWriter {
public:
    explicit BgzfWriter(const std::string &path) : f_(fopen(path.c_str(), "wb"))
    {
        if (!f_) throw std::runtime_error("cannot create " + path);
        buf_.reserve(kBlock);
        const unsigned hc = std::thread::hardware_concurrency();
        threads_ = std::max(1u, std::min(hc ? hc : 1u, 16u));
    }
    ~BgzfWriter()
    {
        try { close(); } catch (...) {}
    }
    void write(const void *src, size_t n)
    {
        const uint8_t *p = static_cast<const uint8_t *>(src);
        while (n) {
            const size_t take = std::min(n, kBlock - buf_.size());
            buf_.insert(buf_.end(), p, p + take);
            p += take;
            n -= take;
            if (buf_.size() == kBlock) queue_block();
        }
    }
    void close()
    {
        if (!f_) return;
        if (!buf_.empty()) queue_block();
        flush_batch();
        static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0,
                                        0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        fwrite(eof, 1, sizeof eof, f_);
        fclose(f_);
        f_ = nullptr;
    }

private:
    static constexpr size_t kBlock = 0xff00;
    static constexpr size_t kBatch = 256;      // blocks deflated together, by the threads in turns (BGZF members are independent)
    // one BGZF member (header, deflate data, crc32, isize) of `in`
    static void deflate_block(const std::vector<uint8_t> &in, std::vector<uint8_t> &out)
    {
        out.resize(compressBound((uLong)in.size()) + 64 + 26);
        z_stream z;
        memset(&z, 0, sizeof z);
        if (deflateInit2(&z, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw std::runtime_error("deflateInit2");
        z.next_in = const_cast<uint8_t *>(in.data());
        z.avail_in = (uInt)in.size();
        z.next_out = out.data() + 18;
        z.avail_out = (uInt)(out.size() - 26);
        if (deflate(&z, Z_FINISH) != Z_STREAM_END) throw std::runtime_error("deflate");
        const size_t clen = z.total_out;
        deflateEnd(&z);
        const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), in.data(), (uInt)in.size());
        const uint16_t bsize = (uint16_t)(clen + 25);  // total block size - 1
        const uint8_t hdr[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, (uint8_t)(bsize & 0xff), (uint8_t)(bsize >> 8)};
        memcpy(out.data(), hdr, 18);
        const uint32_t isize = (uint32_t)in.size();
        memcpy(out.data() + 18 + clen, &crc, 4);
        memcpy(out.data() + 18 + clen + 4, &isize, 4);
        out.resize(18 + clen + 8);
    }
    void queue_block()
    {
        batch_.emplace_back();
        batch_.back().swap(buf_);
        buf_.reserve(kBlock);
        if (batch_.size() == kBatch) flush_batch();
    }
    void flush_batch()
    {
        if (batch_.empty()) return;
        std::vector<std::vector<uint8_t>> comp(batch_.size());
        const unsigned nt = (unsigned)std::min<size_t>(threads_, batch_.size());
        std::atomic<size_t> next{0};
        std::atomic<bool> failed{false};
        auto work = [&]() {
            try {
                for (size_t i = next++; i < batch_.size(); i = next++) deflate_block(batch_[i], comp[i]);
            } catch (...) { failed = true; }
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < nt; ++t) pool.emplace_back(work);
        work();
        for (std::thread &t : pool) t.join();
        if (failed) throw std::runtime_error("deflate");
        for (const std::vector<uint8_t> &c : comp)
            if (fwrite(c.data(), 1, c.size(), f_) != c.size()) throw std::runtime_error("short write");
        batch_.clear();
    }
    FILE *f_;
    std::vector<uint8_t> buf_;
    std::vector<std::vector<uint8_t>> batch_;
    unsigned threads_ = 1;
};

}  // namespace jlhost
