// bgzf.hpp — minimal BGZF (blocked gzip) reader and writer over zlib, enough for PacBio BAM in/out.
// Input contract: doc/JULIET.md:50-58 (aligned CCS reads in BAM).  No htslib in this image.
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace jlhost {

// Sequential reader: BGZF is a series of gzip members, so plain inflate with member restart reads it.
class BgzfReader {
public:
    explicit BgzfReader(const std::string &path) : f_(fopen(path.c_str(), "rb"))
    {
        if (!f_) throw std::runtime_error("cannot open " + path);
        memset(&z_, 0, sizeof z_);
        if (inflateInit2(&z_, 15 + 32) != Z_OK) throw std::runtime_error("inflateInit2 failed");
        in_.resize(1 << 16);
    }
    ~BgzfReader()
    {
        inflateEnd(&z_);
        if (f_) fclose(f_);
    }
    // read exactly n bytes; returns false on clean EOF at a record boundary (n bytes not started)
    bool read(void *dst, size_t n)
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

private:
    FILE *f_;
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
            if (buf_.size() == kBlock) flush_block();
        }
    }
    void close()
    {
        if (!f_) return;
        if (!buf_.empty()) flush_block();
        static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0,
                                        0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        fwrite(eof, 1, sizeof eof, f_);
        fclose(f_);
        f_ = nullptr;
    }

private:
    static constexpr size_t kBlock = 0xff00;
    void flush_block()
    {
        std::vector<uint8_t> comp(compressBound((uLong)buf_.size()) + 64);
        z_stream z;
        memset(&z, 0, sizeof z);
        if (deflateInit2(&z, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw std::runtime_error("deflateInit2");
        z.next_in = buf_.data();
        z.avail_in = (uInt)buf_.size();
        z.next_out = comp.data();
        z.avail_out = (uInt)comp.size();
        if (deflate(&z, Z_FINISH) != Z_STREAM_END) throw std::runtime_error("deflate");
        const size_t clen = z.total_out;
        deflateEnd(&z);
        const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), buf_.data(), (uInt)buf_.size());
        const uint16_t bsize = (uint16_t)(clen + 25);  // total block size - 1
        uint8_t hdr[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0, 0};
        hdr[16] = (uint8_t)(bsize & 0xff);
        hdr[17] = (uint8_t)(bsize >> 8);
        fwrite(hdr, 1, 18, f_);
        fwrite(comp.data(), 1, clen, f_);
        uint8_t tail[8];
        const uint32_t isize = (uint32_t)buf_.size();
        memcpy(tail, &crc, 4);
        memcpy(tail + 4, &isize, 4);
        fwrite(tail, 1, 8, f_);
        buf_.clear();
    }
    FILE *f_;
    std::vector<uint8_t> buf_;
};

}  // namespace jlhost
