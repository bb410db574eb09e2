// inflate_check.cpp — jlz::Inflater (minorseq_amd/host/fast_inflate.hpp) against zlib's inflate on the same raw DEFLATE
// streams: generated inputs at every level/strategy, then corrupted and truncated copies (the two must give the same
// verdict and, when both accept, the same bytes).  `bench <file.bam>` times the two over a BGZF file's blocks.
#include <zlib.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "fast_inflate.hpp"

typedef std::vector<uint8_t> bytes;

static bytes deflate_raw(const bytes &src, int level, int strategy)
{
    z_stream z;
    memset(&z, 0, sizeof z);
    if (deflateInit2(&z, level, Z_DEFLATED, -15, 8, strategy) != Z_OK) abort();
    bytes out(deflateBound(&z, src.size()) + 64);
    z.next_in = const_cast<uint8_t *>(src.data());
    z.avail_in = (uInt)src.size();
    z.next_out = out.data();
    z.avail_out = (uInt)out.size();
    if (deflate(&z, Z_FINISH) != Z_STREAM_END) abort();
    out.resize(z.total_out);
    deflateEnd(&z);
    return out;
}

static bool zlib_inflate(const bytes &comp, bytes &out, size_t n)
{
    out.assign(n + 1, 0xAA);   // one spare byte: a stream that wants more than n is a failure, not a short success
    z_stream z;
    memset(&z, 0, sizeof z);
    if (inflateInit2(&z, -15) != Z_OK) abort();
    z.next_in = const_cast<uint8_t *>(comp.data());
    z.avail_in = (uInt)comp.size();
    z.next_out = out.data();
    z.avail_out = (uInt)(n + 1);
    const int rc = inflate(&z, Z_FINISH);
    const bool ok = rc == Z_STREAM_END && z.total_out == n && z.avail_in == 0;   // (bytes behind the final block: not a BGZF body)
    inflateEnd(&z);
    out.resize(n);
    return ok;
}

static bool mine_inflate(const bytes &comp, bytes &out, size_t n)
{
    // exact-size heap blocks, so AddressSanitizer sees any access outside [in, in+len) and [out, out+n)
    uint8_t *in = (uint8_t *)malloc(comp.size() ? comp.size() : 1);
    memcpy(in, comp.data(), comp.size());
    uint8_t *o = (uint8_t *)malloc(n ? n : 1);
    memset(o, 0xAA, n);
    static jlz::Inflater inf;
    const int rc = inf.run(in, comp.size(), o, n);
    out.assign(o, o + n);
    free(in);
    free(o);
    return rc == 0;
}

static bytes make(int kind, size_t n, std::mt19937 &rng)
{
    bytes b(n);
    switch (kind) {
    case 0: for (auto &c : b) c = (uint8_t)rng(); break;                                   // incompressible
    case 1: for (auto &c : b) c = "ACGT"[rng() & 3]; break;                                // four symbols
    case 2: for (size_t i = 0; i < n; ++i) b[i] = (uint8_t)(i % 7 == 0 ? rng() : 'q'); break;   // runs
    case 3: break;                                                                          // zeros
    case 4: {                                                                               // repeats at every distance
        size_t i = 0;
        while (i < n) {
            if (i > 8 && (rng() & 3)) {
                size_t dist = 1 + rng() % std::min<size_t>(i, 1 + (rng() % 33000));
                size_t len = 3 + rng() % 300;
                for (size_t k = 0; k < len && i < n; ++k, ++i) b[i] = b[i - dist];
            } else
                b[i++] = (uint8_t)rng();
        }
        break;
    }
    case 5: {                                                                               // skewed alphabet: long codes
        for (auto &c : b) {
            unsigned r = rng();
            unsigned s = 0;
            while ((r & 1) && s < 250) { ++s; r >>= 1; if (!r) r = rng(); }
            c = (uint8_t)s;
        }
        break;
    }
    default: {                                                                              // BAM-like: names, packed bases, quals
        size_t i = 0;
        while (i < n) {
            char nm[64];
            int l = snprintf(nm, sizeof nm, "m64011_190830/%u/ccs", (unsigned)(rng() % 1000000));
            for (int k = 0; k < l && i < n; ++k) b[i++] = (uint8_t)nm[k];
            for (int k = 0; k < 1500 && i < n; ++k) b[i++] = (uint8_t)(((1u << (rng() & 3)) << 4) | (1u << (rng() & 3)));
            for (int k = 0; k < 3000 && i < n; ++k) b[i++] = (uint8_t)(93 - (rng() % 20 == 0 ? rng() % 60 : 0));
        }
    }
    }
    return b;
}

static int selftest(unsigned seed, int rounds)
{
    std::mt19937 rng(seed);
    const int strategies[] = {Z_DEFAULT_STRATEGY, Z_FILTERED, Z_HUFFMAN_ONLY, Z_RLE, Z_FIXED};
    const size_t sizes[] = {0, 1, 2, 15, 100, 319, 320, 321, 1000, 4096, 65280, 65536, 200000};
    long streams = 0, corrupt = 0, corrupt_ok = 0;
    bytes ref, got;
    for (int kind = 0; kind <= 6; ++kind)
        for (size_t n : sizes) {
            const bytes src = make(kind, n, rng);
            for (int level : {0, 1, 6, 9})
                for (int st : strategies) {
                    const bytes comp = deflate_raw(src, level, st);
                    ++streams;
                    if (!mine_inflate(comp, got, n) || got != src) {
                        fprintf(stderr, "MISMATCH kind %d n %zu level %d strategy %d\n", kind, n, level, st);
                        return 1;
                    }
                    // wrong sizes are refused
                    if (mine_inflate(comp, got, n + 1) || (n && mine_inflate(comp, got, n - 1))) {
                        fprintf(stderr, "wrong size accepted: kind %d n %zu level %d strategy %d\n", kind, n, level, st);
                        return 1;
                    }
                    if (n > 70000 || comp.size() < 4) continue;
                    for (int r = 0; r < rounds; ++r) {
                        bytes bad = comp;
                        const int how = rng() % 3;
                        if (how == 0) bad[rng() % bad.size()] ^= (uint8_t)(1u << (rng() & 7));
                        else if (how == 1) bad.resize(rng() % bad.size());
                        else for (int k = 0; k < 4; ++k) bad[rng() % bad.size()] = (uint8_t)rng();
                        const bool a = zlib_inflate(bad, ref, n), b = mine_inflate(bad, got, n);
                        ++corrupt;
                        corrupt_ok += a;
                        if (a != b || (a && ref != got)) {
                            fprintf(stderr, "corrupted stream: zlib %d mine %d (kind %d n %zu level %d strategy %d how %d)\n", a, b, kind,
                                    n, level, st, how);
                            return 1;
                        }
                    }
                }
        }
    printf("ok: %ld streams, %ld corrupted copies (%ld still valid)\n", streams, corrupt, corrupt_ok);
    return 0;
}

static int bench(const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f) return 2;
    bytes file;
    uint8_t buf[1 << 16];
    size_t k;
    while ((k = fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + k);
    fclose(f);
    struct blk { size_t off, len, isize; };
    std::vector<blk> blks;
    size_t total = 0;
    for (size_t p = 0; p + 18 <= file.size();) {
        const size_t bsize = (file[p + 16] | (file[p + 17] << 8)) + 1u;
        const size_t isize = file[p + bsize - 4] | (file[p + bsize - 3] << 8) | (file[p + bsize - 2] << 16) | ((size_t)file[p + bsize - 1] << 24);
        blks.push_back({p + 18, bsize - 26, isize});
        total += isize;
        p += bsize;
    }
    bytes out(65536 + 64), out2(65536 + 64);
    jlz::Inflater inf;
    for (int pass = 0; pass < 2; ++pass) {
        auto t0 = std::chrono::steady_clock::now();
        for (const blk &b : blks) {
            z_stream z;
            memset(&z, 0, sizeof z);
            inflateInit2(&z, -15);
            z.next_in = file.data() + b.off;
            z.avail_in = (uInt)b.len;
            z.next_out = out.data();
            z.avail_out = (uInt)b.isize;
            if (inflate(&z, Z_FINISH) != Z_STREAM_END) return 3;
            inflateEnd(&z);
        }
        auto t1 = std::chrono::steady_clock::now();
        for (const blk &b : blks)
            if (inf.run(file.data() + b.off, b.len, out2.data(), b.isize)) return 4;
        auto t2 = std::chrono::steady_clock::now();
        const double a = std::chrono::duration<double>(t1 - t0).count(), c = std::chrono::duration<double>(t2 - t1).count();
        printf("%zu blocks, %.1f MB inflated: zlib %.3f s (%.0f MB/s), jlz %.3f s (%.0f MB/s)\n", blks.size(), total / 1e6, a,
               total / 1e6 / a, c, total / 1e6 / c);
    }
    // and the bytes agree
    for (const blk &b : blks) {
        z_stream z;
        memset(&z, 0, sizeof z);
        inflateInit2(&z, -15);
        z.next_in = file.data() + b.off;
        z.avail_in = (uInt)b.len;
        z.next_out = out.data();
        z.avail_out = (uInt)b.isize;
        inflate(&z, Z_FINISH);
        inflateEnd(&z);
        if (inf.run(file.data() + b.off, b.len, out2.data(), b.isize) || memcmp(out.data(), out2.data(), b.isize)) return 5;
    }
    printf("bytes agree\n");
    return 0;
}

int main(int argc, char **argv)
{
    if (argc >= 3 && std::string(argv[1]) == "bench") return bench(argv[2]);
    return selftest(argc >= 2 ? (unsigned)atoi(argv[1]) : 1u, argc >= 3 ? atoi(argv[2]) : 6);
}
