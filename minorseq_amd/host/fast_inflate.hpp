// fast_inflate.hpp — raw DEFLATE (RFC 1951) for one BGZF block: the whole compressed body and the exact inflated size
// are in hand (SAM spec §4.1: BSIZE and ISIZE frame every block), so the decoder works buffer to buffer with no
// streaming state.  zlib's inflate() runs 0.37 GB/s per core on CCS BAM blocks and was the floor of `juliet`'s decode
// stage (doc/JULIET.md:50-58 is the input); this one is built for 64-bit hosts:
//   * a 64-bit bit buffer refilled by one unaligned 8-byte load (the bytes above the valid count are the stream's own
//     next bytes, so OR-ing them in again is harmless);
//   * one table lookup per symbol: 11 root bits for literals/lengths, 8 for distances, second-level tables for the rare
//     longer codes; an entry carries the symbol's base value and its extra-bit count, so no second table is consulted;
//   * up to three literals per refill; matches copied sixteen bytes at a time;
//   * the unchecked loop runs while 16 input bytes and 320 output bytes remain; the last stretch of every block takes
//     the checked loop, so nothing is read or written outside [in, in+in_len) and [out, out+out_len).
// Malformed input returns an error; it never reads or writes out of bounds (tests/cpp/inflate_check.cpp, run by
// tests/test_host_frontend.py under AddressSanitizer: every level and strategy of zlib's deflate, then corrupted and
// truncated copies, which must get zlib's verdict and zlib's bytes).  Measured on this container: 1.6 GB/s against
// zlib's 0.7-0.85 GB/s on the synthetic CCS BAM's blocks, one core.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

namespace jlz {

class Inflater {
public:
    // 0 when the stream ended with its final block exactly at out_len bytes and at the end of the input; negative for
    // malformed or mis-sized input
    int run(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len)
    {
        const uint8_t *ip = in, *const iend = in + in_len;
        uint8_t *op = out, *const oend = out + out_len;
        uint64_t bb = 0;     // bit buffer: the next stream bit is bit 0
        unsigned bc = 0;     // valid bits in bb
        unsigned over = 0;   // zero bytes supplied past the end of the input

#define JLZ_NEED(n)                                                                                                              \
    do {                                                                                                                         \
        while (bc < (n)) {                                                                                                       \
            if (ip < iend) bb |= (uint64_t)*ip++ << bc;                                                                          \
            else if (++over > 16) return -2;                                                                                     \
            bc += 8;                                                                                                             \
        }                                                                                                                        \
    } while (0)
#define JLZ_DROP(n)                                                                                                              \
    do {                                                                                                                         \
        bb >>= (n);                                                                                                              \
        bc -= (n);                                                                                                               \
    } while (0)
#define JLZ_REFILL()                                                                                                             \
    do {                                                                                                                         \
        uint64_t w_;                                                                                                             \
        memcpy(&w_, ip, 8);                                                                                                      \
        bb |= w_ << bc;                                                                                                          \
        ip += (63u - bc) >> 3;                                                                                                   \
        bc |= 56u;                                                                                                               \
    } while (0)

        for (;;) {
            JLZ_NEED(3);
            const unsigned final_block = (unsigned)bb & 1u, type = ((unsigned)bb >> 1) & 3u;
            JLZ_DROP(3);
            if (type == 0) {
                // stored: to the byte boundary, hand the whole bytes back, LEN / ~LEN, copy
                if (over) return -3;
                JLZ_DROP(bc & 7u);
                ip -= bc >> 3;
                bb = 0;
                bc = 0;
                if (iend - ip < 4) return -3;
                const unsigned len = ip[0] | (ip[1] << 8), nlen = ip[2] | (ip[3] << 8);
                ip += 4;
                if ((len ^ 0xFFFFu) != nlen) return -3;
                if ((size_t)(iend - ip) < len || (size_t)(oend - op) < len) return -3;
                memcpy(op, ip, len);
                ip += len;
                op += len;
                if (final_block) break;
                continue;
            }
            if (type == 3) return -4;
            if (type == 1) {
                if (!fixed_) {
                    uint8_t lens[288 + 32];
                    for (unsigned i = 0; i < 144; ++i) lens[i] = 8;
                    for (unsigned i = 144; i < 256; ++i) lens[i] = 9;
                    for (unsigned i = 256; i < 280; ++i) lens[i] = 7;
                    for (unsigned i = 280; i < 288; ++i) lens[i] = 8;
                    for (unsigned i = 0; i < 32; ++i) lens[288 + i] = 5;
                    if (build(ll_, kLR, kLCap, lens, 288, kLitLen) || build(dd_, kDR, kDCap, lens + 288, 32, kDist)) return -5;
                    fixed_ = true;
                }
            } else {
                fixed_ = false;
                JLZ_NEED(14);
                const unsigned hlit = ((unsigned)bb & 31u) + 257u, hdist = (((unsigned)bb >> 5) & 31u) + 1u,
                               hclen = (((unsigned)bb >> 10) & 15u) + 4u;
                JLZ_DROP(14);
                if (hlit > 286u || hdist > 30u) return -6;
                static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                uint8_t plens[19];
                memset(plens, 0, sizeof plens);
                for (unsigned i = 0; i < hclen; ++i) {
                    JLZ_NEED(3);
                    plens[order[i]] = (uint8_t)(bb & 7u);
                    JLZ_DROP(3);
                }
                if (build(pre_, kPR, 1u << kPR, plens, 19, kPre)) return -7;
                uint8_t lens[286 + 30 + 138];
                const unsigned total = hlit + hdist;
                unsigned i = 0;
                while (i < total) {
                    JLZ_NEED(14);
                    const uint32_t e = pre_[bb & ((1u << kPR) - 1u)];
                    if (e & kExc) return -7;
                    JLZ_DROP(e & 31u);
                    const unsigned sym = e >> 16;
                    if (sym < 16) {
                        lens[i++] = (uint8_t)sym;
                        continue;
                    }
                    unsigned rep;
                    uint8_t v = 0;
                    if (sym == 16) {
                        if (!i) return -7;
                        v = lens[i - 1];
                        rep = 3u + ((unsigned)bb & 3u);
                        JLZ_DROP(2);
                    } else if (sym == 17) {
                        rep = 3u + ((unsigned)bb & 7u);
                        JLZ_DROP(3);
                    } else {
                        rep = 11u + ((unsigned)bb & 127u);
                        JLZ_DROP(7);
                    }
                    if (i + rep > total) return -7;
                    memset(lens + i, v, rep);
                    i += rep;
                }
                if (lens[256] == 0) return -7;   // a block must be able to end
                if (build(ll_, kLR, kLCap, lens, hlit, kLitLen) || build(dd_, kDR, kDCap, lens + hlit, hdist, kDist)) return -8;
            }

            // ---- the block's symbols
            bool ended = false;
            while (iend - ip >= 16 && oend - op >= 320) {
                JLZ_REFILL();
                uint32_t e = ll_[bb & kLMask];
                if (e & kLit) {
                    JLZ_DROP(e & 31u);
                    *op++ = (uint8_t)(e >> 16);
                    e = ll_[bb & kLMask];
                    if (e & kLit) {
                        JLZ_DROP(e & 31u);
                        *op++ = (uint8_t)(e >> 16);
                        e = ll_[bb & kLMask];
                        if (e & kLit) {
                            JLZ_DROP(e & 31u);
                            *op++ = (uint8_t)(e >> 16);
                            continue;
                        }
                    }
                }
                if (e & kExc) {
                    if (e & kSub) {
                        JLZ_DROP(e & 31u);
                        e = ll_[((e >> 12) & 0xFFFFu) + ((unsigned)bb & ((1u << ((e >> 8) & 15u)) - 1u))];
                    }
                    if (e & kExc) {
                        if (!(e & kEob)) return -9;
                        JLZ_DROP(e & 31u);
                        ended = true;
                        break;
                    }
                    if (e & kLit) {
                        JLZ_DROP(e & 31u);
                        *op++ = (uint8_t)(e >> 16);
                        continue;
                    }
                }
                JLZ_DROP(e & 31u);
                const unsigned lx = (e >> 8) & 15u;
                const unsigned len = ((e >> 16) & 0x1FFu) + ((unsigned)bb & ((1u << lx) - 1u));
                JLZ_DROP(lx);
                JLZ_REFILL();
                uint32_t d = dd_[bb & kDMask];
                if (d & kExc) {
                    if (!(d & kSub)) return -10;
                    JLZ_DROP(d & 31u);
                    d = dd_[((d >> 12) & 0xFFFFu) + ((unsigned)bb & ((1u << ((d >> 8) & 15u)) - 1u))];
                    if (d & kExc) return -10;
                }
                JLZ_DROP(d & 31u);
                const unsigned dx = (d >> 8) & 15u;
                const unsigned dist = ((d >> 12) & 0x7FFFu) + ((unsigned)bb & ((1u << dx) - 1u));
                JLZ_DROP(dx);
                if (dist > (size_t)(op - out)) return -11;
                const uint8_t *s = op - dist;
                uint8_t *t = op;
                op += len;
                if (dist >= 16) {
                    // sixteen bytes a step (every load lies wholly behind the bytes its step writes)
                    struct w16 { uint64_t a, b; } w;
                    memcpy(&w, s, 16);
                    memcpy(t, &w, 16);
                    for (unsigned k = 16; k < len; k += 16) {
                        memcpy(&w, s + k, 16);
                        memcpy(t + k, &w, 16);
                    }
                } else if (dist >= 8) {
                    uint64_t w;
                    memcpy(&w, s, 8);
                    memcpy(t, &w, 8);
                    memcpy(&w, s + 8, 8);
                    memcpy(t + 8, &w, 8);
                    for (unsigned k = 16; k < len; k += 8) {
                        memcpy(&w, s + k, 8);
                        memcpy(t + k, &w, 8);
                    }
                } else if (dist == 1) {
                    struct w16 { uint64_t a, b; } w;
                    w.a = w.b = 0x0101010101010101ull * s[0];
                    for (unsigned k = 0; k < len; k += 16) memcpy(t + k, &w, 16);
                } else {
                    for (unsigned k = 0; k < len; ++k) t[k] = s[k];
                }
            }
            if (!ended) {
                if (bc < 64) bb &= (1ull << bc) - 1ull;   // (the unchecked refill leaves the stream's next bytes above bc)
                for (;;) {
                    JLZ_NEED(48);
                    uint32_t e = ll_[bb & kLMask];
                    if (e & kExc) {
                        if (e & kSub) {
                            JLZ_DROP(e & 31u);
                            e = ll_[((e >> 12) & 0xFFFFu) + ((unsigned)bb & ((1u << ((e >> 8) & 15u)) - 1u))];
                        }
                        if (e & kExc) {
                            if (!(e & kEob)) return -9;
                            JLZ_DROP(e & 31u);
                            break;
                        }
                    }
                    JLZ_DROP(e & 31u);
                    if (e & kLit) {
                        if (op >= oend) return -12;
                        *op++ = (uint8_t)(e >> 16);
                        continue;
                    }
                    const unsigned lx = (e >> 8) & 15u;
                    const unsigned len = ((e >> 16) & 0x1FFu) + ((unsigned)bb & ((1u << lx) - 1u));
                    JLZ_DROP(lx);
                    uint32_t d = dd_[bb & kDMask];   // 48 - 20 = 28 bits left: a distance code and its extra bits
                    if (d & kExc) {
                        if (!(d & kSub)) return -10;
                        JLZ_DROP(d & 31u);
                        d = dd_[((d >> 12) & 0xFFFFu) + ((unsigned)bb & ((1u << ((d >> 8) & 15u)) - 1u))];
                        if (d & kExc) return -10;
                    }
                    JLZ_DROP(d & 31u);
                    const unsigned dx = (d >> 8) & 15u;
                    const unsigned dist = ((d >> 12) & 0x7FFFu) + ((unsigned)bb & ((1u << dx) - 1u));
                    JLZ_DROP(dx);
                    if (dist > (size_t)(op - out) || len > (size_t)(oend - op)) return -11;
                    const uint8_t *s = op - dist;
                    for (unsigned k = 0; k < len; ++k) op[k] = s[k];
                    op += len;
                }
            }
            if (over * 8u > bc) return -2;   // symbols were read out of the padding
            if (final_block) break;
        }
        if (over * 8u > bc) return -2;
        if (op != oend) return -13;
        // the body must end with its final block: whole bytes still in the bit buffer (or behind ip) are not part of a
        // stream a BGZF writer made, and a block that is damaged but happens to inflate to ISIZE bytes often leaves some
        if (ip - ((ptrdiff_t)(bc >> 3) - (ptrdiff_t)over) != iend) return -14;
        return 0;
#undef JLZ_NEED
#undef JLZ_DROP
#undef JLZ_REFILL
    }

private:
    // Table entries (32 bits).  bits 0-4: code bits this lookup consumes.
    //   literal           kLit | value << 16
    //   length            base << 16 | extra-bit count << 8
    //   distance          base << 12 | extra-bit count << 8
    //   second level      kExc | kSub | first index << 12 | index bits << 8   (consumes the root bits)
    //   end of block      kExc | kEob
    //   no such code      kExc
    static constexpr uint32_t kLit = 1u << 31, kExc = 1u << 30, kSub = 1u << 29, kEob = 1u << 28;
    static constexpr unsigned kLR = 11, kDR = 8, kPR = 7;
    static constexpr uint32_t kLMask = (1u << kLR) - 1u, kDMask = (1u << kDR) - 1u;
    // second-level tables all take (longest code - root) bits; a complete code puts at least two codes behind every
    // long prefix: at most 143 tables of 16 entries for 286 literal/length codes, 15 of 128 for 30 distance codes
    static constexpr unsigned kLCap = (1u << kLR) + 144u * 16u, kDCap = (1u << kDR) + 16u * 128u;
    enum Kind { kLitLen, kDist, kPre };

    static uint32_t entry_of(Kind kind, unsigned sym)
    {
        static const uint16_t lbase[29] = {3,  4,  5,  6,  7,  8,  9,  10, 11,  13,  15,  17,  19,  23, 27,
                                           31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        static const uint16_t dbase[30] = {1,   2,   3,   4,   5,   7,    9,    13,   17,   25,   33,   49,   65,    97,    129,
                                           193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        if (kind == kPre) return (uint32_t)sym << 16;
        if (kind == kDist) return sym < 30 ? ((uint32_t)dbase[sym] << 12) | ((uint32_t)dext[sym] << 8) : kExc;
        if (sym < 256) return kLit | ((uint32_t)sym << 16);
        if (sym == 256) return kExc | kEob;
        if (sym < 286) return ((uint32_t)lbase[sym - 257] << 16) | ((uint32_t)lext[sym - 257] << 8);
        return kExc;
    }

    static unsigned reverse_bits(unsigned v, unsigned n)
    {
        v = ((v & 0x5555u) << 1) | ((v >> 1) & 0x5555u);
        v = ((v & 0x3333u) << 2) | ((v >> 2) & 0x3333u);
        v = ((v & 0x0F0Fu) << 4) | ((v >> 4) & 0x0F0Fu);
        v = ((v & 0x00FFu) << 8) | ((v >> 8) & 0x00FFu);
        return v >> (16u - n);
    }

    // canonical Huffman code of lens[0..n) into a root table of 2^root entries plus second-level tables; nonzero when
    // the lengths over-subscribe the code space, leave it incomplete (other than the one-code case RFC 1951 allows) or
    // need more room than cap
    static int build(uint32_t *tab, unsigned root, unsigned cap, const uint8_t *lens, unsigned n, Kind kind)
    {
        unsigned count[16];
        memset(count, 0, sizeof count);
        for (unsigned i = 0; i < n; ++i) ++count[lens[i] & 15u];
        unsigned maxlen = 15;
        while (maxlen && !count[maxlen]) --maxlen;
        for (unsigned i = 0; i < (1u << root); ++i) tab[i] = kExc;
        if (!maxlen) return 0;   // no codes at all (a block of literals needs no distance code)
        int left = 1;
        for (unsigned l = 1; l <= 15; ++l) {
            left = (left << 1) - (int)count[l];
            if (left < 0) return -1;
        }
        if (left > 0 && (kind == kPre || maxlen != 1)) return -1;
        unsigned offs[16];
        offs[1] = 0;
        for (unsigned l = 1; l < 15; ++l) offs[l + 1] = offs[l] + count[l];
        uint16_t sorted[288];
        for (unsigned i = 0; i < n; ++i)
            if (lens[i]) sorted[offs[lens[i]]++] = (uint16_t)i;
        const unsigned sub_bits = maxlen > root ? maxlen - root : 0;
        unsigned next = 1u << root, code = 0, at = 0;
        for (unsigned l = 1; l <= maxlen; ++l, code <<= 1)
            for (unsigned c = 0; c < count[l]; ++c, ++code, ++at) {
                const unsigned rev = reverse_bits(code, l);
                const uint32_t e = entry_of(kind, sorted[at]);
                if (l <= root) {
                    for (unsigned k = rev; k < (1u << root); k += 1u << l) tab[k] = e | l;
                    continue;
                }
                const unsigned prefix = rev & ((1u << root) - 1u);
                if (!(tab[prefix] & kSub)) {
                    if (next + (1u << sub_bits) > cap) return -1;
                    for (unsigned k = 0; k < (1u << sub_bits); ++k) tab[next + k] = kExc;
                    tab[prefix] = kExc | kSub | (next << 12) | (sub_bits << 8) | root;
                    next += 1u << sub_bits;
                }
                const unsigned first = (tab[prefix] >> 12) & 0xFFFFu;
                for (unsigned k = rev >> root; k < (1u << sub_bits); k += 1u << (l - root)) tab[first + k] = e | (l - root);
            }
        return 0;
    }

    uint32_t ll_[kLCap];
    uint32_t dd_[kDCap];
    uint32_t pre_[1u << kPR];
    bool fixed_ = false;
};

}  // namespace jlz
