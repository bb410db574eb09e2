// crc32_fold.hpp — the CRC-32 of a BGZF block's inflated bytes (RFC 1952: polynomial 0xEDB88320, bit-reflected) by carry-less
// multiplication: four 16-byte lanes folded 64 bytes at a time, then 128 -> 64 -> 32 bits with a Barrett reduction — the
// scheme of Intel's "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ" with the constants it gives for this
// polynomial.  About ten times zlib 1.2.11's table walk (1 GB/s per core), which would have added half as much CPU time to
// the decode as the inflate itself.  The tail that is not a multiple of 16 bytes, short inputs and CPUs without PCLMULQDQ go
// through zlib's crc32; tests/cpp/crc_check.cpp compares the two on random lengths and alignments.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <zlib.h>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace jlhost {

#if defined(__x86_64__)
// len >= 64 and a multiple of 16; crc in, crc out are the register values (the caller complements)
__attribute__((target("pclmul,sse4.1"))) static inline uint32_t crc32_fold_blocks(const uint8_t *buf, size_t len, uint32_t crc)
{
    const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596ll, 0x0154442bd4ll);
    const __m128i k3k4 = _mm_set_epi64x(0x00ccaa009ell, 0x01751997d0ll);
    const __m128i k5k0 = _mm_set_epi64x(0, 0x0163cd6124ll);
    const __m128i poly = _mm_set_epi64x(0x01f7011641ll, 0x01db710641ll);
    __m128i x1 = _mm_loadu_si128((const __m128i *)(buf + 0x00)), x2 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
    __m128i x3 = _mm_loadu_si128((const __m128i *)(buf + 0x20)), x4 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    buf += 64;
    len -= 64;
    while (len >= 64) {   // four lanes, 64 bytes a step
        const __m128i a1 = _mm_clmulepi64_si128(x1, k1k2, 0x00), a2 = _mm_clmulepi64_si128(x2, k1k2, 0x00);
        const __m128i a3 = _mm_clmulepi64_si128(x3, k1k2, 0x00), a4 = _mm_clmulepi64_si128(x4, k1k2, 0x00);
        x1 = _mm_clmulepi64_si128(x1, k1k2, 0x11);
        x2 = _mm_clmulepi64_si128(x2, k1k2, 0x11);
        x3 = _mm_clmulepi64_si128(x3, k1k2, 0x11);
        x4 = _mm_clmulepi64_si128(x4, k1k2, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, a1), _mm_loadu_si128((const __m128i *)(buf + 0x00)));
        x2 = _mm_xor_si128(_mm_xor_si128(x2, a2), _mm_loadu_si128((const __m128i *)(buf + 0x10)));
        x3 = _mm_xor_si128(_mm_xor_si128(x3, a3), _mm_loadu_si128((const __m128i *)(buf + 0x20)));
        x4 = _mm_xor_si128(_mm_xor_si128(x4, a4), _mm_loadu_si128((const __m128i *)(buf + 0x30)));
        buf += 64;
        len -= 64;
    }
    // the four lanes into one
    __m128i t = _mm_clmulepi64_si128(x1, k3k4, 0x00);
    x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), x2), t);
    t = _mm_clmulepi64_si128(x1, k3k4, 0x00);
    x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), x3), t);
    t = _mm_clmulepi64_si128(x1, k3k4, 0x00);
    x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), x4), t);
    while (len >= 16) {   // single 16-byte steps
        t = _mm_clmulepi64_si128(x1, k3k4, 0x00);
        x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), _mm_loadu_si128((const __m128i *)buf)), t);
        buf += 16;
        len -= 16;
    }
    // 128 -> 64 bits
    const __m128i mask = _mm_setr_epi32(~0, 0, ~0, 0);
    x2 = _mm_clmulepi64_si128(x1, k3k4, 0x10);
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), x2);
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, mask);
    x1 = _mm_xor_si128(_mm_clmulepi64_si128(x1, k5k0, 0x00), x2);
    // Barrett reduction to 32 bits
    x2 = _mm_and_si128(x1, mask);
    x2 = _mm_clmulepi64_si128(x2, poly, 0x10);
    x2 = _mm_and_si128(x2, mask);
    x2 = _mm_clmulepi64_si128(x2, poly, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
#endif

// the CRC-32 zlib's crc32(0, buf, len) returns
static inline uint32_t crc32_of(const uint8_t *buf, size_t len)
{
#if defined(__x86_64__)
    static const bool have = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    if (have && len >= 64) {
        const size_t body = len & ~(size_t)15;
        const uint32_t c = ~crc32_fold_blocks(buf, body, ~0u);
        return body == len ? c : (uint32_t)crc32(c, buf + body, (uInt)(len - body));
    }
#endif
    return (uint32_t)crc32(0L, buf, (uInt)len);
}

}  // namespace jlhost
