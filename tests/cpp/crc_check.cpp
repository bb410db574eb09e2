// crc32_of (minorseq_amd/host/crc32_fold.hpp) against zlib's crc32 on random contents, lengths 0..70000 and alignments.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../minorseq_amd/host/crc32_fold.hpp"
int main()
{
    std::vector<uint8_t> b(70100);
    uint64_t s = 88172645463325252ull;
    for (auto &x : b) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (uint8_t)(s >> 24); }
    int bad = 0, n = 0;
    for (size_t len = 0; len <= 70000; len += (len < 300 ? 1 : 977)) {
        for (size_t off = 0; off < 19; off += 3) {
            const uint32_t a = jlhost::crc32_of(b.data() + off, len), z = (uint32_t)crc32(0L, b.data() + off, (uInt)len);
            ++n;
            if (a != z) { if (++bad < 5) printf("mismatch len %zu off %zu: %08x vs %08x\n", len, off, a, z); }
        }
    }
    printf("%d comparisons, %d mismatches\n", n, bad);
    return bad ? 1 : 0;
}
