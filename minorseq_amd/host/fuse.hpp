// fuse.hpp — consensus of a window from the column pileup and the insertion counters, the documented scope of the
// reference's `fuse` tool (doc/FUSE.md:17-20): "creation of a high-quality consensus sequence.  Fuse includes in-frame
// insertions with a certain distance to each other.  Major deletions are being removed."  docs/SPEC.md §11; the
// fraction and the distance are UNPINNED and therefore flags of the front end.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace jlhost {

// col_counts[n_cols][6] (A C G T - N), len_hist[n_cols][32], base_counts[n_cols][30][4]; the insertion arrays may be
// empty (no tracking).  Per column: an accepted insertion's bases, then the majority of A C G T - (lowest code on
// ties) — '-' drops the column, no coverage prints N.
inline std::string fuse_consensus(uint32_t n_cols, const std::vector<uint32_t> &col_counts, const std::vector<uint32_t> &len_hist,
                                  const std::vector<uint32_t> &base_counts, double min_frac, uint32_t min_distance)
{
    std::string out;
    int64_t last_ins = -(int64_t)min_distance - 1;
    const bool have_ins = len_hist.size() >= (size_t)n_cols * 32 && base_counts.size() >= (size_t)n_cols * 120;
    for (uint32_t c = 0; c < n_cols; ++c) {
        const uint32_t *k = &col_counts[(size_t)c * 6];
        const uint32_t covering = k[0] + k[1] + k[2] + k[3] + k[4] + k[5];
        if (have_ins && covering) {
            uint32_t bestL = 0, bestN = 0;
            for (uint32_t L = 3; L <= 30; L += 3) {   // in-frame lengths; the most frequent, the shorter on ties
                const uint32_t v = len_hist[(size_t)c * 32 + L];
                if (v > bestN) { bestN = v; bestL = L; }
            }
            if (bestL && (double)bestN > min_frac * (double)covering && (int64_t)c - last_ins >= (int64_t)min_distance) {
                for (uint32_t j = 0; j < bestL; ++j) {
                    const uint32_t *b = &base_counts[((size_t)c * 30 + j) * 4];
                    uint32_t best = 0;
                    for (uint32_t s = 1; s < 4; ++s)
                        if (b[s] > b[best]) best = s;
                    out += "ACGT"[best];
                }
                last_ins = c;
            }
        }
        uint32_t best = 0, bv = k[0];
        for (uint32_t s = 1; s < 5; ++s)
            if (k[s] > bv) { bv = k[s]; best = s; }
        if (bv == 0) out += 'N';
        else if (best < 4) out += "ACGT"[best];
    }
    return out;
}

}  // namespace jlhost
