// capi_merge.hip — the host side of phasing sharded by reads (SURVEY §8e option A): merge of the per-window variant
// tables, merge of the per-slice group tables, selection of the haplotypes on the MERGED counts, and the schedule of
// the column-slice exchange as data.  No device code and no HIP call: everything here runs wherever the library loads.
// Behaviour: doc/JULIET.md:192-211 (haplotype ids, haplotype_hit), :253-254 (>= 10 reads, here of the merged count),
// :372-381 (read categories); docs/SPEC.md §8 (order: count descending, then pattern ascending).
#include <string.h>

#include <algorithm>
#include <numeric>
#include <vector>

#include "xwin_schedule.h"

extern "C" {

int jl_merge_tables(const jl_variant *const *tables, const uint32_t *counts, const uint32_t *win_begin, uint32_t n_tables,
                    jl_variant *merged, uint32_t cap, uint32_t *n_merged)
{
    if ((!tables || !counts || !win_begin) && n_tables) return JL_ERR_ARG;
    if (!n_merged || (!merged && cap)) return JL_ERR_ARG;
    uint64_t total = 0;
    for (uint32_t t = 0; t < n_tables; ++t) {
        if (counts[t] && !tables[t]) return JL_ERR_ARG;
        total += counts[t];
    }
    *n_merged = (uint32_t)std::min<uint64_t>(total, 0xFFFFFFFFu);
    if (total > cap) return JL_ERR_OVERFLOW;
    uint32_t o = 0;
    for (uint32_t t = 0; t < n_tables; ++t)
        for (uint32_t r = 0; r < counts[t]; ++r) {
            merged[o] = tables[t][r];
            merged[o].col += win_begin[t];
            ++o;
        }
    // every codon is evaluated by exactly one window, so the keys are distinct; a stable sort keeps the windows' order
    // for callers that pass overlapping tables all the same
    std::stable_sort(merged, merged + o, [](const jl_variant &a, const jl_variant &b) {
        if (a.gene != b.gene) return a.gene < b.gene;
        if (a.codon_pos != b.codon_pos) return a.codon_pos < b.codon_pos;
        return a.codon < b.codon;
    });
    return JL_OK;
}

int jl_merge_groups(const uint8_t *const *patterns, const uint32_t *pattern_stride, const uint32_t *const *counts,
                    const uint32_t *n_groups, uint32_t n_tables, uint32_t vp, uint8_t *merged_patterns, uint64_t *merged_counts,
                    uint32_t cap, uint32_t *n_merged, uint32_t *const *index)
{
    if (!n_merged || ((!patterns || !pattern_stride || !counts || !n_groups) && n_tables)) return JL_ERR_ARG;
    size_t total = 0;
    for (uint32_t t = 0; t < n_tables; ++t) {
        if (n_groups[t] && (!counts[t] || (vp && (!patterns[t] || pattern_stride[t] < vp)))) return JL_ERR_ARG;
        total += n_groups[t];
    }
    // A group = (pattern, table, row).  Patterns of up to eight positions compare as ONE big-endian 64-bit word (codon
    // codes are bytes, so word order is position-by-position order); longer ones by memcmp.
    struct ref_t { uint64_t key; uint32_t t, q; };
    std::vector<ref_t> all;
    all.reserve(total);
    const bool by_word = vp <= 8u;
    for (uint32_t t = 0; t < n_tables; ++t)
        for (uint32_t q = 0; q < n_groups[t]; ++q) {
            uint64_t k = 0;
            if (by_word) {
                const uint8_t *p = patterns[t] + (size_t)q * pattern_stride[t];
                for (uint32_t j = 0; j < vp; ++j) k = (k << 8) | p[j];
            }
            all.push_back({k, t, q});
        }
    auto pat = [&](const ref_t &r) { return patterns[r.t] + (size_t)r.q * pattern_stride[r.t]; };
    if (by_word)
        std::sort(all.begin(), all.end(), [](const ref_t &a, const ref_t &b) {
            if (a.key != b.key) return a.key < b.key;
            return a.t != b.t ? a.t < b.t : a.q < b.q;
        });
    else
        std::sort(all.begin(), all.end(), [&](const ref_t &a, const ref_t &b) {
            const int c = memcmp(pat(a), pat(b), vp);
            if (c) return c < 0;
            return a.t != b.t ? a.t < b.t : a.q < b.q;
        });
    uint32_t m = 0;
    for (size_t i = 0; i < all.size(); ++i) {
        const bool fresh = i == 0 || (by_word ? all[i].key != all[i - 1].key : memcmp(pat(all[i]), pat(all[i - 1]), vp) != 0);
        if (fresh) {
            if (m < cap && merged_patterns && vp) memcpy(merged_patterns + (size_t)m * vp, pat(all[i]), vp);
            if (m < cap && merged_counts) merged_counts[m] = 0;
            ++m;
        }
        if (m <= cap && merged_counts) merged_counts[m - 1] += counts[all[i].t][all[i].q];
        if (index && index[all[i].t]) index[all[i].t][all[i].q] = m - 1;
    }
    *n_merged = m;
    return (m > cap && (merged_patterns || merged_counts)) ? JL_ERR_OVERFLOW : JL_OK;
}

int jl_select_haplotypes(const uint8_t *patterns, const uint64_t *counts, uint32_t n_groups, uint32_t vp, const jl_variant *variants,
                         uint32_t n_var, const uint32_t *pos_cols, uint32_t min_reads, const jl_phase_summary *partials,
                         uint32_t n_partials, jl_phase_summary *summary, uint32_t *hap_count, uint8_t *hap_pattern, uint8_t *hit,
                         uint32_t hit_stride, uint32_t *cooc, uint16_t *hap_of_group)
{
    if ((n_groups && (!counts || (vp && !patterns))) || (n_var && !variants) || (vp && !pos_cols) || (n_partials && !partials))
        return JL_ERR_ARG;
    std::vector<uint32_t> order;
    uint64_t clean = 0;
    for (uint32_t q = 0; q < n_groups; ++q) {
        clean += counts[q];
        if (counts[q] >= min_reads) order.push_back(q);
    }
    // (merged groups arrive in ascending pattern order, so the index breaks ties the way the pattern would; callers that
    // pass unsorted groups get the comparison itself)
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        if (counts[a] != counts[b]) return counts[a] > counts[b];
        const int c = vp ? memcmp(patterns + (size_t)a * vp, patterns + (size_t)b * vp, vp) : 0;
        return c ? c < 0 : a < b;
    });
    if (order.size() > JL_MAX_HAPLOTYPES) order.resize(JL_MAX_HAPLOTYPES);   // at most 702 have names (J:198)
    const uint32_t H = (uint32_t)order.size();
    if (hit && H > hit_stride) return JL_ERR_OVERFLOW;
    if (hap_of_group)
        for (uint32_t q = 0; q < n_groups; ++q) hap_of_group[q] = (uint16_t)JL_HAP_INSUFFICIENT;
    uint64_t reported = 0;
    for (uint32_t h = 0; h < H; ++h) {
        const uint32_t q = order[h];
        reported += counts[q];
        if (hap_of_group) hap_of_group[q] = (uint16_t)h;
        if (hap_count) hap_count[h] = (uint32_t)counts[q];
        if (hap_pattern && vp) memcpy(hap_pattern + (size_t)h * vp, patterns + (size_t)q * vp, vp);
    }
    // hit[v][h]: haplotype h carries variant v's codon at v's position (J:207-209)
    std::vector<uint8_t> hit_own;
    const uint8_t *hitp = hit;
    uint32_t stride = hit_stride;
    if ((hit || cooc) && n_var) {
        if (!hit) { hit_own.assign((size_t)n_var * std::max(1u, H), 0); stride = std::max(1u, H); }
        uint8_t *w = hit ? hit : hit_own.data();
        for (uint32_t v = 0; v < n_var; ++v) {
            uint32_t k = 0xFFFFFFFFu;
            for (uint32_t p = 0; p < vp; ++p)
                if (pos_cols[p] == variants[v].col) { k = p; break; }
            for (uint32_t h = 0; h < H; ++h)
                w[(size_t)v * stride + h] = (k != 0xFFFFFFFFu && patterns[(size_t)order[h] * vp + k] == variants[v].codon) ? 1 : 0;
        }
        hitp = w;
    }
    if (cooc)
        for (uint32_t v = 0; v < n_var; ++v)
            for (uint32_t w = 0; w < n_var; ++w) {
                uint64_t s = 0;
                for (uint32_t h = 0; h < H; ++h)
                    if (hitp[(size_t)v * stride + h] && hitp[(size_t)w * stride + h]) s += counts[order[h]];
                cooc[(size_t)v * n_var + w] = (uint32_t)s;
            }
    if (summary) {
        jl_phase_summary s;
        memset(&s, 0, sizeof s);
        s.reported_reads = (uint32_t)reported;
        s.insufficient_reads = (uint32_t)(clean - reported);
        for (uint32_t k = 0; k < n_partials; ++k) {
            s.damaged_reads += partials[k].damaged_reads;
            s.marginal_gap += partials[k].marginal_gap;
            s.marginal_heteroduplex += partials[k].marginal_heteroduplex;
            s.marginal_partial += partials[k].marginal_partial;
        }
        s.n_positions = vp;
        s.n_haplotypes = H;
        *summary = s;
    }
    return JL_OK;
}

int jl_xwin_slice_plan(const uint32_t *win_begin, const uint32_t *win_ncols, const int32_t *win_rank, uint32_t n_windows,
                       const jl_variant *merged, uint32_t n_var, const uint64_t *slice_begin, int32_t world, int32_t rank,
                       jl_xwin_op *ops, uint32_t cap_ops, uint32_t *n_ops)
{
    if (!win_begin || !win_ncols || !win_rank || !n_windows || (!merged && n_var) || !slice_begin || !n_ops || world < 1 ||
        rank < 0 || rank >= world || (!ops && cap_ops))
        return JL_ERR_ARG;
    xwin_schedule sch;
    std::string err;
    int rc = xwin_make_schedule(win_begin, win_ncols, win_rank, n_windows, merged, n_var, slice_begin, world, nullptr, &sch, &err);
    if (rc) return rc;
    std::vector<jl_xwin_op> list;
    xwin_ops_of_rank(sch, slice_begin, world, rank, &list);
    *n_ops = (uint32_t)list.size();
    if (list.size() > cap_ops) return JL_ERR_OVERFLOW;
    if (!list.empty()) memcpy(ops, list.data(), list.size() * sizeof(jl_xwin_op));
    return JL_OK;
}

}  // extern "C"
