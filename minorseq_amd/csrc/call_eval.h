// call_eval.h — one codon position evaluated by one wavefront (lane = codon): coverage, reference / majority codon,
// error model, Fisher's exact x Bonferroni, filters, and the hand-over of the called codons.
// Shared by call_kernel (histograms read back from HBM) and by the epilogue of the pileup kernel (histogram still in
// LDS: jl_run_async / jl_group_run_async evaluate a position in the workgroup that just counted it).
// Behaviour: doc/JULIET.md:38-42, :133-134, :342-357, :370; docs/SPEC.md §4-7.
#pragma once
#include <string.h>

#include "jl_fisher.h"
#include "jl_internal.h"

__device__ __forceinline__ uint32_t jl_wave_sum_all(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ uint64_t jl_wave_max_all(uint64_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint64_t u = __shfl_xor(v, o, 64);
        v = u > v ? u : v;
    }
    return v;
}

// h = this lane's bin of the position's histogram.  Writes the position's call mask and the finished 48-byte rows of
// its called codons to the staging area [p][codon].  COHERENT: the consumer is another workgroup of the SAME launch
// (call_kernel's last block), so the stores are write-through (agent scope); otherwise plain stores (the consumer is
// the next kernel on the stream).
template <bool COHERENT>
__device__ __forceinline__ void jl_call_position(const jl_call_args &A, uint32_t p, uint32_t col, uint32_t h, uint32_t refcfg,
                                                 uint32_t gene, uint32_t codon_pos, const uint64_t *drm, uint64_t *called,
                                                 jl_variant *staged)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t cov = jl_wave_sum_all(h);
    uint32_t ref = refcfg;
    if (ref == JL_REF_MAJORITY) {
        // argmax, lowest codon index on ties (SPEC §4)
        const uint64_t key = ((uint64_t)h << 8) | (uint64_t)(63u - lane);
        const uint64_t best = jl_wave_max_all(key);
        ref = cov ? 63u - (uint32_t)(best & 0xFFu) : JL_REF_SKIP;
    }
    bool is_called = false;
    double p_adj = 1.0, lp = 0.0;
    uint32_t e = 0;
    if (ref < 64u && h > 0 && lane != ref) {
        double perr = 1.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int sh = 4 - 2 * i;
            perr = perr * ((((ref >> sh) & 3u) == ((lane >> sh) & 3u)) ? A.match : A.substitution);
        }
        const double x = (double)cov * perr;
        double r = A.expected_round == 1 ? floor(x) : (A.expected_round == 2 ? floor(x + 0.5) : ceil(x));
        if (r < 0.0) r = 0.0;
        if (r > (double)cov) r = (double)cov;
        e = (uint32_t)r;
        if (A.tail == 0) {
            // An observed count at or below the expected one has p >= 1/2 (the null is symmetric about K/2
            // because both rows sum to the coverage), so it cannot be called once min(1, n_tests/2) >= alpha;
            // uncalled codons are never reported, so their p-value is not needed.
            const double floor_adj = 0.5 * A.n_tests < 1.0 ? 0.5 * A.n_tests : 1.0;
            if (h > e || !(floor_adj >= A.alpha)) {
                bool skipped;
                const double pv = jl_fisher_greater_equal_rows_or_skip(h, e, cov, A.n_tests, A.alpha, &lp, &skipped);
                p_adj = pv * A.n_tests;
                if (p_adj > 1.0) p_adj = 1.0;
                is_called = !skipped && p_adj < A.alpha;
            }
        } else {
            const double pv = jl_fisher_two_sided_equal_rows(h, e, cov, &lp);
            p_adj = pv * A.n_tests;
            if (p_adj > 1.0) p_adj = 1.0;
            is_called = p_adj < A.alpha;
        }
        const double perc = 100.0 * (double)h / (double)cov;
        if (A.min_perc >= 0.0 && !(perc > A.min_perc)) is_called = false;
        if (A.max_perc >= 0.0 && !(perc < A.max_perc)) is_called = false;
        if (drm && !((drm[p] >> lane) & 1ull)) is_called = false;
    }
    const uint64_t mask = __ballot(is_called);
    if (lane == 0) {
        if (COHERENT) __hip_atomic_store(&called[p], mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else called[p] = mask;
    }
    if (is_called) {
        jl_variant v;
        v.gene = gene;
        v.codon_pos = codon_pos;
        v.col = col;
        v.ref_codon = (uint8_t)ref;
        v.codon = (uint8_t)lane;
        v.flags = 0;
        v.count = h;
        v.coverage = cov;
        v.expected = e;
        v.pad_ = 0;
        v.p_value = p_adj;
        v.log_p = lp;
        uint64_t w[6];
        memcpy(w, &v, sizeof v);
        uint64_t *dst = reinterpret_cast<uint64_t *>(staged + (uint64_t)p * 64u + lane);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            if (COHERENT) __hip_atomic_store(dst + k, w[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else dst[k] = w[k];
        }
    }
}

// Ordered compaction of the staged rows into the fixed-stride table: positions are laid out in (gene, codon) order
// (SPEC §6), so rows in (position, codon index) order are the table.  Call with all 256 threads of a block; `s_scan`
// [4] and `s_running` [1] are LDS words of the caller.  COHERENT as above (masks and rows written by other
// workgroups of this launch are read past the L1).  `lds_cols` (optional, [n_lds]) receives the columns of the first
// rows.  Returns the number of rows needed (may exceed cap; only the first cap are written).
template <bool COHERENT>
__device__ __forceinline__ uint32_t jl_compact_rows_block(uint32_t P, const uint64_t *called, const jl_variant *staged,
                                                          jl_variant *rows, uint32_t cap, uint32_t *s_scan,
                                                          uint32_t *s_running, uint32_t *lds_cols, uint32_t n_lds,
                                                          bool rows_coherent)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    if (tid == 0) *s_running = 0;
    __syncthreads();
    constexpr uint32_t kPer = 8;  // consecutive positions per thread and pass
    for (uint32_t base = 0; base < P; base += 256u * kPer) {
        uint64_t m[kPer];
        uint32_t c = 0;
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k) {
            const uint32_t q = base + tid * kPer + k;
            m[k] = 0ull;
            if (q < P) m[k] = COHERENT ? __hip_atomic_load(&called[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : called[q];
        }
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k) c += (uint32_t)__popcll(m[k]);
        uint32_t inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t u = __shfl_up(inc, o, 64);
            if ((int)lane >= o) inc += u;
        }
        if (lane == 63) s_scan[wid] = inc;
        __syncthreads();
        uint32_t wave_off = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t v = s_scan[w];
            if (w < (int)wid) wave_off += v;
            total += v;
        }
        uint32_t o = *s_running + wave_off + inc - c;
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k) {
            uint64_t mk = m[k];
            const uint32_t q = base + tid * kPer + k;
            while (mk) {
                const uint32_t j = (uint32_t)__ffsll((unsigned long long)mk) - 1u;
                mk &= mk - 1ull;
                if (o < cap) {
                    const uint64_t *src = reinterpret_cast<const uint64_t *>(staged + (uint64_t)q * 64u + j);
                    uint64_t w[6];
#pragma unroll
                    for (int i = 0; i < 6; ++i)
                        w[i] = COHERENT ? __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : src[i];
                    uint64_t *dst = reinterpret_cast<uint64_t *>(rows + o);
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        if (rows_coherent) __hip_atomic_store(dst + i, w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        else dst[i] = w[i];
                    }
                    if (lds_cols && o < n_lds) lds_cols[o] = (uint32_t)(w[1] & 0xFFFFFFFFull);  // jl_variant.col
                }
                ++o;
            }
        }
        __syncthreads();
        if (tid == 0) *s_running += total;
        __syncthreads();
    }
    return *s_running;
}
