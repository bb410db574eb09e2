// kernels_call.hip — reference/majority codon, error model, Fisher's exact x Bonferroni, variant table
// (SURVEY §8 a4-a7).  Behaviour: doc/JULIET.md:38-42 ("comparing the number of observed mutated codons
// to the number of expected mutations ... Bonferroni-corrected Fisher's Exact test"), :133-134
// (reference codon vs major codon), :342-357 (min/max percentage), :370 (drm-only); docs/SPEC.md §4-7.
//
// One wave per codon position, one lane per codon (64 codons = one wavefront): coverage and the
// majority codon are wave reductions, every observed non-reference codon is tested by its own lane in
// FP64.  The point probability uses the saddle-point (Loader) form of the binomial — both table rows sum
// to the coverage, so the hypergeometric is a ratio of three Binomial(.,1/2) masses — which keeps ~1e-14
// relative accuracy at 1e7 coverage where a plain lgamma difference loses 7 digits; the tail is at most
// `expected`+1 terms of a ratio recurrence.
#include "jl_internal.h"
#include "jl_fisher.h"
#include "phase_plan.h"

namespace {

__device__ __forceinline__ uint32_t wave_sum_all(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ uint64_t wave_max_all(uint64_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint64_t u = __shfl_xor(v, o, 64);
        v = u > v ? u : v;
    }
    return v;
}

struct call_args {
    double alpha, n_tests, match, substitution, min_perc, max_perc;
    int32_t expected_round;
    uint32_t P;
};

__global__ __launch_bounds__(256) void fisher_kernel(call_args A, const uint32_t *__restrict__ pos_col,
                                                      const uint8_t *__restrict__ pos_refcfg,
                                                      const uint32_t *__restrict__ hist,
                                                      const uint64_t *__restrict__ drm,
                                                      uint64_t *__restrict__ called, double *__restrict__ cand_p,
                                                      double *__restrict__ cand_lp, uint32_t *__restrict__ cand_e,
                                                      uint32_t *__restrict__ pos_cov, uint8_t *__restrict__ pos_ref)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t p = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (p >= A.P) return;
    const uint32_t h = hist[(uint64_t)pos_col[p] * 64u + lane];
    const uint32_t cov = wave_sum_all(h);
    uint32_t ref = pos_refcfg[p];
    if (ref == JL_REF_MAJORITY) {
        // argmax, lowest codon index on ties (SPEC §4)
        const uint64_t key = ((uint64_t)h << 8) | (uint64_t)(63u - lane);
        const uint64_t best = wave_max_all(key);
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
        // An observed count at or below the expected one has p >= 1/2 (the null is symmetric about K/2
        // because both rows sum to the coverage), so it cannot be called once min(1, n_tests/2) >= alpha;
        // uncalled codons are never reported, so their p-value is not needed.
        const double floor_adj = 0.5 * A.n_tests < 1.0 ? 0.5 * A.n_tests : 1.0;
        if (h > e || !(floor_adj >= A.alpha)) {
            const double pv = jl_fisher_greater_equal_rows(h, e, cov, &lp);
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
        called[p] = mask;
        pos_cov[p] = cov;
        pos_ref[p] = (uint8_t)ref;
    }
    if (is_called) {
        const uint64_t o = (uint64_t)p * 64u + lane;
        cand_p[o] = p_adj;
        cand_lp[o] = lp;
        cand_e[o] = e;
    }
}

// Ordered compaction of the called (position, codon) pairs into the fixed-stride variant table:
// ascending (gene, k, codon) because positions are laid out in that order (SPEC §6).
__global__ __launch_bounds__(1024) void compact_kernel(uint32_t P, const uint64_t *__restrict__ called,
                                                        const uint32_t *__restrict__ pos_gene,
                                                        const uint32_t *__restrict__ pos_codon,
                                                        const uint32_t *__restrict__ pos_col,
                                                        const uint32_t *__restrict__ hist,
                                                        const double *__restrict__ cand_p,
                                                        const double *__restrict__ cand_lp,
                                                        const uint32_t *__restrict__ cand_e,
                                                        const uint32_t *__restrict__ pos_cov,
                                                        const uint8_t *__restrict__ pos_ref,
                                                        jl_variant *__restrict__ rows, uint32_t cap,
                                                        uint32_t *__restrict__ n_rows, uint32_t n_cols,
                                                        uint8_t *__restrict__ varcol, uint32_t *__restrict__ vpcols,
                                                        uint32_t *__restrict__ col2pos, uint32_t kwords_cap,
                                                        uint32_t fast_only, jl_phase_meta *__restrict__ meta)
{
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_running;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    if (tid == 0) s_running = 0;
    __syncthreads();
    for (uint32_t base = 0; base < P; base += 1024u) {
        const uint32_t p = base + tid;
        uint64_t m = p < P ? called[p] : 0ull;
        const uint32_t c = (uint32_t)__popcll(m);
        // inclusive wave scan
        uint32_t inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t u = __shfl_up(inc, o, 64);
            if ((int)lane >= o) inc += u;
        }
        if (lane == 63) s_wave[wid] = inc;
        __syncthreads();
        uint32_t wave_off = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const uint32_t v = s_wave[w];
            if (w < (int)wid) wave_off += v;
            total += v;
        }
        uint32_t o = s_running + wave_off + inc - c;
        while (m) {
            const uint32_t j = (uint32_t)__ffsll((unsigned long long)m) - 1u;
            m &= m - 1ull;
            if (o < cap) {
                const uint64_t q = (uint64_t)p * 64u + j;
                jl_variant v;
                v.gene = pos_gene[p];
                v.codon_pos = pos_codon[p];
                v.col = pos_col[p];
                v.ref_codon = pos_ref[p];
                v.codon = (uint8_t)j;
                v.flags = 0;
                v.count = hist[(uint64_t)pos_col[p] * 64u + j];
                v.coverage = pos_cov[p];
                v.expected = cand_e[q];
                v.pad_ = 0;
                v.p_value = cand_p[q];
                v.log_p = cand_lp[q];
                rows[o] = v;
            }
            ++o;
        }
        __syncthreads();
        if (tid == 0) s_running += total;
        __syncthreads();
    }
    if (tid == 0) n_rows[0] = s_running;
    if (meta) {  // phasing follows: distinct variant columns now, saving a dependent launch
        __syncthreads();
        const uint32_t nv = s_running < cap ? s_running : cap;
        __syncthreads();
        jl_phase_plan_block(rows, nv, n_cols, varcol, vpcols, col2pos, kwords_cap, fast_only, meta);
    }
}

__global__ __launch_bounds__(256) void fisher_eval_kernel(uint32_t n, const uint32_t *__restrict__ a,
                                                           const uint32_t *__restrict__ c,
                                                           const uint32_t *__restrict__ cov, double *__restrict__ p,
                                                           double *__restrict__ lp)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    double l;
    p[i] = jl_fisher_greater_equal_rows(a[i], c[i], cov[i], &l);
    lp[i] = l;
}

}  // namespace

void jl_launch_fisher_eval(jl_ctx *ctx, uint32_t n, const uint32_t *a, const uint32_t *c, const uint32_t *cov,
                           double *p, double *lp)
{
    hipLaunchKernelGGL(fisher_eval_kernel, dim3((n + 255u) / 256u), dim3(256), 0, ctx->stream, n, a, c, cov, p, lp);
}

void jl_launch_call(jl_ctx *ctx, const jl_params *prm, double n_tests, bool use_drm, bool with_plan)
{
    call_args A;
    A.alpha = prm->alpha;
    A.n_tests = n_tests;
    A.match = prm->err.match;
    A.substitution = prm->err.substitution;
    A.min_perc = prm->min_perc;
    A.max_perc = prm->max_perc;
    A.expected_round = prm->expected_round;
    A.P = ctx->P;
    if (ctx->P) {
        hipLaunchKernelGGL(fisher_kernel, dim3((ctx->P + 3) / 4), dim3(256), 0, ctx->stream, A, ctx->d_pos_col,
                           ctx->d_pos_refcfg, ctx->d_hist, use_drm ? ctx->d_drm : nullptr, ctx->d_called,
                           ctx->d_cand_p, ctx->d_cand_lp, ctx->d_cand_e, ctx->d_pos_cov, ctx->d_pos_ref);
    }
    hipLaunchKernelGGL(compact_kernel, dim3(1), dim3(1024), 0, ctx->stream, ctx->P, ctx->d_called, ctx->d_pos_gene,
                       ctx->d_pos_codon, ctx->d_pos_col, ctx->d_hist, ctx->d_cand_p, ctx->d_cand_lp, ctx->d_cand_e,
                       ctx->d_pos_cov, ctx->d_pos_ref, ctx->d_variants, JL_VARIANT_CAP, ctx->d_nvar, ctx->n_cols,
                       ctx->d_varcol, ctx->d_vpcols, ctx->d_col2pos, ctx->keys_words, ctx->phase_generic ? 0u : 1u,
                       with_plan ? ctx->d_meta : nullptr);
}
