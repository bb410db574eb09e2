// kernels_call.hip — reference/majority codon, error model, Fisher's exact x Bonferroni, variant table
// (SURVEY §8 a4-a7).  Behaviour: doc/JULIET.md:38-42 ("comparing the number of observed mutated codons
// to the number of expected mutations ... Bonferroni-corrected Fisher's Exact test"), :133-134
// (reference codon vs major codon), :342-357 (min/max percentage), :370 (drm-only); docs/SPEC.md §4-7.
//
// One launch (call_kernel).  One wave per codon position, one lane per codon (64 codons = one wavefront): coverage and the
// majority codon are wave reductions, every observed non-reference codon is tested by its own lane in
// FP64.  The point probability uses the saddle-point (Loader) form of the binomial — both table rows sum
// to the coverage, so the hypergeometric is a ratio of three Binomial(.,1/2) masses — which keeps ~1e-14
// relative accuracy at 1e7 coverage where a plain lgamma difference loses 7 digits; the tail is at most
// `expected`+1 terms of a ratio recurrence.
#include <string.h>

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

typedef jl_call_args call_args;

// One launch for the whole call stage.  Every wave evaluates one codon position and its called lanes write their
// finished variant rows into a staging slot [p][codon]; the block that arrives last at the launch's counter then
// compacts the staged rows into the fixed-stride table in (gene, k, codon) order (positions are laid out in that
// order, SPEC §6) and, when phasing follows, derives the distinct variant columns (phase_plan.h).
// Hand-off between workgroups (XCD L2s are not coherent): the few bytes a wave hands over (its call mask, the rows
// of its called codons) are stored write-through (agent-scope relaxed stores = `sc1`), so no release fence is
// needed — an L2 write-back per block serialises in the L2 and cost more than the launch it replaces; every wave
// drains its stores -> block barrier -> one lane: the arrival add; the last arriver reads every handed-over byte
// with agent-scope relaxed loads (`sc1`, past its L1), which takes the place of an acquire fence (an L1
// invalidate costs ~1.7 us).  The counter is zero before the first launch and the last arriver leaves it zero.
__device__ __forceinline__ void call_body(const jl_win_call &w)
{
    const call_args A = w.A;
    const uint32_t *pos_gene = w.pos_gene, *pos_codon = w.pos_codon, *pos_col = w.pos_col;
    const uint8_t *pos_refcfg = w.pos_refcfg;
    const uint32_t *hist = w.hist;
    const uint64_t *drm = w.drm;
    uint64_t *called = w.called;
    jl_variant *staged = w.staged, *rows = w.rows;
    const uint32_t cap = w.cap, n_cols = w.n_cols, kwords_cap = w.kwords_cap, fast_only = w.fast_only;
    uint32_t *n_rows = w.n_rows, *vpcols = w.vpcols, *col2pos = w.col2pos, *arrive = w.arrive;
    uint8_t *varcol = w.varcol;
    jl_phase_meta *meta = w.meta;
    constexpr uint32_t kPlanCols = 1024;   // columns of the first rows kept in LDS for the plan
    __shared__ uint32_t s_scan[16];
    __shared__ uint32_t s_running, s_last;
    __shared__ uint32_t s_vcol[kPlanCols];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    const uint32_t p = blockIdx.x * 4u + wid;
    if (p < A.P) {  // wave-uniform
        const uint32_t col = pos_col[p];
        const uint32_t h = hist[(uint64_t)col * 64u + lane];
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
                bool skipped;
                const double pv = jl_fisher_greater_equal_rows_or_skip(h, e, cov, A.n_tests, A.alpha, &lp, &skipped);
                p_adj = pv * A.n_tests;
                if (p_adj > 1.0) p_adj = 1.0;
                is_called = !skipped && p_adj < A.alpha;
            }
            const double perc = 100.0 * (double)h / (double)cov;
            if (A.min_perc >= 0.0 && !(perc > A.min_perc)) is_called = false;
            if (A.max_perc >= 0.0 && !(perc < A.max_perc)) is_called = false;
            if (drm && !((drm[p] >> lane) & 1ull)) is_called = false;
        }
        const uint64_t mask = __ballot(is_called);
        if (lane == 0) __hip_atomic_store(&called[p], mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (is_called) {
            jl_variant v;
            v.gene = pos_gene[p];
            v.codon_pos = pos_codon[p];
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
            // six 8-byte write-through stores
            uint64_t w[6];
            memcpy(w, &v, sizeof v);
            uint64_t *dst = reinterpret_cast<uint64_t *>(staged + (uint64_t)p * 64u + lane);
#pragma unroll
            for (int k = 0; k < 6; ++k) __hip_atomic_store(dst + k, w[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // ---- hand-off: the last block to arrive owns everything the others wrote
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const uint32_t prev = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t last = prev == w.n_blocks - 1u;
        if (last) __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = last;
        s_running = 0;
    }
    __syncthreads();
    if (!s_last) return;

    // ---- ordered compaction.  A latency chain on one CU: every dependent global round trip costs ~1 us (several
    // under a streaming neighbour), so a pass loads the call masks of 8 x 256 positions at once, ranks them with
    // one block scan, and copies all called rows with independent loads.
    const uint32_t P = A.P;
    constexpr uint32_t kPer = 8;  // consecutive positions per thread and pass
    for (uint32_t base = 0; base < P; base += 256u * kPer) {
        uint64_t m[kPer];
        uint32_t c = 0;
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k) {
            const uint32_t q = base + tid * kPer + k;
            m[k] = q < P ? __hip_atomic_load(&called[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
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
        uint32_t o = s_running + wave_off + inc - c;
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k) {
            uint64_t mk = m[k];
            const uint32_t q = base + tid * kPer + k;
            while (mk) {
                const uint32_t j = (uint32_t)__ffsll((unsigned long long)mk) - 1u;
                mk &= mk - 1ull;
                if (o < cap) {
                    // 48-byte rows: six write-through-coherent loads, three 16-byte stores
                    uint64_t *src = reinterpret_cast<uint64_t *>(staged + (uint64_t)q * 64u + j);
                    uint64_t w[6];
#pragma unroll
                    for (int i = 0; i < 6; ++i) w[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(rows + o);
#pragma unroll
                    for (int i = 0; i < 3; ++i) { ulonglong2 v2; v2.x = w[2 * i]; v2.y = w[2 * i + 1]; dst[i] = v2; }
                    if (o < kPlanCols) s_vcol[o] = (uint32_t)(w[1] & 0xFFFFFFFFull);  // jl_variant.col
                }
                ++o;
            }
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
        jl_phase_plan_block(rows, nv, n_cols, varcol, vpcols, col2pos, kwords_cap, fast_only, meta,
                            nv <= kPlanCols ? s_vcol : nullptr);
    }
}

__global__ __launch_bounds__(256) void call_kernel(jl_win_call w) { call_body(w); }

// one launch for several windows: blockIdx.z = window, argument blocks in device memory
__global__ __launch_bounds__(256) void call_group_kernel(jl_call_group_args args)
{
    const jl_win_call &w = args.w[blockIdx.z];
    if (blockIdx.x >= w.n_blocks) return;
    call_body(w);
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

void jl_fill_win_call(jl_ctx *ctx, const jl_params *prm, double n_tests, bool use_drm, bool with_plan, jl_win_call *w)
{
    memset(w, 0, sizeof *w);
    w->A.alpha = prm->alpha;
    w->A.n_tests = n_tests;
    w->A.match = prm->err.match;
    w->A.substitution = prm->err.substitution;
    w->A.min_perc = prm->min_perc;
    w->A.max_perc = prm->max_perc;
    w->A.expected_round = prm->expected_round;
    w->A.P = ctx->P;
    w->pos_gene = ctx->d_pos_gene; w->pos_codon = ctx->d_pos_codon; w->pos_col = ctx->d_pos_col;
    w->pos_refcfg = ctx->d_pos_refcfg;
    w->hist = ctx->d_hist;
    w->drm = use_drm ? ctx->d_drm : nullptr;
    w->called = ctx->d_called;
    w->staged = ctx->d_staged; w->rows = ctx->d_variants;
    w->cap = JL_VARIANT_CAP; w->n_cols = ctx->n_cols;
    w->n_rows = ctx->d_nvar;
    w->varcol = ctx->d_varcol; w->vpcols = ctx->d_vpcols; w->col2pos = ctx->d_col2pos;
    w->kwords_cap = ctx->keys_words; w->fast_only = ctx->phase_generic ? 0u : 1u;
    w->meta = with_plan ? ctx->d_meta : nullptr;
    w->arrive = ctx->d_sync + 1;
    // at least one block even without positions: the last (only) block still writes the row count and the plan
    w->n_blocks = ctx->P ? (ctx->P + 3u) / 4u : 1u;
}

void jl_launch_call(jl_ctx *ctx, const jl_params *prm, double n_tests, bool use_drm, bool with_plan)
{
    jl_win_call w;
    jl_fill_win_call(ctx, prm, n_tests, use_drm, with_plan, &w);
    hipLaunchKernelGGL(call_kernel, dim3(w.n_blocks), dim3(256), 0, ctx->stream, w);
}

void jl_launch_call_group(const jl_win_call *h_wins, uint32_t n_win, uint32_t max_blocks, hipStream_t st)
{
    jl_call_group_args args;
    memset(&args, 0, sizeof args);
    memcpy(args.w, h_wins, sizeof(jl_win_call) * (n_win < JL_GROUP_MAX ? n_win : JL_GROUP_MAX));
    hipLaunchKernelGGL(call_group_kernel, dim3(max_blocks, 1, n_win), dim3(256), 0, st, args);
}
