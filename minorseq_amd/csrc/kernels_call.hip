// kernels_call.hip — reference/majority codon, error model, Fisher's exact x Bonferroni, variant table
// (SURVEY §8 a4-a7).  Behaviour: doc/JULIET.md:38-42 ("comparing the number of observed mutated codons
// to the number of expected mutations ... Bonferroni-corrected Fisher's Exact test"), :133-134
// (reference codon vs major codon), :342-357 (min/max percentage), :370 (drm-only); docs/SPEC.md §4-7.
//
// The evaluation of one position (one wave, lane = codon) is call_eval.h.  The kernels here run it:
//   call_kernel     from the histograms the pileup launch left in HBM — every run and the stage API (jl_call_async).
//                   (A variant that rides in the epilogue of the pileup launch, the workgroup that counted a codon
//                   testing it from the histogram still in LDS, is built only with -DJL_FUSED_CALL: measured slower,
//                   DESIGN.md "Tried and not kept".)
//   compact_kernel  one workgroup per window: the called rows in (gene, codon, codon index) order into the
//                   fixed-stride table (SPEC §6); optionally the distinct variant columns for the multi-word phasing
//                   pipeline (phase_plan.h) and, for runs without phasing, the result block and the completion word
// The point probability uses the saddle-point (Loader) form of the binomial — both table rows sum to the coverage,
// so the hypergeometric is a ratio of three Binomial(.,1/2) masses — which keeps ~1e-14 relative accuracy at 1e7
// coverage where a plain lgamma difference loses 7 digits; the tail is at most `expected`+1 terms of a ratio recurrence.
#include <string.h>

#include "call_eval.h"
#include "jl_internal.h"
#include "phase_plan.h"
#include "result_pack.h"

namespace {

__device__ __forceinline__ void call_body(const jl_win_call &w)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    if (blockIdx.x == 0 && tid == 0 && w.meta) {   // counters of the phasing launch that follows on the stream
        w.meta->n_occupied = 0;
        w.meta->overflow = 0;
        jl_phase_summary z = {0, 0, 0, 0, 0, 0, 0, 0};
        w.meta->summary = z;
    }
    const uint32_t p = blockIdx.x * 4u + wid;
    if (p >= w.A.P) return;  // wave-uniform
    const uint32_t col = w.pos_col[p];
    const uint32_t h = w.hist[(uint64_t)col * 64u + lane];
    jl_call_position<false>(w.A, p, col, h, w.pos_refcfg[p], w.pos_gene[p], w.pos_codon[p], w.drm, w.called, w.staged);
}

__global__ __launch_bounds__(256) void call_kernel(jl_win_call w) { call_body(w); }

// one launch for several windows: blockIdx.z = window
__global__ __launch_bounds__(256) void call_group_kernel(jl_call_group_args args)
{
    const jl_win_call &w = args.w[blockIdx.z];
    if (blockIdx.x >= w.n_blocks) return;
    call_body(w);
}

__device__ __forceinline__ void compact_body(const jl_win_compact &w)
{
    __shared__ uint32_t s_scan[4];
    __shared__ uint32_t s_running;
    constexpr uint32_t kPlanCols = 1024;   // columns of the first rows kept in LDS for the plan
    __shared__ uint32_t s_vcol[kPlanCols];
    const uint32_t n = jl_compact_rows_block<false>(w.P, w.called, w.staged, w.rows, w.cap, s_scan, &s_running, s_vcol,
                                                    kPlanCols, false);
    if (threadIdx.x == 0) w.n_rows[0] = n;
    const uint32_t nv = n < w.cap ? n : w.cap;
    if (w.plan) {
        __syncthreads();
        jl_phase_plan_block(w.rows, nv, w.n_cols, w.varcol, w.vpcols, w.col2pos, w.kwords_cap, w.fast_only, w.meta,
                            nv <= kPlanCols ? s_vcol : nullptr);
    }
    if (w.pack) {   // a run without phasing ends here: the table into the result block (device copy + pinned mirror)
        __syncthreads();
        jl_pack *pk = w.pk + (__hip_atomic_load(w.seq_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u);
        jl_result_pack_block(w.rows, n, nullptr, 0u, nullptr, nullptr, nullptr, nullptr, nullptr, 0u, 0u, pk, w.mirror);
        if (w.xhead) {   // bound exchange
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            jl_result_head_copy(pk, w.xhead);
            if (threadIdx.x == 0) __threadfence_system();
        }
        if (w.seq_host) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) jl_signal_done(w.seq_dev, w.seq_host);
        }
    }
}

__global__ __launch_bounds__(256) void compact_kernel(jl_win_compact w) { compact_body(w); }
__global__ __launch_bounds__(256) void compact_group_kernel(jl_compact_group_args args) { compact_body(args.w[blockIdx.x]); }

__global__ __launch_bounds__(256) void fisher_eval_kernel(uint32_t n, const uint32_t *__restrict__ a,
                                                           const uint32_t *__restrict__ c,
                                                           const uint32_t *__restrict__ cov, int tail,
                                                           double *__restrict__ p, double *__restrict__ lp)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    double l;
    p[i] = tail ? jl_fisher_two_sided_equal_rows(a[i], c[i], cov[i], &l) : jl_fisher_greater_equal_rows(a[i], c[i], cov[i], &l);
    lp[i] = l;
}

}  // namespace

void jl_launch_fisher_eval(jl_ctx *ctx, uint32_t n, const uint32_t *a, const uint32_t *c, const uint32_t *cov, int tail,
                           double *p, double *lp)
{
    hipLaunchKernelGGL(fisher_eval_kernel, dim3((n + 255u) / 256u), dim3(256), 0, ctx->stream, n, a, c, cov, tail, p, lp);
}

void jl_fill_call_args(jl_ctx *ctx, const jl_params *prm, double n_tests, jl_call_args *A)
{
    memset(A, 0, sizeof *A);
    A->alpha = prm->alpha;
    A->n_tests = n_tests;
    A->match = prm->err.match;
    A->substitution = prm->err.substitution;
    A->min_perc = prm->min_perc;
    A->max_perc = prm->max_perc;
    A->expected_round = prm->expected_round;
    A->P = ctx->P;
    A->tail = prm->tail;
}

void jl_fill_win_call(jl_ctx *ctx, const jl_params *prm, double n_tests, bool use_drm, bool with_meta, jl_win_call *w)
{
    memset(w, 0, sizeof *w);
    jl_fill_call_args(ctx, prm, n_tests, &w->A);
    w->pos_gene = ctx->d_pos_gene; w->pos_codon = ctx->d_pos_codon; w->pos_col = ctx->d_pos_col;
    w->pos_refcfg = ctx->d_pos_refcfg;
    w->hist = ctx->d_hist;
    w->drm = use_drm ? ctx->d_drm : nullptr;
    w->called = ctx->d_called;
    w->staged = ctx->d_staged;
    w->meta = with_meta ? ctx->d_meta : nullptr;
    w->n_blocks = ctx->P ? (ctx->P + 3u) / 4u : 1u;   // at least one block: it zeroes the run counters
}

// `plan`: distinct variant columns for the multi-word pipeline / a stage-API phase; `pack`: result block of a run
// without phasing (`signal`: and its completion word)
void jl_fill_win_compact(jl_ctx *ctx, bool plan, bool pack, bool signal, jl_win_compact *w)
{
    memset(w, 0, sizeof *w);
    w->P = ctx->P; w->cap = JL_VARIANT_CAP; w->n_cols = ctx->n_cols; w->kwords_cap = ctx->keys_words;
    w->called = ctx->d_called; w->staged = ctx->d_staged; w->rows = ctx->d_variants; w->n_rows = ctx->d_nvar;
    w->varcol = ctx->d_varcol; w->vpcols = ctx->d_vpcols; w->col2pos = ctx->d_col2pos; w->meta = ctx->d_meta;
    w->plan = plan ? 1u : 0u; w->fast_only = ctx->phase_generic ? 0u : (ctx->phase_two ? 2u : 1u);
    w->pack = pack ? 1u : 0u;
    w->pk = ctx->d_pack; w->mirror = ctx->pack_mirror;
    w->seq_dev = ctx->d_sync; w->seq_host = signal ? ctx->h_seq : nullptr;
}

void jl_launch_call(jl_ctx *ctx, hipStream_t st, const jl_params *prm, double n_tests, bool use_drm, bool with_meta)
{
    jl_win_call w;
    jl_fill_win_call(ctx, prm, n_tests, use_drm, with_meta, &w);
    hipLaunchKernelGGL(call_kernel, dim3(w.n_blocks), dim3(256), 0, st, w);
}

void jl_launch_compact(jl_ctx *ctx, hipStream_t st, bool plan, bool pack, bool signal)
{
    jl_win_compact w;
    jl_fill_win_compact(ctx, plan, pack, signal, &w);
    hipLaunchKernelGGL(compact_kernel, dim3(1), dim3(256), 0, st, w);
}

void jl_launch_call_group(const jl_win_call *h_wins, uint32_t n_win, uint32_t max_blocks, hipStream_t st)
{
    jl_call_group_args args;
    memset(&args, 0, sizeof args);
    memcpy(args.w, h_wins, sizeof(jl_win_call) * (n_win < JL_GROUP_MAX ? n_win : JL_GROUP_MAX));
    hipLaunchKernelGGL(call_group_kernel, dim3(max_blocks, 1, n_win), dim3(256), 0, st, args);
}

void jl_launch_compact_group(const jl_win_compact *h_wins, uint32_t n_win, hipStream_t st)
{
    jl_compact_group_args args;
    memset(&args, 0, sizeof args);
    memcpy(args.w, h_wins, sizeof(jl_win_compact) * (n_win < JL_GROUP_MAX ? n_win : JL_GROUP_MAX));
    hipLaunchKernelGGL(compact_group_kernel, dim3(n_win), dim3(256), 0, st, args);
}
