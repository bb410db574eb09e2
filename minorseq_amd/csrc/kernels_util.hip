// kernels_util.hip — device-side producers of the resident column-packed matrix:
//   synth_kernel      synthetic aligned CCS reads (jl_synth.h), one dword (8 reads) of one column per step
//   pack_rows_kernel  by-row uint8 codes -> column-packed nibbles (the first step of SURVEY §8 f1)
#include <algorithm>

#include "jl_internal.h"

namespace {

constexpr int kSynthColsPerBlock = 64;

__global__ __launch_bounds__(256) void synth_kernel(jl_synth_plan pl, const uint8_t *__restrict__ ref,
                                                     uint8_t *__restrict__ msa, uint64_t col_stride,
                                                     uint64_t n_reads)
{
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;  // dword index in a column
    if (t * 4u >= col_stride) return;
    uint32_t hap[8], st[8], en[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const uint64_t i = t * 8u + r;
        if (i < n_reads) jl_synth_read(&pl, i, &hap[r], &st[r], &en[r]);
        else { hap[r] = 0; st[r] = 1; en[r] = 0; }  // empty range: padding reads are uncovered
    }
    const uint32_t c0 = blockIdx.y * kSynthColsPerBlock;
    const uint32_t c1 = min(pl.n_cols, c0 + kSynthColsPerBlock);
    for (uint32_t c = c0; c < c1; ++c) {
        const uint32_t rb = ref[c];
        uint32_t w = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) w |= jl_synth_cell(&pl, t * 8u + r, c, hap[r], st[r], en[r], rb) << (4 * r);
        *reinterpret_cast<uint32_t *>(msa + (uint64_t)c * col_stride + t * 4u) = w;
    }
}

// thread (x = column, y = dword of the column): gathers 8 reads of one column
__global__ __launch_bounds__(256) void pack_rows_kernel(const uint8_t *__restrict__ rows, uint64_t n_reads,
                                                         uint32_t n_cols, uint8_t *__restrict__ msa,
                                                         uint64_t col_stride)
{
    const uint32_t c = blockIdx.x * 64u + (threadIdx.x & 63u);
    const uint64_t t = (uint64_t)blockIdx.y * 4u + (threadIdx.x >> 6);
    if (c >= n_cols || t * 4u >= col_stride) return;
    uint32_t w = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const uint64_t i = t * 8u + r;
        const uint32_t s = i < n_reads ? rows[i * n_cols + c] : 6u;
        w |= (s & 7u) << (4 * r);
    }
    *reinterpret_cast<uint32_t *>(msa + (uint64_t)c * col_stride + t * 4u) = w;
}


// ---------------------------------------------------------------------------------------- record ingest
// Aligned records -> by-row symbols, one wave per read: doc/JULIET.md:26-27 (insertions dropped, deletions '-'),
// :53 (PacBio cigars = X I D S H N; M rejected on the host), :256-259 (filtered base = N).
// Lanes walk REFERENCE positions; the operation covering a position is found by binary search in the read's
// scanned cigar (reference and query offsets per op, kept in LDS).
constexpr int kIngestMaxOps = 1024;  // ops staged per pass; longer cigars are processed in several passes

__global__ __launch_bounds__(256) void ingest_kernel(uint64_t n_reads, uint32_t n_cols, uint32_t win_begin,
                                                      const int32_t *__restrict__ pos,
                                                      const uint32_t *__restrict__ cigar,
                                                      const uint64_t *__restrict__ cig_off,
                                                      const uint8_t *__restrict__ seq4,
                                                      const uint64_t *__restrict__ seq_off,
                                                      const uint8_t *__restrict__ qual,
                                                      const uint64_t *__restrict__ qual_off, uint32_t min_qv,
                                                      uint8_t *__restrict__ rows)
{
    __shared__ uint32_t s_rend[4][kIngestMaxOps];  // reference offset AFTER op k (relative to the pass start)
    __shared__ uint32_t s_qbeg[4][kIngestMaxOps];  // query offset BEFORE op k
    __shared__ uint8_t s_op[4][kIngestMaxOps];
    const uint32_t wid = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint64_t r = (uint64_t)blockIdx.x * 4u + wid;
    if (r >= n_reads) return;
    const uint64_t c0 = cig_off[r], c1 = cig_off[r + 1];
    const uint8_t *sq = seq4 + seq_off[r];
    const uint8_t *ql = qual ? qual + qual_off[r] : nullptr;
    uint8_t *row = rows + r * (uint64_t)n_cols;
    int64_t ref_cur = pos[r];
    uint32_t q_cur = 0;
    for (uint64_t base = c0; base < c1; base += kIngestMaxOps) {
        const uint32_t nops = (uint32_t)min((uint64_t)kIngestMaxOps, c1 - base);
        // scan the ops of this pass (64 at a time) into LDS
        uint32_t racc = 0, qacc = q_cur;
        for (uint32_t k0 = 0; k0 < nops; k0 += 64u) {
            const uint32_t k = k0 + lane;
            uint32_t op = 15u, len = 0;
            if (k < nops) { const uint32_t c = cigar[base + k]; op = c & 15u; len = c >> 4; }
            const uint32_t rl = (op == 2u || op == 3u || op == 7u || op == 8u) ? len : 0u;   // D N = X consume reference
            const uint32_t qn = (op == 1u || op == 4u || op == 7u || op == 8u) ? len : 0u;   // I S = X consume query
            uint32_t ri = rl, qi = qn;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t a = __shfl_up(ri, o, 64), b = __shfl_up(qi, o, 64);
                if ((int)lane >= o) { ri += a; qi += b; }
            }
            if (k < nops) {
                s_rend[wid][k] = racc + ri;
                s_qbeg[wid][k] = qacc + qi - qn;
                s_op[wid][k] = (uint8_t)op;
            }
            racc += __shfl(ri, 63, 64);
            qacc += __shfl(qi, 63, 64);
        }
        __builtin_amdgcn_wave_barrier();
        const uint32_t rtot = racc;
        // reference positions of this pass, 64 at a time
        for (uint32_t p0 = 0; p0 < rtot; p0 += 64u) {
            const uint32_t p = p0 + lane;
            const int64_t col = ref_cur + p - (int64_t)win_begin;
            if (p < rtot && col >= 0 && col < (int64_t)n_cols) {
                uint32_t lo = 0, hi = nops - 1u;  // first op with rend > p
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (s_rend[wid][mid] > p) hi = mid; else lo = mid + 1u;
                }
                const uint32_t op = s_op[wid][lo];
                const uint32_t rbeg = lo ? s_rend[wid][lo - 1u] : 0u;
                uint8_t sym;
                if (op == 2u) sym = JL_SYM_GAP;
                else if (op == 3u) sym = JL_SYM_NONE;
                else {
                    const uint32_t q = s_qbeg[wid][lo] + (p - rbeg);
                    const uint8_t b16 = (q & 1u) ? (sq[q >> 1] & 15u) : (sq[q >> 1] >> 4);
                    // BAM nibble codes: A=1 C=2 G=4 T=8; anything else is an ambiguous base
                    sym = b16 == 1u ? 0 : b16 == 2u ? 1 : b16 == 4u ? 2 : b16 == 8u ? 3 : (uint8_t)JL_SYM_MASK;
                    if (ql && min_qv) { const uint8_t qv = ql[q]; if (qv != 0xFFu && qv < min_qv) sym = JL_SYM_MASK; }
                }
                row[col] = sym;
            }
        }
        ref_cur += rtot;
        q_cur = qacc;
        __builtin_amdgcn_wave_barrier();
    }
}

// Per-column consensus from the pileup (doc/FUSE.md:17-20, the part that needs no insertion tracking):
// majority among A C G T -; a column whose majority is '-' is marked removed (4); no covering read => 5.
__global__ __launch_bounds__(256) void consensus_kernel(const uint32_t *__restrict__ counts, uint32_t n_cols,
                                                         uint8_t *__restrict__ out)
{
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c >= n_cols) return;
    const uint32_t *k = counts + (uint64_t)c * 6u;
    uint32_t best = 0, bv = k[0];
#pragma unroll
    for (uint32_t s = 1; s < 5; ++s)
        if (k[s] > bv) { bv = k[s]; best = s; }
    out[c] = bv == 0 ? (uint8_t)5 : (uint8_t)best;
}

// any nibble outside 0..6 (code 7 or bit 3 set) is rejected at upload (SPEC §1)
__global__ __launch_bounds__(256) void validate_kernel(const uint8_t *__restrict__ msa, uint64_t n_bytes,
                                                        uint32_t *__restrict__ bad)
{
    uint32_t any = 0;
    for (uint64_t i = ((uint64_t)blockIdx.x * 256u + threadIdx.x) * 16u; i < n_bytes; i += (uint64_t)gridDim.x * 256u * 16u) {
        const uint4 v = *reinterpret_cast<const uint4 *>(msa + i);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) any |= (w[q] & 0x88888888u) | (w[q] & (w[q] >> 1) & (w[q] >> 2) & 0x11111111u);
    }
    if (any) atomicOr(bad, 1u);
}

}  // namespace

// Last node of a run: everything before it on the stream has completed (including the stores the result
// kernels made into pinned host memory), so a sequence word stored behind a system-scope fence tells a
// polling host that the results are there — no hipStreamSynchronize on the hot path.
__global__ void done_kernel(uint32_t *__restrict__ seq_dev, volatile uint32_t *__restrict__ seq_host)
{
    const uint32_t v = *seq_dev + 1u;
    *seq_dev = v;
    __threadfence_system();
    __hip_atomic_store(const_cast<uint32_t *>(seq_host), v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// the same for every window of a group launch: thread k ends window k's run
__global__ void done_group_kernel(const jl_done_ent *__restrict__ ents, uint32_t n)
{
    const uint32_t k = threadIdx.x;
    if (k >= n) return;
    uint32_t *seq_dev = ents[k].seq_dev;
    const uint32_t v = *seq_dev + 1u;
    *seq_dev = v;
    __threadfence_system();
    __hip_atomic_store(const_cast<uint32_t *>(ents[k].seq_host), v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
void jl_launch_done_group(const jl_done_ent *d_ents, uint32_t n, hipStream_t st)
{
    hipLaunchKernelGGL(done_group_kernel, dim3(1), dim3(64), 0, st, d_ents, n);
}

void jl_launch_done(jl_ctx *ctx)
{
    hipLaunchKernelGGL(done_kernel, dim3(1), dim3(1), 0, ctx->stream, ctx->d_sync, ctx->h_seq);
}

// tuning aid (JL_TIMELINE=1): a one-thread node that records the device's constant-rate clock between the stages of
// a run, row = runs completed so far; read back with jl_debug_timeline (tools_tuning/timeline.py)
__global__ void stamp_kernel(const uint32_t *__restrict__ seq_dev, uint64_t *__restrict__ tl, uint32_t slot)
{
    tl[(uint64_t)(*seq_dev % JL_TIMELINE_ROWS) * JL_TIMELINE_SLOTS + slot] = wall_clock64();
}
void jl_launch_stamp(jl_ctx *ctx, uint32_t slot)
{
    if (!ctx->d_timeline) return;
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, ctx->stream, ctx->d_sync, ctx->d_timeline, slot);
}

// tuning probe: what does one more (empty) dependent node cost a pipelined step?
__global__ void noop_kernel(uint32_t *p) { if (p == nullptr && threadIdx.x == 12345u) *p = 0; }
void jl_launch_noop(jl_ctx *ctx) { hipLaunchKernelGGL(noop_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->d_nvar); }

void jl_launch_consensus(jl_ctx *ctx, uint8_t *d_out)
{
    hipLaunchKernelGGL(consensus_kernel, dim3((ctx->n_cols + 255u) / 256u), dim3(256), 0, ctx->stream, ctx->d_counts,
                       ctx->n_cols, d_out);
}

void jl_launch_validate(jl_ctx *ctx, uint32_t *d_flag)
{
    const uint64_t n_bytes = (uint64_t)ctx->col_stride * ctx->n_cols;  // multiple of 128
    uint32_t blocks = (uint32_t)std::min<uint64_t>(2048, (n_bytes / 16 + 255) / 256);
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(validate_kernel, dim3(blocks), dim3(256), 0, ctx->stream, ctx->d_msa, n_bytes, d_flag);
}

void jl_launch_ingest(jl_ctx *ctx, const int32_t *d_pos, const uint32_t *d_cigar, const uint64_t *d_cig_off,
                      const uint8_t *d_seq4, const uint64_t *d_seq_off, const uint8_t *d_qual,
                      const uint64_t *d_qual_off, uint32_t min_qv, uint8_t *d_rows)
{
    const uint32_t blocks = (uint32_t)((ctx->n_reads + 3u) / 4u);
    hipLaunchKernelGGL(ingest_kernel, dim3(blocks), dim3(256), 0, ctx->stream, ctx->n_reads, ctx->n_cols,
                       ctx->win_begin, d_pos, d_cigar, d_cig_off, d_seq4, d_seq_off, d_qual, d_qual_off, min_qv, d_rows);
}

void jl_launch_synth(jl_ctx *ctx, const jl_synth_plan *plan, const uint8_t *d_ref)
{
    const uint32_t n_dwords = (uint32_t)(ctx->col_stride / 4u);
    dim3 grid((n_dwords + 255u) / 256u, (ctx->n_cols + kSynthColsPerBlock - 1) / kSynthColsPerBlock);
    hipLaunchKernelGGL(synth_kernel, grid, dim3(256), 0, ctx->stream, *plan, d_ref, ctx->d_msa, ctx->col_stride,
                       ctx->n_reads);
}

void jl_launch_pack_rows(jl_ctx *ctx, const uint8_t *d_rows)
{
    const uint32_t n_dwords = (uint32_t)(ctx->col_stride / 4u);
    dim3 grid((ctx->n_cols + 63u) / 64u, (n_dwords + 3u) / 4u);
    hipLaunchKernelGGL(pack_rows_kernel, grid, dim3(256), 0, ctx->stream, d_rows, ctx->n_reads, ctx->n_cols,
                       ctx->d_msa, ctx->col_stride);
}
