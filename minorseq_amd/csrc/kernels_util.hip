// kernels_util.hip — device-side producers of the resident column-packed matrix:
//   synth_kernel      synthetic aligned CCS reads (jl_synth.h), one dword (8 reads) of one column per step
//   pack_rows_kernel  by-row uint8 codes -> column-packed nibbles (jl_msa_pack_rows)
//   ingest_cols_kernel  aligned BAM records -> column-packed nibbles
#include <string.h>

#include <algorithm>

#include "jl_internal.h"

namespace {

constexpr int kSynthColsPerBlock = 64;

// `ref`, `msa`: the window's columns [col0, col0 + win_cols) of the pl.n_cols-column reference the plan describes
__global__ __launch_bounds__(256) void synth_kernel(jl_synth_plan pl, const uint8_t *__restrict__ ref,
                                                     uint8_t *__restrict__ msa, uint64_t col_stride,
                                                     uint64_t n_reads, uint32_t col0, uint32_t win_cols)
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
    const uint32_t c1 = min(win_cols, c0 + kSynthColsPerBlock);
    for (uint32_t c = c0; c < c1; ++c) {
        const uint32_t rb = ref[c];
        uint32_t w = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) w |= jl_synth_cell(&pl, t * 8u + r, col0 + c, hap[r], st[r], en[r], rb) << (4 * r);
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
// Aligned records -> column-packed nibbles in ONE pass: doc/JULIET.md:26-27 (insertions dropped, deletions '-'),
// :53 (PacBio cigars = X I D S H N; M rejected on the host), :256-259 (filtered base = N).
// One LANE per read: the lane walks its own cigar while the wave walks the columns of a segment of the window, so at
// every column the wave holds the 64 symbols of 64 consecutive reads — the 32 bytes that are contiguous in the
// column-packed matrix.  Four columns are gathered per lane into one dword, eight such dwords are transposed through a
// 2 KiB LDS tile per wave, and every lane then stores the packed nibbles of 8 reads for 4 columns.  The next cigar word
// and the next eight bases of every lane are loaded ahead of their use.  (Loading a tile's worth of cigar words and
// bases per lane up front was measured and is slower: the kernel is bound by the instructions of the per-lane cursor
// under divergence — some lane of 64 changes its op at nearly every column — not by load latency.)  Reads past n_reads (the padding of a column up
// to its 128-byte stride) and columns outside a read's span are 'not covered'.
constexpr uint32_t kIngestSegAlign = 32;   // columns per transposed tile

__device__ __forceinline__ bool cig_ref(uint32_t op) { return op == 2u || op == 3u || op == 7u || op == 8u; }    // D N = X
__device__ __forceinline__ bool cig_query(uint32_t op) { return op == 1u || op == 4u || op == 7u || op == 8u; }  // I S = X

__global__ __launch_bounds__(256) void ingest_cols_kernel(uint64_t n_reads, uint32_t n_cols, uint32_t win_begin, uint32_t seg_cols,
                                                           const int32_t *__restrict__ pos,
                                                           const uint32_t *__restrict__ cigar,
                                                           const uint64_t *__restrict__ cig_off,
                                                           const uint8_t *__restrict__ seq4,
                                                           const uint64_t *__restrict__ seq_off,
                                                           const uint8_t *__restrict__ qual,
                                                           const uint64_t *__restrict__ qual_off, uint32_t min_qv,
                                                           uint8_t *__restrict__ msa, uint64_t col_stride)
{
    __shared__ uint32_t s_t[4][8][64];   // per wave: 8 groups of 4 columns x 64 reads
    const uint32_t wid = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint64_t wave_r0 = ((uint64_t)blockIdx.x * 4u + wid) * 64u;
    if (wave_r0 * 4u >= col_stride * 8u) return;   // col_stride * 2 reads per column (wave-uniform)
    const uint64_t r = wave_r0 + lane;
    const bool have = r < n_reads;
    const uint32_t cs = blockIdx.y * seg_cols, ce = min(n_cols, cs + seg_cols);

    uint64_t ci = 0, cend = 0;
    const uint32_t *sqw = nullptr;
    uint32_t q_adj = 0, sq_last = 0, ql_last = 0;
    const uint8_t *ql = nullptr;
    int64_t rel = 0;   // reference offset, relative to the read's first base, of column cs
    if (have) {
        ci = cig_off[r];
        cend = cig_off[r + 1];
        const uint64_t so = seq_off[r];
        sqw = reinterpret_cast<const uint32_t *>(seq4 + (so & ~(uint64_t)3));
        q_adj = (uint32_t)(so & 3u) * 2u;
        sq_last = (uint32_t)((seq_off[r + 1] - (so & ~(uint64_t)3) + 3u) >> 2);   // the arrays are padded by 16 bytes
        if (qual && min_qv) {
            ql = qual + qual_off[r];
            const uint64_t nq = qual_off[r + 1] - qual_off[r];
            ql_last = nq ? (uint32_t)(nq - 1) : 0u;
            if (!nq) ql = nullptr;
        }
        rel = (int64_t)win_begin + cs - (int64_t)pos[r];
    }
    // cursor: the op covering reference offsets [r_beg, r_end), which starts at query offset q_beg
    int64_t r_beg = 0, r_end = 0;
    uint32_t q_beg = 0, q_next = 0, op = 15u;
    bool done = !have;
    // ---- skip to the segment: four cigar words per step
    if (have && rel > 0) {
        while (ci < cend && r_end <= rel) {
            uint32_t cw[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) cw[k] = cigar[min(ci + k, cend - 1)];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (ci < cend && r_end <= rel) {
                    const uint32_t o = cw[k] & 15u, len = cw[k] >> 4;
                    op = o; r_beg = r_end; q_beg = q_next;
                    if (cig_ref(o)) r_end += len;
                    if (cig_query(o)) q_next += len;
                    ++ci;
                }
            }
        }
    }
    uint32_t cw_next = (have && ci < cend) ? cigar[ci] : 0u;
    uint32_t seq_idx = 0xFFFFFFFEu, seq_w = 0, seq_wn = 0;   // dword seq_idx of the read's bases, and the one after it

    for (uint32_t c0 = cs; c0 < ce; c0 += kIngestSegAlign) {
#pragma unroll 1
        for (uint32_t g = 0; g < 8u; ++g) {
            uint32_t pk = 0;
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j) {
                const int64_t x = rel + (int64_t)(c0 - cs + 4u * g + j);
                uint32_t sym = JL_SYM_NONE;
                if (!done && x >= 0) {
                    while (x >= r_end) {   // next op (those that consume no reference leave r_end where it is)
                        if (ci >= cend) { done = true; break; }
                        const uint32_t o = cw_next & 15u, len = cw_next >> 4;
                        ++ci;
                        if (ci < cend) cw_next = cigar[ci];
                        op = o; r_beg = r_end; q_beg = q_next;
                        if (cig_ref(o)) r_end += len;
                        if (cig_query(o)) q_next += len;
                    }
                    if (!done) {
                        if (op == 2u) sym = JL_SYM_GAP;
                        else if (op == 3u) sym = JL_SYM_NONE;
                        else {
                            const uint32_t q = q_beg + (uint32_t)(x - r_beg);
                            const uint32_t qa = q + q_adj, idx = qa >> 3;
                            if (idx != seq_idx) {
                                if (idx == seq_idx + 1u) seq_w = seq_wn;
                                else seq_w = sqw[min(idx, sq_last)];
                                seq_wn = sqw[min(idx + 1u, sq_last)];
                                seq_idx = idx;
                            }
                            // BAM: two bases per byte, the first in the high nibble
                            const uint32_t b16 = (seq_w >> (8u * ((qa >> 1) & 3u) + ((qa & 1u) ? 0u : 4u))) & 15u;
                            // A=1 C=2 G=4 T=8 -> 0..3; anything else is an ambiguous base (N)
                            sym = (uint32_t)((0x5555555355525105ull >> (4u * b16)) & 15ull);
                            // a cigar that runs past the read's bases (malformed input) stays inside the read's arrays
                            if (ql) { const uint8_t qv = ql[min(q, ql_last)]; if (qv != 0xFFu && qv < min_qv) sym = JL_SYM_MASK; }
                        }
                    }
                }
                pk |= sym << (8u * j);
            }
            s_t[wid][g][lane] = pk;
        }
        __builtin_amdgcn_wave_barrier();
        // transposed: lane = (group of 4 columns, group of 8 reads)
        const uint32_t tg = lane >> 3, tk = lane & 7u;
        uint32_t w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = s_t[wid][tg][8u * tk + j];
#pragma unroll
        for (uint32_t cc = 0; cc < 4u; ++cc) {
            const uint32_t c = c0 + 4u * tg + cc;
            uint32_t o = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) o |= ((w[j] >> (8u * cc)) & 15u) << (4 * j);
            if (c < ce) *reinterpret_cast<uint32_t *>(msa + (uint64_t)c * col_stride + wave_r0 / 2u + 4u * tk) = o;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Insertions per window column (doc/FUSE.md:19 "Fuse includes in-frame insertions"): they are not part of the MSA
// (doc/JULIET.md:26-27), so they are counted from the records.  One thread per read walks its cigar; an insertion sits
// BEFORE the window column of the next reference base: len_hist[c][min(len, 31)]++, base_counts[c][j][base]++ for the
// inserted bases at offsets j < 30.  Integer atomics commute: bit-exact against the oracle's loops.
__global__ __launch_bounds__(256) void insertions_kernel(uint64_t n_reads, uint32_t n_cols, uint32_t win_begin,
                                                          const int32_t *__restrict__ pos, const uint32_t *__restrict__ cigar,
                                                          const uint64_t *__restrict__ cig_off, const uint8_t *__restrict__ seq4,
                                                          const uint64_t *__restrict__ seq_off, uint32_t *__restrict__ len_hist,
                                                          uint32_t *__restrict__ base_counts)
{
    const uint64_t r = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (r >= n_reads) return;
    int64_t rp = pos[r];
    uint64_t qp = 0;
    const uint8_t *sq = seq4 + seq_off[r];
    const uint64_t n_bases = (seq_off[r + 1] - seq_off[r]) * 2u;
    for (uint64_t k = cig_off[r]; k < cig_off[r + 1]; ++k) {
        const uint32_t op = cigar[k] & 15u, len = cigar[k] >> 4;
        if (op == 1u) {
            const int64_t c = rp - (int64_t)win_begin;
            if (c >= 0 && c < (int64_t)n_cols) {
                atomicAdd(&len_hist[(uint64_t)c * JL_INS_LEN_BINS + (len < 31u ? len : 31u)], 1u);
                for (uint32_t j = 0; j < len && j < JL_INS_MAX_BASES; ++j) {
                    const uint64_t q = qp + j;
                    if (q >= n_bases) break;   // malformed input stays inside the read's bases
                    const uint32_t b16 = (q & 1u) ? (sq[q >> 1] & 15u) : (sq[q >> 1] >> 4);
                    const uint32_t b = (uint32_t)((0x5555555355525105ull >> (4u * b16)) & 15ull);   // A=1 C=2 G=4 T=8 -> 0..3, else 5
                    if (b < 4u) atomicAdd(&base_counts[((uint64_t)c * JL_INS_MAX_BASES + j) * 4u + b], 1u);
                }
            }
            qp += len;
        } else if (op == 4u) {
            qp += len;
        } else if (op == 7u || op == 8u) {
            qp += len;
            rp += len;
        } else if (op == 2u || op == 3u) {
            rp += len;
        }
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

// the heads (header + first 128 rows) of n result blocks, one workgroup each, into one contiguous buffer
__global__ __launch_bounds__(256) void gather_heads_kernel(jl_gather_args a, uint8_t *__restrict__ dst)
{
    static_assert(JL_PACK_HEAD_BYTES % 4 == 0, "heads are copied as dwords");
    const uint32_t *s = reinterpret_cast<const uint32_t *>(a.src[blockIdx.x]);
    uint32_t *d = reinterpret_cast<uint32_t *>(dst + (size_t)blockIdx.x * JL_PACK_HEAD_BYTES);
    for (uint32_t i = threadIdx.x; i < JL_PACK_HEAD_BYTES / 4; i += 256) d[i] = s[i];
}

void jl_launch_gather_heads(const uint8_t *const *srcs, uint32_t n, uint8_t *dst, hipStream_t st)
{
    jl_gather_args a;
    memset(&a, 0, sizeof a);
    for (uint32_t k = 0; k < n && k < JL_GATHER_MAX; ++k) a.src[k] = srcs[k];
    hipLaunchKernelGGL(gather_heads_kernel, dim3(n), dim3(256), 0, st, a, dst);
}

void jl_launch_done(jl_ctx *ctx)
{
    hipLaunchKernelGGL(done_kernel, dim3(1), dim3(1), 0, ctx->stream, ctx->d_sync, ctx->h_seq);
}

#ifdef JL_TUNING
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
#else
void jl_launch_stamp(jl_ctx *, uint32_t) {}
#endif

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
                      const uint64_t *d_qual_off, uint32_t min_qv)
{
    // waves = 64-read groups x column segments; enough segments for some thousands of waves
    const uint64_t waves_x = (ctx->col_stride * 2u + 63u) / 64u;
    const uint32_t max_seg = (ctx->n_cols + kIngestSegAlign - 1u) / kIngestSegAlign;
    uint32_t nseg = (uint32_t)std::min<uint64_t>(max_seg, std::max<uint64_t>(1, (8192u + waves_x - 1u) / waves_x));
    uint32_t seg_cols = (ctx->n_cols + nseg - 1u) / nseg;
    seg_cols = (seg_cols + kIngestSegAlign - 1u) / kIngestSegAlign * kIngestSegAlign;
    nseg = (ctx->n_cols + seg_cols - 1u) / seg_cols;
    dim3 grid((uint32_t)((waves_x + 3u) / 4u), nseg);
    hipLaunchKernelGGL(ingest_cols_kernel, grid, dim3(256), 0, ctx->stream, ctx->n_reads, ctx->n_cols, ctx->win_begin,
                       seg_cols, d_pos, d_cigar, d_cig_off, d_seq4, d_seq_off, d_qual, d_qual_off, min_qv, ctx->d_msa,
                       ctx->col_stride);
}

void jl_launch_insertions(jl_ctx *ctx, const int32_t *d_pos, const uint32_t *d_cigar, const uint64_t *d_cig_off,
                          const uint8_t *d_seq4, const uint64_t *d_seq_off)
{
    hipLaunchKernelGGL(insertions_kernel, dim3((uint32_t)((ctx->n_reads + 255u) / 256u)), dim3(256), 0, ctx->stream, ctx->n_reads,
                       ctx->n_cols, ctx->win_begin, d_pos, d_cigar, d_cig_off, d_seq4, d_seq_off, ctx->d_ins_len, ctx->d_ins_base);
}

void jl_launch_synth(jl_ctx *ctx, const jl_synth_plan *plan, const uint8_t *d_ref, uint32_t col0)
{
    const uint32_t n_dwords = (uint32_t)(ctx->col_stride / 4u);
    dim3 grid((n_dwords + 255u) / 256u, (ctx->n_cols + kSynthColsPerBlock - 1) / kSynthColsPerBlock);
    hipLaunchKernelGGL(synth_kernel, grid, dim3(256), 0, ctx->stream, *plan, d_ref, ctx->d_msa, ctx->col_stride,
                       ctx->n_reads, col0, ctx->n_cols);
}

void jl_launch_pack_rows(jl_ctx *ctx, const uint8_t *d_rows)
{
    const uint32_t n_dwords = (uint32_t)(ctx->col_stride / 4u);
    dim3 grid((ctx->n_cols + 63u) / 64u, (n_dwords + 3u) / 4u);
    hipLaunchKernelGGL(pack_rows_kernel, grid, dim3(256), 0, ctx->stream, d_rows, ctx->n_reads, ctx->n_cols,
                       ctx->d_msa, ctx->col_stride);
}
