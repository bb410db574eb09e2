// kernels_util.hip — device-side producers of the resident matrix (three bit planes per column, jl_internal.h); every one
// of them writes the planes directly:
//   synth_kernel              synthetic aligned CCS reads (jl_synth.h)
//   pack_rows_kernel          by-row uint8 codes (jl_msa_pack_rows)
//   nibbles_to_planes_kernel  the interchange format of jl_msa_upload, validated on the way (planes_to_nibbles_kernel: jl_msa_download)
//   (aligned BAM records: kernels_ingest.hip)
#include <string.h>

#include <algorithm>

#include "jl_internal.h"
#include "planes.h"

namespace {

constexpr int kSynthColsPerBlock = 64;

// four lanes hold one byte each (lane & 3 = byte index): every one of them gets the dword
__device__ __forceinline__ uint32_t quad_bytes_to_dword(uint32_t byte, uint32_t lane)
{
    uint32_t x = byte << (8u * (lane & 3u));
    x |= (uint32_t)__shfl_xor((int)x, 1, 64);
    x |= (uint32_t)__shfl_xor((int)x, 2, 64);
    return x;
}

// A thread's eight codes of one column (nibbles of w) into the planes: the four lanes of a quad put their bytes together
// and lanes 0..2 of the quad store the dword of plane 0..2.  `t` = the thread's byte within a plane (8 reads).
__device__ __forceinline__ void store_planes_quad(uint8_t *__restrict__ col_base, uint64_t plane_stride, uint64_t t, uint32_t w)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t d0 = quad_bytes_to_dword(jl_plane_bits8(w, 0), lane);
    const uint32_t d1 = quad_bytes_to_dword(jl_plane_bits8(w, 1), lane);
    const uint32_t d2 = quad_bytes_to_dword(jl_plane_bits8(w, 2), lane);
    const uint32_t k = lane & 3u;
    if (k < 3u && t < plane_stride)
        *reinterpret_cast<uint32_t *>(col_base + (uint64_t)k * plane_stride + (t & ~(uint64_t)3)) = k == 0 ? d0 : (k == 1 ? d1 : d2);
}

// `ref`, `msa`: the window's columns [col0, col0 + win_cols) of the pl.n_cols-column reference the plan describes.
// thread = 8 reads (one byte of every plane), a block's 256 threads x 64 columns
__global__ __launch_bounds__(256) void synth_kernel(jl_synth_plan pl, const uint8_t *__restrict__ ref,
                                                     uint8_t *__restrict__ msa, uint64_t plane_stride,
                                                     uint64_t n_reads, uint32_t col0, uint32_t win_cols)
{
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;  // byte index in a plane (plane_stride is a multiple of 4: whole quads)
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
        store_planes_quad(msa + (uint64_t)c * 3u * plane_stride, plane_stride, t, w);
    }
}

// by-row uint8 codes -> planes.  thread (x = 8 reads, y = column): gathers 8 reads of one column; the quad's bytes leave
// as dwords.  (A test-size path: the by-row reads are strided.)
__global__ __launch_bounds__(256) void pack_rows_kernel(const uint8_t *__restrict__ rows, uint64_t n_reads,
                                                         uint32_t n_cols, uint8_t *__restrict__ msa,
                                                         uint64_t plane_stride)
{
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint32_t c = blockIdx.y;
    uint32_t w = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const uint64_t i = t * 8u + r;
        const uint32_t s = i < n_reads ? rows[i * n_cols + c] : 6u;
        w |= (s & 7u) << (4 * r);
    }
    store_planes_quad(msa + (uint64_t)c * 3u * plane_stride, plane_stride, t, w);
}

// Interchange format (column-packed nibbles: the host form of jl_msa_upload / jl_msa_download) -> planes, fused with the
// validation of an upload: any nibble outside 0..6 (code 7 or bit 3 set) is rejected (SPEC §1).  One thread = 64 reads of
// one column: 32 bytes of nibbles in, 8 bytes of each plane out.  Reads past the source column's stride are padding (6).
__global__ __launch_bounds__(256) void nibbles_to_planes_kernel(const uint8_t *__restrict__ nib, uint64_t nib_stride, uint64_t n_reads,
                                                                uint8_t *__restrict__ planes, uint64_t plane_stride, uint32_t c0,
                                                                uint32_t *__restrict__ bad)
{
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;   // 64-read unit within the column
    if (t * 8u >= plane_stride) return;
    const uint32_t c = blockIdx.y;
    uint32_t w[8];
    if (t * 32u + 32u <= nib_stride) {
        const uint4 *src = reinterpret_cast<const uint4 *>(nib + (uint64_t)c * nib_stride + t * 32u);
        const uint4 a = src[0], b = src[1];
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
    } else {
#pragma unroll
        for (uint32_t q = 0; q < 8u; ++q)
            w[q] = t * 32u + q * 4u + 4u <= nib_stride ? *reinterpret_cast<const uint32_t *>(nib + (uint64_t)c * nib_stride + t * 32u + q * 4u)
                                                        : 0x66666666u;
    }
    // whatever the caller's buffer holds past its last read is padding here
    if (t * 64u + 64u > n_reads) {
#pragma unroll
        for (uint32_t q = 0; q < 8u; ++q) {
            const uint64_t first = t * 64u + q * 8u;
            if (first >= n_reads) w[q] = 0x66666666u;
            else if (first + 8u > n_reads) {
                const uint32_t keep = (1u << (4u * (uint32_t)(n_reads - first))) - 1u;
                w[q] = (w[q] & keep) | (0x66666666u & ~keep);
            }
        }
    }
    uint32_t any = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) any |= (w[q] & 0x88888888u) | (w[q] & (w[q] >> 1) & (w[q] >> 2) & 0x11111111u);
    if (any && bad) atomicOr(bad, 1u);
#pragma unroll
    for (uint32_t k = 0; k < 3u; ++k) {
        uint2 o;
        o.x = jl_plane_bits8(w[0], k) | (jl_plane_bits8(w[1], k) << 8) | (jl_plane_bits8(w[2], k) << 16) | (jl_plane_bits8(w[3], k) << 24);
        o.y = jl_plane_bits8(w[4], k) | (jl_plane_bits8(w[5], k) << 8) | (jl_plane_bits8(w[6], k) << 16) | (jl_plane_bits8(w[7], k) << 24);
        *reinterpret_cast<uint2 *>(planes + ((uint64_t)(c0 + c) * 3u + k) * plane_stride + t * 8u) = o;
    }
}

// ... and back (jl_msa_download): one thread = 32 reads of one column, a dword of each plane in, 16 bytes of nibbles out
__global__ __launch_bounds__(256) void planes_to_nibbles_kernel(const uint8_t *__restrict__ planes, uint64_t plane_stride, uint32_t c0,
                                                                uint8_t *__restrict__ nib, uint64_t nib_stride)
{
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;   // 32-read unit
    if (t * 16u >= nib_stride) return;
    const uint32_t c = blockIdx.y;
    uint32_t b[3] = {0u, 0xFFFFFFFFu, 0xFFFFFFFFu};   // past the planes: padding
    if (t * 4u < plane_stride) {
#pragma unroll
        for (uint32_t k = 0; k < 3u; ++k) b[k] = *reinterpret_cast<const uint32_t *>(planes + ((uint64_t)(c0 + c) * 3u + k) * plane_stride + t * 4u);
    }
    uint32_t o[4];
#pragma unroll
    for (uint32_t q = 0; q < 4u; ++q) o[q] = jl_planes_to_nibbles8((b[0] >> (8u * q)) & 0xFFu, (b[1] >> (8u * q)) & 0xFFu, (b[2] >> (8u * q)) & 0xFFu);
    *reinterpret_cast<uint4 *>(nib + (uint64_t)c * nib_stride + t * 16u) = make_uint4(o[0], o[1], o[2], o[3]);
}


// Insertions per window column (doc/FUSE.md:19 "Fuse includes in-frame insertions"): they are not part of the MSA
// (doc/JULIET.md:26-27), so they are counted from the records.  One thread per read walks its cigar; an insertion sits
// BEFORE the window column of the next reference base: len_hist[c][min(len, 31)]++, and for IN-FRAME insertions of at most
// 30 bases base_counts[c][j][base]++ for the inserted bases.  Integer atomics commute: bit-exact against the oracle's loops.
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
                // bases only of insertions that can enter a consensus: in-frame, at most 30 long (an out-of-frame insertion
                // at the same column must not vote on the bases of the accepted one; SPEC §11)
                const bool votes = len % 3u == 0u && len <= JL_INS_MAX_BASES;
                for (uint32_t j = 0; votes && j < len; ++j) {
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

// the staged form of a group's exchange: the gathered heads, device region -> pinned host region, a workgroup a head (a kernel
// on the group's stream costs the launching thread and the device a third of what a device-to-host hipMemcpyAsync of the
// same 50 KB does)
__global__ __launch_bounds__(256) void heads_to_host_kernel(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst)
{
    const size_t at = (size_t)blockIdx.x * (JL_PACK_HEAD_BYTES / 4);
    for (uint32_t i = threadIdx.x; i < JL_PACK_HEAD_BYTES / 4; i += 256) dst[at + i] = src[at + i];
}
void jl_launch_heads_to_host(const uint8_t *d_region, uint8_t *h_region, uint32_t n_heads, hipStream_t st)
{
    hipLaunchKernelGGL(heads_to_host_kernel, dim3(n_heads), dim3(256), 0, st, reinterpret_cast<const uint32_t *>(d_region), reinterpret_cast<uint32_t *>(h_region));
}

void jl_launch_done(jl_ctx *ctx)
{
    hipLaunchKernelGGL(done_kernel, dim3(1), dim3(1), 0, ctx->stream, ctx->d_sync, ctx->h_seq);
}
void jl_launch_done_on(jl_ctx *ctx, hipStream_t st)
{
    hipLaunchKernelGGL(done_kernel, dim3(1), dim3(1), 0, st, ctx->d_sync, ctx->h_seq);
}

// jl_run_pileup_clock: a one-thread node that stores the device's constant-rate (100 MHz) clock into pinned host memory
__global__ void clock_kernel(volatile unsigned long long *dst) { *dst = wall_clock64(); }
void jl_launch_clock(jl_ctx *ctx, hipStream_t st, uint32_t which)
{
    hipLaunchKernelGGL(clock_kernel, dim3(1), dim3(1), 0, st, reinterpret_cast<volatile unsigned long long *>(const_cast<uint32_t *>(ctx->h_seq) + 8u + 2u * which));
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

void jl_launch_nibbles_to_planes(jl_ctx *ctx, const uint8_t *d_nib, uint64_t nib_stride, uint32_t c0, uint32_t n, uint32_t *d_bad)
{
    const uint64_t units = ctx->plane_stride / 8u;   // plane_stride is a multiple of 16
    hipLaunchKernelGGL(nibbles_to_planes_kernel, dim3((uint32_t)((units + 255u) / 256u), n), dim3(256), 0, ctx->stream, d_nib, nib_stride,
                       ctx->n_reads, ctx->d_msa, ctx->plane_stride, c0, d_bad);
}

void jl_launch_planes_to_nibbles(jl_ctx *ctx, uint8_t *d_nib, uint64_t nib_stride, uint32_t c0, uint32_t n)
{
    const uint64_t units = nib_stride / 16u;
    hipLaunchKernelGGL(planes_to_nibbles_kernel, dim3((uint32_t)((units + 255u) / 256u), n), dim3(256), 0, ctx->stream, ctx->d_msa,
                       ctx->plane_stride, c0, d_nib, nib_stride);
}

void jl_launch_insertions(jl_ctx *ctx, const int32_t *d_pos, const uint32_t *d_cigar, const uint64_t *d_cig_off,
                          const uint8_t *d_seq4, const uint64_t *d_seq_off)
{
    hipLaunchKernelGGL(insertions_kernel, dim3((uint32_t)((ctx->n_reads + 255u) / 256u)), dim3(256), 0, ctx->stream, ctx->n_reads,
                       ctx->n_cols, ctx->win_begin, d_pos, d_cigar, d_cig_off, d_seq4, d_seq_off, ctx->d_ins_len, ctx->d_ins_base);
}

void jl_launch_synth(jl_ctx *ctx, const jl_synth_plan *plan, const uint8_t *d_ref, uint32_t col0)
{
    dim3 grid((uint32_t)((ctx->plane_stride + 255u) / 256u), (ctx->n_cols + kSynthColsPerBlock - 1) / kSynthColsPerBlock);
    hipLaunchKernelGGL(synth_kernel, grid, dim3(256), 0, ctx->stream, *plan, d_ref, ctx->d_msa, ctx->plane_stride,
                       ctx->n_reads, col0, ctx->n_cols);
}

void jl_launch_pack_rows(jl_ctx *ctx, const uint8_t *d_rows)
{
    dim3 grid((uint32_t)((ctx->plane_stride + 255u) / 256u), ctx->n_cols);
    hipLaunchKernelGGL(pack_rows_kernel, grid, dim3(256), 0, ctx->stream, d_rows, ctx->n_reads, ctx->n_cols,
                       ctx->d_msa, ctx->plane_stride);
}
