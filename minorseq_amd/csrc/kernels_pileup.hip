// kernels_pileup.hip — column pileup + per-codon histogram over column-packed reads.
//
// Behaviour implemented: doc/JULIET.md:99-100 (per-column counts of A C G T - N), :21-27/:94-98
// (codon-wise counting, deletions ignored), :256-259 (N does not count towards the coverage);
// details in docs/SPEC.md §2-3.
//
// Shape of the work.  The matrix is n_cols columns of n_reads 4-bit codes; every cell is read exactly
// once (algorithmic bytes = n_reads * n_cols / 2), so the kernel is an HBM stream and the design goal is
// to keep the per-nibble VALU work below the ~6 lane-ops the chip has per nibble at full HBM rate:
//   * a lane loads 16 B = 32 reads of one column (a wave = 1 KiB contiguous per load),
//   * column counts are bit-sliced: six masked popcounts per 8 nibbles give {b0,b1,b2,b0&b1,b1&b2,b0&b2}
//     plane counts, from which A C G T - N follow by a linear solve once per block,
//   * codon histograms are counted against a per-column seed base: eight reads are compared at once with
//     xor/or on the three column words; reads equal to the seed codon are counted by popcount (the
//     contended "major codon" bin never sees an atomic), the rare valid mismatches take an LDS atomic each.
//     The seed only steers which bin is counted the fast way — any seed gives the same histogram.
// A block owns W consecutive columns (plus a 2-column halo when a codon straddles its right edge) and a
// strided set of 8192-read tiles; per-thread counters are reduced once per block and flushed with integer
// atomics, which commute, so results are bit-exact and order-independent.
#include "jl_internal.h"

namespace {

constexpr uint32_t kM1 = 0x11111111u, kM2 = 0x22222222u, kM4 = 0x44444444u;
constexpr uint32_t kNone = 0x66666666u;

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
    // full 64-lane sum by DPP; the total lands in lane 63
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);  // row_half_mirror
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);  // row_mirror
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false); // row_bcast:15 -> rows 1,3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false); // row_bcast:31 -> rows 2,3
    return v;
}

// raw plane counts kept per column: 0:b0 1:b1 2:b2 3:b0&b1 (T) 4:b1&b2 (uncovered) 5:b0&b2 (N)
__device__ __forceinline__ void count_planes(uint32_t w, uint32_t (&a)[6])
{
    a[0] += __popc(w & kM1);
    a[1] += __popc(w & kM2);
    a[2] += __popc(w & kM4);
    uint32_t t = w & (w >> 1);
    a[3] += __popc(t & kM1);
    a[4] += __popc(t & kM2);
    a[5] += __popc(w & (w >> 2) & kM1);
}

template <int W>
__global__ __launch_bounds__(256) void pileup_kernel(const uint8_t *__restrict__ msa, uint64_t col_stride,
                                                      uint32_t n_cols, uint32_t n_tiles,
                                                      const uint8_t *__restrict__ colflag,
                                                      const uint8_t *__restrict__ guess,
                                                      uint32_t *__restrict__ counts, uint32_t *__restrict__ hist)
{
    __shared__ uint32_t s_hist[W][64];
    __shared__ uint32_t s_raw[W][6];
    __shared__ uint32_t s_mism[W];
    __shared__ uint32_t s_words;

    const uint32_t tid = threadIdx.x;
    const uint32_t c0 = blockIdx.x * W;

    for (uint32_t i = tid; i < W * 64; i += 256) (&s_hist[0][0])[i] = 0;
    if (tid < W * 6) (&s_raw[0][0])[tid] = 0;
    if (tid < W) s_mism[tid] = 0;
    if (tid == 0) s_words = 0;
    __syncthreads();

    // per-column metadata is block-uniform
    uint32_t startf = 0;     // bit j: a codon starts at column c0+j
    uint32_t g[W + 2];       // seed base of column c0+j replicated into every nibble
#pragma unroll
    for (int j = 0; j < W + 2; ++j) {
        uint32_t c = c0 + j;
        uint32_t b = c < n_cols ? guess[c] & 3u : 0u;
        g[j] = b * kM1;
        if (j < W && c < n_cols && (colflag[c] & 1u)) startf |= 1u << j;
    }
    const bool need_halo = (startf >> (W - 2)) != 0;

    uint32_t acc[W][6];
    uint32_t mism[W];
#pragma unroll
    for (int j = 0; j < W; ++j) {
        mism[j] = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) acc[j][k] = 0;
    }
    uint32_t words = 0;

    for (uint32_t tile = blockIdx.y; tile < n_tiles; tile += gridDim.y) {
        const uint64_t off = (uint64_t)tile * JL_PILEUP_TILE_BYTES + (uint64_t)tid * 16u;
        if (off >= col_stride) continue;  // col_stride is a multiple of 128: a 16-B chunk is all in or all out
        uint32_t d[W + 2][4];
#pragma unroll
        for (int j = 0; j < W + 2; ++j) {
            const bool live = (c0 + j < n_cols) && (j < W || need_halo);
            if (live) {
                const uint4 v = *reinterpret_cast<const uint4 *>(msa + (uint64_t)(c0 + j) * col_stride + off);
                d[j][0] = v.x; d[j][1] = v.y; d[j][2] = v.z; d[j][3] = v.w;
            } else {
                d[j][0] = d[j][1] = d[j][2] = d[j][3] = kNone;
            }
        }
        words += 4;
#pragma unroll
        for (int j = 0; j < W; ++j) {
            if (c0 + j < n_cols) {
#pragma unroll
                for (int q = 0; q < 4; ++q) count_planes(d[j][q], acc[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < W; ++j) {
            if (startf & (1u << j)) {
                uint32_t mm[4];
                uint32_t any = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t w0 = d[j][q], w1 = d[j + 1][q], w2 = d[j + 2][q];
                    const uint32_t x = (w0 ^ g[j]) | (w1 ^ g[j + 1]) | (w2 ^ g[j + 2]);
                    const uint32_t m = (x | (x >> 1) | (x >> 2)) & kM1;  // read differs from the seed codon
                    mism[j] += __popc(m);
                    const uint32_t inv = ((w0 | w1 | w2) >> 2) & kM1;   // some code >= 4: not in coverage
                    mm[q] = m & ~inv;                                    // valid codon, not the seed one
                    any |= mm[q];
                }
                if (any) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        uint32_t rest = mm[q];
                        while (rest) {
                            const int b = __ffs((int)rest) - 1;
                            rest &= rest - 1;
                            const uint32_t idx = (((d[j][q] >> b) & 3u) << 4) | (((d[j + 1][q] >> b) & 3u) << 2) |
                                                 ((d[j + 2][q] >> b) & 3u);
                            atomicAdd(&s_hist[j][idx], 1u);
                        }
                    }
                }
            }
        }
    }

    // ---- block reduction: wave sums by DPP, one LDS atomic per wave and counter
    const bool last_lane = (tid & 63u) == 63u;
#pragma unroll
    for (int j = 0; j < W; ++j) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const uint32_t s = wave_sum(acc[j][k]);
            if (last_lane && s) atomicAdd(&s_raw[j][k], s);
        }
        const uint32_t sm = wave_sum(mism[j]);
        if (last_lane && sm) atomicAdd(&s_mism[j], sm);
    }
    {
        const uint32_t sw = wave_sum(words);
        if (last_lane && sw) atomicAdd(&s_words, sw);
    }
    __syncthreads();

    const uint32_t nib = s_words * 8u;  // nibbles this block looked at, per column
    if (nib == 0) return;
    if (tid < W && c0 + tid < n_cols) {
        const uint32_t j = tid;
        const uint32_t b0 = s_raw[j][0], b1 = s_raw[j][1], b2 = s_raw[j][2];
        const uint32_t nT = s_raw[j][3], nU = s_raw[j][4], nN = s_raw[j][5];
        const uint32_t nC = b0 - nT - nN;   // 1 = 001 ; b0 set in {1,3,5}
        const uint32_t nG = b1 - nT - nU;   // 2 = 010 ; b1 set in {2,3,6}
        const uint32_t nD = b2 - nN - nU;   // 4 = 100 ; b2 set in {4,5,6}
        const uint32_t nA = nib - (nC + nG + nT + nD + nN + nU);
        uint32_t *o = counts + (uint64_t)(c0 + j) * 6u;
        if (nA) atomicAdd(o + 0, nA);
        if (nC) atomicAdd(o + 1, nC);
        if (nG) atomicAdd(o + 2, nG);
        if (nT) atomicAdd(o + 3, nT);
        if (nD) atomicAdd(o + 4, nD);
        if (nN) atomicAdd(o + 5, nN);
        if (startf & (1u << j)) {
            // reads equal to the seed codon were only counted, never binned
            const uint32_t seed = ((g[j] & 3u) << 4) | ((g[j + 1] & 3u) << 2) | (g[j + 2] & 3u);
            s_hist[j][seed] += nib - s_mism[j];
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < W * 64; i += 256) {
        const uint32_t j = i >> 6;
        const uint32_t v = s_hist[j][i & 63u];
        if (v && (startf & (1u << j))) atomicAdd(hist + (uint64_t)(c0 + j) * 64u + (i & 63u), v);
    }
}

// Seed base per column for majority-codon mode: majority base among the first reads of the column.
// (Any value is correct; a good seed keeps the codon compare on its fast path.)
__global__ __launch_bounds__(64) void guess_kernel(const uint8_t *__restrict__ msa, uint64_t col_stride,
                                                   uint32_t n_cols, uint8_t *__restrict__ guess)
{
    const uint32_t c = blockIdx.x;
    if (c >= n_cols) return;
    const uint32_t lane = threadIdx.x;
    uint32_t a[6] = {0, 0, 0, 0, 0, 0};
    uint32_t words = 0;
    // up to 64 lanes x 16 B = 2048 reads
    const uint64_t off = (uint64_t)lane * 16u;
    if (off < col_stride) {
        const uint4 v = *reinterpret_cast<const uint4 *>(msa + (uint64_t)c * col_stride + off);
        count_planes(v.x, a); count_planes(v.y, a); count_planes(v.z, a); count_planes(v.w, a);
        words = 4;
    }
    uint32_t s[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) s[k] = wave_sum(a[k]);
    const uint32_t sw = wave_sum(words);
    if (lane == 63) {
        const uint32_t nT = s[3], nU = s[4], nN = s[5];
        const uint32_t nC = s[0] - nT - nN, nG = s[1] - nT - nU, nD = s[2] - nN - nU;
        const uint32_t nA = sw * 8u - (nC + nG + nT + nD + nN + nU);
        uint32_t best = 0, bv = nA;
        if (nC > bv) { bv = nC; best = 1; }
        if (nG > bv) { bv = nG; best = 2; }
        if (nT > bv) { bv = nT; best = 3; }
        guess[c] = (uint8_t)best;
    }
}

}  // namespace

#ifndef JL_PILEUP_W
#define JL_PILEUP_W 12
#endif

const char *jl_pileup_kernel_name(void) { return "pileup_kernel"; }

void jl_launch_guess(jl_ctx *ctx)
{
    hipLaunchKernelGGL(guess_kernel, dim3(ctx->n_cols), dim3(64), 0, ctx->stream, ctx->d_msa, ctx->col_stride,
                       ctx->n_cols, ctx->d_guess);
}

void jl_launch_pileup(jl_ctx *ctx)
{
    constexpr int W = JL_PILEUP_W;
    const uint32_t n_chunks = (ctx->n_cols + W - 1) / W;
    const uint32_t n_tiles = (uint32_t)((ctx->col_stride + JL_PILEUP_TILE_BYTES - 1) / JL_PILEUP_TILE_BYTES);
    // enough blocks to fill 256 CUs a few times over, without splitting the reads finer than one tile
    uint32_t rsplit = (2048u + n_chunks - 1) / n_chunks;
    if (rsplit > n_tiles) rsplit = n_tiles;
    if (rsplit < 1) rsplit = 1;
    if (rsplit > 65535u) rsplit = 65535u;
    hipLaunchKernelGGL(pileup_kernel<W>, dim3(n_chunks, rsplit), dim3(256), 0, ctx->stream, ctx->d_msa,
                       ctx->col_stride, ctx->n_cols, n_tiles, ctx->d_colflag, ctx->d_guess, ctx->d_counts,
                       ctx->d_hist);
}
