// kernels_pileup.hip — column pileup + per-codon histogram over column-packed reads.
//
// Behaviour implemented: doc/JULIET.md:99-100 (per-column counts of A C G T - N), :21-27/:94-98
// (codon-wise counting, deletions ignored), :256-259 (N does not count towards the coverage);
// details in docs/SPEC.md §2-3.
//
// Shape of the work.  The matrix is n_cols columns of n_reads 4-bit codes; every cell is read exactly
// once (algorithmic bytes = n_reads * n_cols / 2), so the kernel is an HBM stream and the design goal is
// to keep the per-nibble VALU work well below the ~6 lane-ops the chip has per nibble at full HBM rate:
//   * a lane loads 16 B = 32 reads of one column (a wave = 1 KiB contiguous per load),
//   * column counts are six linear measurements per 8 nibbles, four of them one v_dot8_u32_u4 each
//     (sum of codes, sum of squares, sum over odd codes, sum over codes >= 4) plus two popcounts; the
//     counts of A C G T - N follow from an exact integer solve once per lane and flush,
//   * codon histograms are counted against a per-column seed base: eight reads are compared at once with
//     xor/or on the three column words; reads equal to the seed codon are counted by popcount (the
//     contended "major codon" bin never sees an atomic), the rare valid mismatches take an LDS atomic each.
//     The seed only steers which bin is counted the fast way — any seed gives the same histogram.
// A block owns one chunk of <= W consecutive columns from a host-built table (capi.hip: build_chunks) and a
// strided set of 8192-read tiles.  Chunks start on codon boundaries, so stretches that are locally single-frame
// need no halo; only where a codon of the chunk reaches past its last column are the next two columns loaded
// as well.  Loads are non-temporal (every cell is read once) and the next tile is prefetched into a second register
// set while the current one is counted.  Per-lane counters are packed two per register, wave-reduced by DPP and
// flushed to LDS at most every 31 tiles (16-bit fields cannot overflow).  A block that counts its chunk alone
// (gridDim.y = 1) stores the totals; when the reads of a long column are split over several blocks they go to HBM
// with integer atomics, which commute, so results are bit-exact and order-independent either way.  There is no
// reuse between blocks (halo columns aside), so the block -> XCD mapping does not matter here.
// pileup_group_kernel runs the same body for several windows in one launch (blockIdx.z = window).
#include <stdlib.h>
#include <string.h>

#include "call_eval.h"
#include "jl_internal.h"

#ifndef JL_CALL_MIN_WAVES
#define JL_CALL_MIN_WAVES 5
#endif

// Register prefetch of the next tile while the current one is counted: 27.4 us vs 31.6 us per 150 MB launch
// (rocprofv3), neutral to +3 % on GB-sized windows.  JL_PILEUP_PIPE=0 in the environment selects the plain loop.
#ifndef JL_PILEUP_PIPE
#define JL_PILEUP_PIPE 1
#endif

namespace {

constexpr uint32_t kM1 = 0x11111111u, kM4 = 0x44444444u;
constexpr uint32_t kNone = 0x66666666u;
constexpr uint32_t kFlushTiles = 31;  // 64 lanes x 32 reads x 31 tiles = 63488 < 2^16

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

// Six linear measurements of the eight 4-bit codes in w (codes 0..6):
//   a[0] S1  = sum code            a[1] S2 = sum code^2
//   a[2] B0  = #odd codes          a[3] B0W = sum of the odd codes
//   a[4] E   = #codes >= 4         a[5] D  = 4 * sum of the codes >= 4
__device__ __forceinline__ void measure(uint32_t w, uint32_t (&a)[6])
{
    const uint32_t t1 = w & kM1;
    const uint32_t t2 = w & kM4;
    a[0] = __builtin_amdgcn_udot8(w, kM1, a[0], false);
    a[1] = __builtin_amdgcn_udot8(w, w, a[1], false);
    a[2] += __popc(t1);
    a[3] = __builtin_amdgcn_udot8(t1, w, a[3], false);
    a[4] += __popc(t2);
    a[5] = __builtin_amdgcn_udot8(t2, w, a[5], false);
}

// the same six measurements without v_dot8 (tuning comparison): S1, S2 from bit-plane popcounts
__device__ __forceinline__ void measure_popc(uint32_t w, uint32_t (&a)[6])
{
    const uint32_t p0 = __popc(w & kM1), p1 = __popc(w & 0x22222222u), p2 = __popc(w & kM4);
    const uint32_t t = w & (w >> 1);
    const uint32_t n01 = __popc(t & kM1), n12 = __popc(t & 0x22222222u), n02 = __popc(w & (w >> 2) & kM1);
    // codes: C=1 G=2 T=3 D=4 N=5 U=6 ; nT=n01, nU=n12, nN=n02
    const uint32_t nC = p0 - n01 - n02, nG = p1 - n01 - n12, nD = p2 - n02 - n12;
    a[0] += nC + 2 * nG + 3 * n01 + 4 * nD + 5 * n02 + 6 * n12;
    a[1] += nC + 4 * nG + 9 * n01 + 16 * nD + 25 * n02 + 36 * n12;
    a[2] += p0;
    a[3] += nC + 3 * n01 + 5 * n02;
    a[4] += p2;
    a[5] += 4 * (4 * nD + 5 * n02 + 6 * n12);
}

// Exact integer solve of the measurements for the counts of C G T - N and uncovered.
__device__ __forceinline__ void solve(const uint32_t (&a)[6], uint32_t &nC, uint32_t &nG, uint32_t &nT, uint32_t &nD,
                                      uint32_t &nN, uint32_t &nU)
{
    const int S1 = (int)a[0], S2 = (int)a[1], B0 = (int)a[2], B0W = (int)a[3], E = (int)a[4], Dq = (int)(a[5] >> 2);
    const int n = (-3 * B0 + 2 * B0W + 2 * S1 + 8 * Dq - 24 * E - S2) >> 3;
    const int t = ((B0W - B0) >> 1) - 2 * n;
    const int c = B0 - t - n;
    const int u = (Dq - 4 * E - n) >> 1;
    const int d = E - n - u;
    const int g = (S1 - B0W - Dq + 5 * n) >> 1;
    nC = (uint32_t)c; nG = (uint32_t)g; nT = (uint32_t)t; nD = (uint32_t)d; nN = (uint32_t)n; nU = (uint32_t)u;
}

template <int W>
struct tile_regs {
    uint32_t d[W + 2][4];
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// chunk record (host-built, capi.hip: build_chunks): x = first column, y = own columns (bits 0-3) | codon-start
// flags of the own columns (bits 4-15) | "a codon reaches into the next two columns" (bit 16)
#define JL_CHUNK_META(ncols, startf, halo) ((uint32_t)(ncols) | ((uint32_t)(startf) << 4) | ((uint32_t)(halo) << 16))

// FAST = the common chunk of a single-frame stretch: exactly three own columns that are one codon (no halo).
// Everything the generic path decides per column at run time is a compile-time constant then.
template <int W, bool NT, bool FAST>
__device__ __forceinline__ void load_tile(tile_regs<W> &r, const uint8_t JL_AS1 *msa, uint64_t col_stride,
                                          uint32_t n_cols, uint32_t c0, uint32_t ncols, uint64_t off, bool need_halo)
{
#pragma unroll
    for (int j = 0; j < W + 2; ++j) {
        if (FAST && j >= 3) continue;  // never touched
        // own columns j < ncols; the two columns after them only when a codon of this chunk reaches into them
        const bool live = FAST ? true
                               : (c0 + j < n_cols) && ((uint32_t)j < ncols || (need_halo && (uint32_t)j < ncols + 2u));
        if (live) {
            const u32x4 JL_AS1 *src = (const u32x4 JL_AS1 *)(msa + (uint64_t)(c0 + j) * col_stride + off);
            // every cell is read exactly once: a non-temporal load keeps the stream from displacing L2 lines
            const u32x4 v = NT ? __builtin_nontemporal_load(src) : *src;
            r.d[j][0] = v.x; r.d[j][1] = v.y; r.d[j][2] = v.z; r.d[j][3] = v.w;
        } else {
            r.d[j][0] = r.d[j][1] = r.d[j][2] = r.d[j][3] = kNone;
        }
    }
}

template <int W, bool PIPE, int MODE, bool FAST>
__device__ __forceinline__ void pileup_stream(const uint8_t JL_AS1 *msa, uint64_t col_stride, uint32_t n_cols,
                                              uint32_t n_tiles, uint32_t c0, uint32_t ncols, uint32_t startf,
                                              bool need_halo, const uint32_t (&g)[W + 2], uint32_t (*s_hist)[64],
                                              uint32_t (*s_col)[6], uint32_t *s_match)
{
    constexpr bool NT = (MODE & 4) != 0;
    const uint32_t tid = threadIdx.x;
    const bool last_lane = (tid & 63u) == 63u;
    const uint64_t lane_off = (uint64_t)tid * 16u;

    uint32_t tile = blockIdx.y;
    tile_regs<W> nxt;
    bool nxt_live = false;
    if (PIPE && tile < n_tiles) {
        const uint64_t off = (uint64_t)tile * JL_PILEUP_TILE_BYTES + lane_off;
        nxt_live = off < col_stride;
        if (nxt_live) load_tile<W, NT, FAST>(nxt, msa, col_stride, n_cols, c0, ncols, off, need_halo);
    }

    while (tile < n_tiles) {
        uint32_t acc[W][6];
        uint32_t mism[W];
#pragma unroll
        for (int j = 0; j < W; ++j) {
            mism[j] = 0;
#pragma unroll
            for (int k = 0; k < 6; ++k) acc[j][k] = 0;
        }
        uint32_t words = 0;

        for (uint32_t it = 0; it < kFlushTiles && tile < n_tiles; ++it, tile += gridDim.y) {
            tile_regs<W> cur;
            bool live;
            if (PIPE) {
                cur = nxt;
                live = nxt_live;
                const uint32_t tn = tile + gridDim.y;
                nxt_live = false;
                if (tn < n_tiles) {
                    const uint64_t off = (uint64_t)tn * JL_PILEUP_TILE_BYTES + lane_off;
                    nxt_live = off < col_stride;
                    if (nxt_live) load_tile<W, NT, FAST>(nxt, msa, col_stride, n_cols, c0, ncols, off, need_halo);
                }
            } else {
                const uint64_t off = (uint64_t)tile * JL_PILEUP_TILE_BYTES + lane_off;
                live = off < col_stride;  // col_stride is a multiple of 128: a 16-B chunk is all in or all out
                if (live) load_tile<W, NT, FAST>(cur, msa, col_stride, n_cols, c0, ncols, off, need_halo);
            }
            if (!live) continue;
            words += 4;
            if (MODE & 1) {  // tuning probe: the loads alone (results are wrong by design)
#pragma unroll
                for (int j = 0; j < W; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[j][q] ^= cur.d[j][q];
                continue;
            }
#pragma unroll
            for (int j = 0; j < W; ++j) {
                if (FAST ? j < 3 : (uint32_t)j < ncols) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (MODE & 2) measure_popc(cur.d[j][q], acc[j]);
                        else measure(cur.d[j][q], acc[j]);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < W; ++j) {
                if (FAST ? j == 0 : (startf & (1u << j)) != 0) {
                    uint32_t mm[4];
                    uint32_t any = 0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint32_t w0 = cur.d[j][q], w1 = cur.d[j + 1][q], w2 = cur.d[j + 2][q];
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
                                const uint32_t idx = (((cur.d[j][q] >> b) & 3u) << 4) |
                                                     (((cur.d[j + 1][q] >> b) & 3u) << 2) |
                                                     ((cur.d[j + 2][q] >> b) & 3u);
                                atomicAdd(&s_hist[j][idx], 1u);
                            }
                        }
                    }
                }
            }
        }

        // ---- flush this batch: per-lane solve, 16-bit packing, DPP wave sums, one LDS atomic per wave and counter
        const uint32_t nib = words * 8u;
#pragma unroll
        for (int j = 0; j < W; ++j) {
            if (FAST && j >= 3) continue;
            uint32_t nC, nG, nT, nD, nN, nU;
            solve(acc[j], nC, nG, nT, nD, nN, nU);
            const uint32_t nA = nib - (nC + nG + nT + nD + nN + nU);
            const uint32_t p0 = wave_sum(nA | (nC << 16));
            const uint32_t p1 = wave_sum(nG | (nT << 16));
            const uint32_t p2 = wave_sum(nD | (nN << 16));
            const bool starts = FAST ? j == 0 : (startf & (1u << j)) != 0;
            uint32_t p3 = 0;
            if (starts) p3 = wave_sum((nib - mism[j]) & 0xFFFFu);  // reads equal to the seed codon (block-uniform branch)
            if (last_lane && (uint32_t)j < ncols) {
                if (p0 & 0xFFFFu) atomicAdd(&s_col[j][0], p0 & 0xFFFFu);
                if (p0 >> 16) atomicAdd(&s_col[j][1], p0 >> 16);
                if (p1 & 0xFFFFu) atomicAdd(&s_col[j][2], p1 & 0xFFFFu);
                if (p1 >> 16) atomicAdd(&s_col[j][3], p1 >> 16);
                if (p2 & 0xFFFFu) atomicAdd(&s_col[j][4], p2 & 0xFFFFu);
                if (p2 >> 16) atomicAdd(&s_col[j][5], p2 >> 16);
                if (starts && p3) atomicAdd(&s_match[j], p3);
            }
        }
    }
}

// CALL: the Fisher stage runs here too — the workgroup that counted a codon evaluates its positions from the histogram
// still in LDS (one wave per position, call_eval.h) instead of a later launch reading it back from HBM.  Only when one
// workgroup counts a chunk alone (gridDim.y = 1); the host picks the variant.
template <int W, bool PIPE, int MODE, bool CALL>
__device__ __forceinline__ void pileup_body(const uint8_t JL_AS1 *msa, uint64_t col_stride, uint32_t n_cols,
                                            uint32_t n_tiles, const uint2 JL_AS1 *chunks, const uint32_t JL_AS1 *guess32,
                                            uint32_t JL_AS1 *counts, uint32_t JL_AS1 *hist, const jl_callinfo *ci_mem)
{
    __shared__ uint32_t s_hist[W][64];
    __shared__ uint32_t s_col[W][6];   // A C G T - N
    __shared__ uint32_t s_match[W];

    const uint32_t tid = threadIdx.x;
    // chunks come from a host-built table: each starts on a codon boundary of the locally dominant frame, so
    // single-frame stretches never need halo columns even when different genes use different frames.  The record
    // and the seed bases are block-uniform: scalar loads, all issued before the first wait.
    const uint2 rec = chunks[blockIdx.x];
    const uint32_t c0 = rec.x;
    const uint32_t ncols = rec.y & 15u;               // own columns, 1..W
    const uint32_t startf = (rec.y >> 4) & 0xFFFu;    // bit j: a codon starts at column c0+j
    const bool need_halo = ((rec.y >> 16) & 1u) != 0;

    // seed base of column c0+j replicated into every nibble; the byte array is padded, so the aligned dwords
    // covering bytes c0 .. c0+W+1 are always in bounds
    constexpr int NG = (W + 2 + 3 + 3) / 4;
    uint32_t gw[NG];
#pragma unroll
    for (int k = 0; k < NG; ++k) gw[k] = guess32[(c0 >> 2) + k];
    uint32_t g[W + 2];
#pragma unroll
    for (int j = 0; j < W + 2; ++j) {
        const uint32_t b = (c0 & 3u) + (uint32_t)j;
        uint32_t word = gw[0];
#pragma unroll
        for (int k = 1; k < NG; ++k)
            if ((b >> 2) == (uint32_t)k) word = gw[k];
        g[j] = ((word >> (8u * (b & 3u))) & 3u) * kM1;
    }

    for (uint32_t i = tid; i < W * 64; i += 256) (&s_hist[0][0])[i] = 0;
    if (tid < W * 6) (&s_col[0][0])[tid] = 0;
    if (tid < W) s_match[tid] = 0;
    __syncthreads();

    if (MODE & 8) {  // tuning probe: fixed costs only
    } else if (W == 3 && rec.y == JL_CHUNK_META(3, 1, 0))
        pileup_stream<W, PIPE, MODE, true>(msa, col_stride, n_cols, n_tiles, c0, ncols, startf, need_halo, g, s_hist, s_col, s_match);
    else
        pileup_stream<W, PIPE, MODE, false>(msa, col_stride, n_cols, n_tiles, c0, ncols, startf, need_halo, g, s_hist, s_col, s_match);
    __syncthreads();

    // gridDim.y == 1: this block is the only one that counts its chunk, so the totals are plain stores and the
    // output needs no zeroing pass; otherwise integer atomics into the zeroed region (they commute: bit-exact)
    const bool excl = gridDim.y == 1;
    if (tid < W * 6) {
        const uint32_t j = tid / 6u, k = tid - j * 6u;
        const uint32_t v = s_col[j][k];
        if (j < ncols) {
            if (excl) counts[(uint64_t)(c0 + j) * 6u + k] = v;
            else if (v) atomicAdd((uint32_t *)(counts + (uint64_t)(c0 + j) * 6u + k), v);
        }
    }
    if (tid < W && (startf & (1u << tid))) {
        // reads equal to the seed codon were only counted, never binned
        const uint32_t j = tid;
        const uint32_t seed = ((g[j] & 3u) << 4) | ((g[j + 1] & 3u) << 2) | (g[j + 2] & 3u);
        s_hist[j][seed] += s_match[j];
    }
    __syncthreads();
    for (uint32_t i = tid; i < W * 64; i += 256) {
        const uint32_t j = i >> 6;
        const uint32_t v = s_hist[j][i & 63u];
        if (startf & (1u << j)) {
            if (excl) hist[(uint64_t)(c0 + j) * 64u + (i & 63u)] = v;
            else if (v) atomicAdd((uint32_t *)(hist + (uint64_t)(c0 + j) * 64u + (i & 63u)), v);
        }
    }
    if (CALL) {
        // ---- Fisher stage of this chunk's positions: wave w takes every fourth codon start.
        const uint32_t wid = tid >> 6, lane = tid & 63u;
        const jl_callinfo *ci = ci_mem;
        if (blockIdx.x == 0 && tid == 0 && ci->meta) {
            // counters of the phasing launch that follows on the stream
            jl_phase_meta *m = ci->meta;
            m->n_occupied = 0;
            m->overflow = 0;
            jl_phase_summary z = {0, 0, 0, 0, 0, 0, 0, 0};
            m->summary = z;
        }
        uint32_t nstart = 0;
#pragma clang loop unroll(disable)
        for (uint32_t j = 0; j < (uint32_t)W; ++j) {
            if (!(startf & (1u << j))) continue;
            if ((nstart++ & 3u) != wid) continue;
            const uint32_t col = c0 + j;
            const uint32_t h = (&s_hist[0][0])[j * 64u + lane];
#pragma clang loop unroll(disable)
            for (uint32_t p = ci->col_first[col]; p != 0xFFFFFFFFu; p = ci->pos_next[p])   // wave-uniform
                jl_call_position<false>(ci->A, p, col, h, ci->pos_refcfg[p], ci->pos_gene[p], ci->pos_codon[p], ci->drm,
                                        ci->called, ci->staged);
        }
    }
}

template <int W, bool PIPE, int MODE, bool CALL>
__global__ __launch_bounds__(256, CALL ? JL_CALL_MIN_WAVES : 1) void pileup_kernel(const uint8_t *__restrict__ msa, uint64_t col_stride,
                                                      uint32_t n_cols, uint32_t n_tiles,
                                                      const uint2 *__restrict__ chunks,
                                                      const uint32_t *__restrict__ guess32,
                                                      uint32_t *__restrict__ counts, uint32_t *__restrict__ hist,
                                                      const jl_callinfo *__restrict__ ci)
{
    pileup_body<W, PIPE, MODE, CALL>((const uint8_t JL_AS1 *)msa, col_stride, n_cols, n_tiles, (const uint2 JL_AS1 *)chunks,
                                     (const uint32_t JL_AS1 *)guess32, (uint32_t JL_AS1 *)counts, (uint32_t JL_AS1 *)hist, ci);
}

// One launch over several resident windows (blockIdx.z = window, argument blocks in device memory): the stream of a
// 150 MB window is too short to hide a launch's ramp and drain, four of them in one grid run at the rate of a
// 600 MB stream.  Every window is counted by one block per chunk (gridDim.y = 1: plain stores, no zeroing pass).
template <int W, bool PIPE, int MODE, bool CALL>
__global__ __launch_bounds__(256, CALL ? JL_CALL_MIN_WAVES : 1) void pileup_group_kernel(jl_pileup_group_args args)
{
    const jl_win_pileup &w = args.w[blockIdx.z];
    if (blockIdx.x >= w.n_chunks) return;
    // the pointers come out of memory: say that they are global ones, or the loads become flat loads (JL_AS1)
    pileup_body<W, PIPE, MODE, CALL>((const uint8_t JL_AS1 *)w.msa, w.col_stride, w.n_cols, w.n_tiles, (const uint2 JL_AS1 *)w.chunks,
                                     (const uint32_t JL_AS1 *)w.guess32, (uint32_t JL_AS1 *)w.counts, (uint32_t JL_AS1 *)w.hist, w.ci);
}

// ---------------------------------------------------------------------------------------- the same from BIT PLANES
// The library's own copy of the matrix (jl_ctx::d_planes): per column three planes — bit k of every read's code, reads in
// bit order — so a cell costs 3 bits of HBM traffic where the nibble layout costs 4, and one instruction handles 32 reads
// where it handled 8:
//   * column counts: six popcounts per 32 reads (b0, b1, b2, b0&b1, b0&b2, b1&b2; code 7 does not occur), from which
//     T = n01, N = n02, uncovered = n12, C = n0 - n01 - n02, G = n1 - n01 - n12, '-' = n2 - n02 - n12, A = the rest;
//   * codons against the seed: six xors against the seed's bits (block-uniform all-ones / all-zero words), or-ed together
//     with the three b2 words (a code >= 4 anywhere: not a codon): reads equal to the seed codon are a popcount, the rare
//     valid mismatches are walked bit by bit into the LDS histogram as before.
// Same chunks, same seeds, same outputs, bit for bit: the two kernels count the same cells.  A lane takes 8 bytes = 64
// reads of each plane per tile (a tile = 16384 reads); the next tile's 9 .. 24 words are prefetched as in the nibble kernel.
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
// NQ = dwords of a plane a lane reads per tile.  16 bytes (NQ = 4) stream best: 142 us per 0.9 GB launch of eight windows
// against 149 with 8-byte loads — at 154 registers, three waves per SIMD, which a single short window pays for (22.8 us
// against 20.5 for 100k reads x 3 kb): grouped launches and deep windows take NQ = 4, a short window alone NQ = 2; the
// six-column chunks always NQ = 2 (eight columns of tiles twice over: 249 registers at 16 bytes).
constexpr uint32_t plane_tile_bytes(int nq) { return 256u * 4u * (uint32_t)nq; }
constexpr uint32_t plane_flush_tiles(int nq) { return 1023u / (32u * (uint32_t)nq); }   // 64 lanes x 32 nq reads x tiles < 2^16

template <int W, int NQ>
struct ptile_regs {
    uint32_t d[W + 2][3][NQ];   // [column][plane][dword]
};

template <int W, int NQ, bool FAST>
__device__ __forceinline__ void load_ptile(ptile_regs<W, NQ> &r, const uint8_t JL_AS1 *planes, uint64_t plane_stride, uint32_t n_cols,
                                           uint32_t c0, uint32_t ncols, uint64_t off, bool need_halo)
{
#pragma unroll
    for (int j = 0; j < W + 2; ++j) {
        if (FAST && j >= 3) continue;
        const bool live = FAST ? true : (c0 + j < n_cols) && ((uint32_t)j < ncols || (need_halo && (uint32_t)j < ncols + 2u));
        if (live) {
            const uint8_t JL_AS1 *base = planes + (uint64_t)(c0 + j) * 3u * plane_stride + off;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (NQ == 2) {
                    const u32x2 v = __builtin_nontemporal_load((const u32x2 JL_AS1 *)(base + (uint64_t)k * plane_stride));
                    r.d[j][k][0] = v.x;
                    r.d[j][k][1] = v.y;
                } else {
                    const u32x4 v = __builtin_nontemporal_load((const u32x4 JL_AS1 *)(base + (uint64_t)k * plane_stride));
                    r.d[j][k][0] = v.x;
                    r.d[j][k][1] = v.y;
                    r.d[j][k][NQ - 2] = v.z;
                    r.d[j][k][NQ - 1] = v.w;
                }
            }
        } else {   // code 6 everywhere: b0 = 0, b1 = b2 = 1
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                r.d[j][0][q] = 0u;
                r.d[j][1][q] = r.d[j][2][q] = 0xFFFFFFFFu;
            }
        }
    }
}

template <int W, int NQ, bool FAST>
__device__ __forceinline__ void pileup_planes_stream(const uint8_t JL_AS1 *planes, uint64_t plane_stride, uint32_t n_cols,
                                                     uint32_t n_tiles, uint32_t c0, uint32_t ncols, uint32_t startf, bool need_halo,
                                                     const uint32_t (&g)[W + 2], uint32_t (*s_hist)[64], uint32_t (*s_col)[6],
                                                     uint32_t *s_match)
{
    constexpr uint32_t TILE = plane_tile_bytes(NQ);
    const uint32_t tid = threadIdx.x;
    const bool last_lane = (tid & 63u) == 63u;
    const uint64_t lane_off = (uint64_t)tid * 4u * NQ;
    // the seed codon's bits as words: sh / sl[j] = all ones when bit 1 / bit 0 of column j's seed base is set
    uint32_t sh[W + 2], sl[W + 2];
#pragma unroll
    for (int j = 0; j < W + 2; ++j) {
        sh[j] = (g[j] & 2u) ? 0xFFFFFFFFu : 0u;
        sl[j] = (g[j] & 1u) ? 0xFFFFFFFFu : 0u;
    }

    uint32_t tile = blockIdx.y;
    ptile_regs<W, NQ> nxt;
    bool nxt_live = false;
    if (tile < n_tiles) {
        const uint64_t off = (uint64_t)tile * TILE + lane_off;
        nxt_live = off < plane_stride;
        if (nxt_live) load_ptile<W, NQ, FAST>(nxt, planes, plane_stride, n_cols, c0, ncols, off, need_halo);
    }

    while (tile < n_tiles) {
        uint32_t acc[W][6];   // n0 n1 n2 n01 n02 n12
        uint32_t match[W];
#pragma unroll
        for (int j = 0; j < W; ++j) {
            match[j] = 0;
#pragma unroll
            for (int k = 0; k < 6; ++k) acc[j][k] = 0;
        }
        uint32_t reads = 0;

        for (uint32_t it = 0; it < plane_flush_tiles(NQ) && tile < n_tiles; ++it, tile += gridDim.y) {
            const ptile_regs<W, NQ> cur = nxt;
            const bool live = nxt_live;
            const uint32_t tn = tile + gridDim.y;
            nxt_live = false;
            if (tn < n_tiles) {
                const uint64_t off = (uint64_t)tn * TILE + lane_off;
                nxt_live = off < plane_stride;
                if (nxt_live) load_ptile<W, NQ, FAST>(nxt, planes, plane_stride, n_cols, c0, ncols, off, need_halo);
            }
            if (!live) continue;
            reads += 32u * NQ;
#pragma unroll
            for (int j = 0; j < W; ++j) {
                if (FAST ? j < 3 : (uint32_t)j < ncols) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const uint32_t b0 = cur.d[j][0][q], b1 = cur.d[j][1][q], b2 = cur.d[j][2][q];
                        acc[j][0] += __popc(b0);
                        acc[j][1] += __popc(b1);
                        acc[j][2] += __popc(b2);
                        acc[j][3] += __popc(b0 & b1);
                        acc[j][4] += __popc(b0 & b2);
                        acc[j][5] += __popc(b1 & b2);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < W; ++j) {
                if (FAST ? j == 0 : (startf & (1u << j)) != 0) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const uint32_t inv = cur.d[j][2][q] | cur.d[j + 1][2][q] | cur.d[j + 2][2][q];   // some code >= 4: not in coverage
                        const uint32_t x = (cur.d[j][1][q] ^ sh[j]) | (cur.d[j][0][q] ^ sl[j]) | (cur.d[j + 1][1][q] ^ sh[j + 1]) |
                                           (cur.d[j + 1][0][q] ^ sl[j + 1]) | (cur.d[j + 2][1][q] ^ sh[j + 2]) | (cur.d[j + 2][0][q] ^ sl[j + 2]);
                        match[j] += __popc(~(x | inv));      // valid and equal to the seed codon
                        uint32_t rest = x & ~inv;            // valid codon, not the seed one
                        while (rest) {
                            const int b = __ffs((int)rest) - 1;
                            rest &= rest - 1;
                            const uint32_t idx = (((cur.d[j][1][q] >> b) & 1u) << 5) | (((cur.d[j][0][q] >> b) & 1u) << 4) |
                                                 (((cur.d[j + 1][1][q] >> b) & 1u) << 3) | (((cur.d[j + 1][0][q] >> b) & 1u) << 2) |
                                                 (((cur.d[j + 2][1][q] >> b) & 1u) << 1) | ((cur.d[j + 2][0][q] >> b) & 1u);
                            atomicAdd(&s_hist[j][idx], 1u);
                        }
                    }
                }
            }
        }

        // ---- flush this batch: the counts from the six popcount sums, 16-bit packing, DPP wave sums, LDS atomics
#pragma unroll
        for (int j = 0; j < W; ++j) {
            if (FAST && j >= 3) continue;
            const uint32_t nT = acc[j][3], nN = acc[j][4], nU = acc[j][5];
            const uint32_t nC = acc[j][0] - nT - nN, nG = acc[j][1] - nT - nU, nD = acc[j][2] - nN - nU;
            const uint32_t nA = reads - (nC + nG + nT + nD + nN + nU);
            const uint32_t p0 = wave_sum(nA | (nC << 16));
            const uint32_t p1 = wave_sum(nG | (nT << 16));
            const uint32_t p2 = wave_sum(nD | (nN << 16));
            const bool starts = FAST ? j == 0 : (startf & (1u << j)) != 0;
            uint32_t p3 = 0;
            if (starts) p3 = wave_sum(match[j]);   // (block-uniform branch)
            if (last_lane && (uint32_t)j < ncols) {
                if (p0 & 0xFFFFu) atomicAdd(&s_col[j][0], p0 & 0xFFFFu);
                if (p0 >> 16) atomicAdd(&s_col[j][1], p0 >> 16);
                if (p1 & 0xFFFFu) atomicAdd(&s_col[j][2], p1 & 0xFFFFu);
                if (p1 >> 16) atomicAdd(&s_col[j][3], p1 >> 16);
                if (p2 & 0xFFFFu) atomicAdd(&s_col[j][4], p2 & 0xFFFFu);
                if (p2 >> 16) atomicAdd(&s_col[j][5], p2 >> 16);
                if (starts && p3) atomicAdd(&s_match[j], p3);
            }
        }
    }
}

template <int W, int NQ>
__device__ __forceinline__ void pileup_planes_body(const uint8_t JL_AS1 *planes, uint64_t plane_stride, uint32_t n_cols, uint32_t n_tiles,
                                                   const uint2 JL_AS1 *chunks, const uint32_t JL_AS1 *guess32, uint32_t JL_AS1 *counts,
                                                   uint32_t JL_AS1 *hist)
{
    __shared__ uint32_t s_hist[W][64];
    __shared__ uint32_t s_col[W][6];   // A C G T - N
    __shared__ uint32_t s_match[W];
    const uint32_t tid = threadIdx.x;
    const uint2 rec = chunks[blockIdx.x];
    const uint32_t c0 = rec.x;
    const uint32_t ncols = rec.y & 15u;
    const uint32_t startf = (rec.y >> 4) & 0xFFFu;
    const bool need_halo = ((rec.y >> 16) & 1u) != 0;
    constexpr int NG = (W + 2 + 3 + 3) / 4;
    uint32_t gw[NG];
#pragma unroll
    for (int k = 0; k < NG; ++k) gw[k] = guess32[(c0 >> 2) + k];
    uint32_t g[W + 2];   // the seed base of column c0 + j (0..3)
#pragma unroll
    for (int j = 0; j < W + 2; ++j) {
        const uint32_t b = (c0 & 3u) + (uint32_t)j;
        uint32_t word = gw[0];
#pragma unroll
        for (int k = 1; k < NG; ++k)
            if ((b >> 2) == (uint32_t)k) word = gw[k];
        g[j] = (word >> (8u * (b & 3u))) & 3u;
    }
    for (uint32_t i = tid; i < W * 64; i += 256) (&s_hist[0][0])[i] = 0;
    if (tid < W * 6) (&s_col[0][0])[tid] = 0;
    if (tid < W) s_match[tid] = 0;
    __syncthreads();
    if (W == 3 && rec.y == JL_CHUNK_META(3, 1, 0))
        pileup_planes_stream<W, NQ, true>(planes, plane_stride, n_cols, n_tiles, c0, ncols, startf, need_halo, g, s_hist, s_col, s_match);
    else
        pileup_planes_stream<W, NQ, false>(planes, plane_stride, n_cols, n_tiles, c0, ncols, startf, need_halo, g, s_hist, s_col, s_match);
    __syncthreads();
    const bool excl = gridDim.y == 1;
    if (tid < W * 6) {
        const uint32_t j = tid / 6u, k = tid - j * 6u;
        const uint32_t v = s_col[j][k];
        if (j < ncols) {
            if (excl) counts[(uint64_t)(c0 + j) * 6u + k] = v;
            else if (v) atomicAdd((uint32_t *)(counts + (uint64_t)(c0 + j) * 6u + k), v);
        }
    }
    if (tid < W && (startf & (1u << tid))) {
        const uint32_t j = tid;
        const uint32_t seed = (g[j] << 4) | (g[j + 1] << 2) | g[j + 2];
        s_hist[j][seed] += s_match[j];   // reads equal to the seed codon were only counted, never binned
    }
    __syncthreads();
    for (uint32_t i = tid; i < W * 64; i += 256) {
        const uint32_t j = i >> 6;
        const uint32_t v = s_hist[j][i & 63u];
        if (startf & (1u << j)) {
            if (excl) hist[(uint64_t)(c0 + j) * 64u + (i & 63u)] = v;
            else if (v) atomicAdd((uint32_t *)(hist + (uint64_t)(c0 + j) * 64u + (i & 63u)), v);
        }
    }
}

template <int W, int NQ>
__global__ __launch_bounds__(256) void pileup_planes_kernel(const uint8_t *__restrict__ planes, uint64_t plane_stride, uint32_t n_cols,
                                                            uint32_t n_tiles, const uint2 *__restrict__ chunks,
                                                            const uint32_t *__restrict__ guess32, uint32_t *__restrict__ counts,
                                                            uint32_t *__restrict__ hist)
{
    pileup_planes_body<W, NQ>((const uint8_t JL_AS1 *)planes, plane_stride, n_cols, n_tiles, (const uint2 JL_AS1 *)chunks,
                          (const uint32_t JL_AS1 *)guess32, (uint32_t JL_AS1 *)counts, (uint32_t JL_AS1 *)hist);
}

template <int W, int NQ>
__global__ __launch_bounds__(256) void pileup_planes_group_kernel(jl_pileup_group_args args)
{
    const jl_win_pileup &w = args.w[blockIdx.z];
    if (blockIdx.x >= w.n_chunks) return;
    pileup_planes_body<W, NQ>((const uint8_t JL_AS1 *)w.msa, w.col_stride, w.n_cols, w.n_tiles, (const uint2 JL_AS1 *)w.chunks,
                          (const uint32_t JL_AS1 *)w.guess32, (uint32_t JL_AS1 *)w.counts, (uint32_t JL_AS1 *)w.hist);
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
        measure(v.x, a); measure(v.y, a); measure(v.z, a); measure(v.w, a);
        words = 4;
    }
    uint32_t nC, nG, nT, nD, nN, nU;
    solve(a, nC, nG, nT, nD, nN, nU);
    const uint32_t nA = words * 8u - (nC + nG + nT + nD + nN + nU);
    const uint32_t p0 = wave_sum(nA | (nC << 16));
    const uint32_t p1 = wave_sum(nG | (nT << 16));
    if (lane == 63) {
        uint32_t best = 0, bv = p0 & 0xFFFFu;
        if ((p0 >> 16) > bv) { bv = p0 >> 16; best = 1; }
        if ((p1 & 0xFFFFu) > bv) { bv = p1 & 0xFFFFu; best = 2; }
        if ((p1 >> 16) > bv) { bv = p1 >> 16; best = 3; }
        guess[c] = (uint8_t)best;
    }
}

struct variant_t {
    int w;
    bool pipe;
    int mode;  // bit 0: loads only (probe), bit 1: popcount measurements (probe), bit 2: non-temporal loads, bit 3: no stream (probe)
    void (*fn)(const uint8_t *, uint64_t, uint32_t, uint32_t, const uint2 *, const uint32_t *, uint32_t *, uint32_t *, const jl_callinfo *);
    void (*gfn)(jl_pileup_group_args);
    // the same with the Fisher stage in the epilogue
    void (*fn_call)(const uint8_t *, uint64_t, uint32_t, uint32_t, const uint2 *, const uint32_t *, uint32_t *, uint32_t *, const jl_callinfo *);
    void (*gfn_call)(jl_pileup_group_args);
};

// Default builds use non-temporal loads: every cell is read once, and at 2.4 GB they lift the stream from 5.6 to
// 6.3 TB/s (no difference at 150 MB).
#ifdef JL_FUSED_CALL
#define JL_V(W, P, M) {W, P, M, pileup_kernel<W, P, M, false>, pileup_group_kernel<W, P, M, false>, pileup_kernel<W, P, M, true>, pileup_group_kernel<W, P, M, true>}
#else   // the fused variants are not even compiled
#define JL_V(W, P, M) {W, P, M, pileup_kernel<W, P, M, false>, pileup_group_kernel<W, P, M, false>, nullptr, nullptr}
#endif
const variant_t kVariants[] = {
    JL_V(6, false, 4), JL_V(6, true, 4), JL_V(12, false, 4), JL_V(3, false, 4), JL_V(3, true, 4),
#ifdef JL_PILEUP_TUNING   // probes (results wrong by design except mode 0/6): JL_PILEUP_MODE selects
    JL_V(3, false, 5), JL_V(3, true, 5), JL_V(3, false, 12), JL_V(3, false, 0), JL_V(3, true, 0), JL_V(3, false, 6),
    JL_V(6, false, 5), JL_V(6, false, 0), JL_V(9, false, 4),
#endif
};

// Launch-shape switches exist for tuning builds only (make EXTRA=-DJL_TUNING); the shipped library reads no
// environment variable on its launch paths.
#ifdef JL_TUNING
int env_int(const char *name, int dflt)
{
    const char *s = getenv(name);
    return s && *s ? atoi(s) : dflt;
}
#else
constexpr int env_int(const char *, int dflt) { return dflt; }
#endif

}  // namespace

const char *jl_pileup_kernel_name(void) { return "pileup_planes_kernel"; }   // rocprofv3 prints the template arguments behind it

// the bit-plane kernels, by chunk width (build_chunks makes 3 or 6 in shipped builds)
static bool planes_usable(const jl_ctx *ctx) { return ctx->planes_valid && ctx->d_planes && (ctx->pileup_w == 3 || ctx->pileup_w == 6); }
// the load width of a launch (see plane_tile_bytes above)
static int planes_nq(const jl_ctx *ctx, bool grouped) { return ctx->pileup_w == 3 && (grouped || ctx->plane_stride >= 32768u) ? 4 : 2; }
static uint32_t planes_tiles(const jl_ctx *ctx, bool grouped)
{
    const uint32_t tile = plane_tile_bytes(planes_nq(ctx, grouped));
    return (uint32_t)((ctx->plane_stride + tile - 1) / tile);
}

void jl_launch_guess(jl_ctx *ctx, hipStream_t st)
{
    hipLaunchKernelGGL(guess_kernel, dim3(ctx->n_cols), dim3(64), 0, st, ctx->d_msa, ctx->col_stride,
                       ctx->n_cols, ctx->d_guess);
}

static int pick_variant(const jl_ctx *ctx)
{
    // The chunk table was built for ctx->pileup_w columns per chunk (capi.hip: build_chunks); the kernel
    // variant must match it.  JL_PILEUP_PIPE selects the register-prefetching build (tuning).
    const int want_w = (int)ctx->pileup_w;
    const bool want_pipe = env_int("JL_PILEUP_PIPE", JL_PILEUP_PIPE) != 0;
    const int want_mode = env_int("JL_PILEUP_MODE", 4);
    const int n = (int)(sizeof(kVariants) / sizeof(kVariants[0]));
    for (int i = 0; i < n; ++i)
        if (kVariants[i].w == want_w && kVariants[i].pipe == want_pipe && kVariants[i].mode == want_mode) return i;
    for (int i = 0; i < n; ++i)
        if (kVariants[i].w == want_w && kVariants[i].pipe == want_pipe && kVariants[i].mode == 4) return i;
    for (int i = 0; i < n; ++i)
        if (kVariants[i].w == want_w && !kVariants[i].pipe && kVariants[i].mode == 4) return i;
    return 0;
}

// occupancy query, once per variant and outside any stream capture
void jl_prepare_pileup(jl_ctx *ctx)
{
    // (slots 14 and 15 of the occupancy table: the bit-plane kernels of width 3 and 6)
    const bool planes = planes_usable(ctx);
    const int idx = planes ? (ctx->pileup_w == 3 ? 14 : 15) : pick_variant(ctx);
    if (ctx->pileup_blocks_per_cu[idx] > 0) return;
    // (the single-window launch: a short window alone reads 8 bytes a lane, a deep one 16)
    const void *fn = planes ? (ctx->pileup_w == 3 ? (planes_nq(ctx, false) == 4 ? (const void *)pileup_planes_kernel<3, 4> : (const void *)pileup_planes_kernel<3, 2>)
                                                  : (const void *)pileup_planes_kernel<6, 2>)
                            : (const void *)kVariants[idx].fn;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, 0) != hipSuccess || per_cu < 1) per_cu = 2;
    if (per_cu > 8) per_cu = 8;
    ctx->pileup_blocks_per_cu[idx] = per_cu;
}

// Launch shape: chunks x read splits.  One resident wave of blocks (CUs x the blocks the kernel's registers admit
// per CU) for short columns, four per slot for long ones; reads are split no finer than a tile.
uint32_t jl_pileup_rsplit(jl_ctx *ctx)
{
    const bool planes = planes_usable(ctx);
    const int idx = planes ? (ctx->pileup_w == 3 ? 14 : 15) : pick_variant(ctx);
    jl_prepare_pileup(ctx);
    const uint32_t n_chunks = ctx->n_chunks ? ctx->n_chunks : 1u;
    const uint32_t n_tiles = planes ? planes_tiles(ctx, false) : (uint32_t)((ctx->col_stride + JL_PILEUP_TILE_BYTES - 1) / JL_PILEUP_TILE_BYTES);
    const int per_cu = ctx->pileup_blocks_per_cu[idx];
    // long columns: several blocks per slot smooth the tail; short ones: exactly one resident wave of blocks
    const uint32_t target = 256u * (uint32_t)per_cu * (uint32_t)env_int("JL_PILEUP_WAVES", n_tiles >= 64 ? 4 : 1);
    uint32_t rsplit = target / n_chunks;  // never more blocks than resident slots: a second wave costs more than it balances
    const int forced = env_int("JL_PILEUP_RSPLIT", 0);
    if (forced > 0) rsplit = (uint32_t)forced;
    if (rsplit > n_tiles) rsplit = n_tiles;
    if (rsplit < 1) rsplit = 1;
    if (rsplit > 65535u) rsplit = 65535u;
    return rsplit;
}

// With one read split every chunk is counted by exactly one block, which stores its totals: no zeroing needed.
bool jl_pileup_needs_zero(jl_ctx *ctx) { return jl_pileup_rsplit(ctx) != 1u; }

void jl_launch_pileup(jl_ctx *ctx, hipStream_t st, bool with_call)
{
    if (planes_usable(ctx) && !with_call) {
        const uint32_t rsplit = jl_pileup_rsplit(ctx);
        const uint32_t nt = planes_tiles(ctx, false);
#define JL_LAUNCH_PLANES(W, NQ)                                                                                                          \
    hipLaunchKernelGGL((pileup_planes_kernel<W, NQ>), dim3(ctx->n_chunks, rsplit), dim3(256), 0, st, ctx->d_planes, ctx->plane_stride,   \
                       ctx->n_cols, nt, (const uint2 *)ctx->d_chunks, (const uint32_t *)ctx->d_guess, ctx->d_counts, ctx->d_hist)
        if (ctx->pileup_w == 6) JL_LAUNCH_PLANES(6, 2);
        else if (planes_nq(ctx, false) == 4) JL_LAUNCH_PLANES(3, 4);
        else JL_LAUNCH_PLANES(3, 2);
#undef JL_LAUNCH_PLANES
        return;
    }
    const int idx = pick_variant(ctx);
    const variant_t *var = &kVariants[idx];
    const uint32_t n_tiles = (uint32_t)((ctx->col_stride + JL_PILEUP_TILE_BYTES - 1) / JL_PILEUP_TILE_BYTES);
    const uint32_t rsplit = jl_pileup_rsplit(ctx);
    if (with_call && rsplit != 1u) with_call = false;   // callers check jl_pileup_can_call first; never evaluate partial histograms
    // Unused dynamic LDS caps the blocks per CU: one pileup launch then fills the chip's block slots by itself, so
    // a second batch's pileup (another stream) starts as this one's blocks retire instead of running beside it —
    // two 150 MB streams side by side reach 3.6 TB/s together, one alone 5.5 (tools_tuning/timeline.py).
    const uint32_t lds_pad = (uint32_t)env_int("JL_PILEUP_LDS_KB", 0) * 1024u;
    hipLaunchKernelGGL(with_call ? var->fn_call : var->fn, dim3(ctx->n_chunks, rsplit), dim3(256), lds_pad, st, ctx->d_msa,
                       ctx->col_stride, ctx->n_cols, n_tiles, (const uint2 *)ctx->d_chunks, (const uint32_t *)ctx->d_guess,
                       ctx->d_counts, ctx->d_hist, (const jl_callinfo *)(with_call ? ctx->d_callinfo : nullptr));
}

// The Fisher stage CAN ride in the pileup launch when one workgroup counts every chunk alone (the CALL variants).  It is
// a measured loss and off unless the library is built with -DJL_FUSED_CALL: the FP64 evaluation runs on one wave while
// the workgroup's other three hold their registers for 2-3 us of a 31 us lifetime, and it needs 137 VGPRs where the
// stream needs 92 — 229 us per 1.2 GB launch at three waves per SIMD, 270 us capped to five waves with 240 bytes of
// scratch, against 196 us for the plain kernel; call_kernel evaluates the same 8000 positions in ~5 us of its own.
bool jl_pileup_can_call(jl_ctx *ctx)
{
#ifdef JL_FUSED_CALL
    return jl_pileup_rsplit(ctx) == 1u;
#else
    (void)ctx;
    return false;
#endif
}

void jl_fill_win_pileup(jl_ctx *ctx, jl_win_pileup *w)
{
    w->msa = ctx->d_msa;
    w->col_stride = ctx->col_stride;
    w->n_cols = ctx->n_cols;
    w->n_tiles = (uint32_t)((ctx->col_stride + JL_PILEUP_TILE_BYTES - 1) / JL_PILEUP_TILE_BYTES);
    w->n_chunks = ctx->n_chunks;
    w->pad_ = 0;
    w->chunks = (const uint2 *)ctx->d_chunks;
    w->guess32 = (const uint32_t *)ctx->d_guess;
    w->counts = ctx->d_counts;
    w->hist = ctx->d_hist;
    w->ci = nullptr;
}

// Every window of a group must use the same kernel variant; each is counted by ONE block per chunk whatever its
// depth (a single run would split very long columns over several blocks).
int jl_launch_pileup_group(jl_ctx *const *ctxs, uint32_t n_win, const jl_win_pileup *h_wins, uint32_t max_chunks, hipStream_t st,
                           bool with_call)
{
    const int idx = pick_variant(ctxs[0]);
    if (n_win > JL_GROUP_WINDOWS_MAX) return JL_ERR_ARG;
    for (uint32_t k = 0; k < n_win; ++k)
        if (pick_variant(ctxs[k]) != idx) return JL_ERR_ARG;
    jl_pileup_group_args args;
    memset(&args, 0, sizeof args);
    memcpy(args.w, h_wins, sizeof(jl_win_pileup) * n_win);
    // one kernel for the whole launch: the bit planes when every window has them, else the nibbles for all
    bool planes = !with_call;
    for (uint32_t k = 0; k < n_win; ++k) planes = planes && planes_usable(ctxs[k]);
    for (uint32_t k = 0; k < n_win; ++k) {
        args.w[k].ci = with_call ? ctxs[k]->d_callinfo : nullptr;
        if (planes) {
            args.w[k].msa = ctxs[k]->d_planes;
            args.w[k].col_stride = ctxs[k]->plane_stride;
            args.w[k].n_tiles = planes_tiles(ctxs[k], true);
        } else {
            args.w[k].msa = ctxs[k]->d_msa;
            args.w[k].col_stride = ctxs[k]->col_stride;
            args.w[k].n_tiles = (uint32_t)((ctxs[k]->col_stride + JL_PILEUP_TILE_BYTES - 1) / JL_PILEUP_TILE_BYTES);
        }
    }
    if (planes && ctxs[0]->pileup_w == 3) hipLaunchKernelGGL((pileup_planes_group_kernel<3, 4>), dim3(max_chunks, 1, n_win), dim3(256), 0, st, args);
    else if (planes) hipLaunchKernelGGL((pileup_planes_group_kernel<6, 2>), dim3(max_chunks, 1, n_win), dim3(256), 0, st, args);
    else hipLaunchKernelGGL(with_call ? kVariants[idx].gfn_call : kVariants[idx].gfn, dim3(max_chunks, 1, n_win), dim3(256), 0, st, args);
    return JL_OK;
}
