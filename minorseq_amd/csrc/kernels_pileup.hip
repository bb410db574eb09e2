// kernels_pileup.hip — column pileup + per-codon histogram over the resident bit planes.
//
// Behaviour implemented: doc/JULIET.md:99-100 (per-column counts of A C G T - N), :21-27/:94-98
// (codon-wise counting, deletions ignored), :256-259 (N does not count towards the coverage);
// details in docs/SPEC.md §2-3.
//
// Shape of the work.  The matrix is n_cols columns of three BIT PLANES each (plane k = bit k of every read's 3-bit code,
// reads in bit order; jl_internal.h): every cell is read exactly once — algorithmic bytes = n_reads * n_cols * 3 / 8 — so
// the kernel is an HBM stream and one instruction handles 32 reads:
//   * a lane loads 16 (or 8) bytes of each plane of each of its chunk's columns (a wave = 1 KiB contiguous per load),
//   * column counts are six popcounts per 32 reads (b0, b1, b2, b0&b1, b0&b2, b1&b2; code 7 does not occur),
//   * codon histograms are counted against a per-column seed base: six xors with the seed's bits, or-ed with the three b2
//     words (a code >= 4 anywhere: no codon): reads equal to the seed codon are ONE popcount per 32 reads (the contended
//     "major codon" bin never sees an atomic), the rare valid mismatches take an LDS atomic each.
//     The seed only steers which bin is counted the fast way — any seed gives the same histogram.
// A block owns one chunk of <= W consecutive columns from a host-built table (capi.hip: build_chunks) and a strided set of
// read tiles.  Chunks start on codon boundaries, so stretches that are locally single-frame need no halo; only where a
// codon of the chunk reaches past its last column are the next two columns loaded as well.  Loads are non-temporal (every
// cell is read once) and the next tile is prefetched into a second register set while the current one is counted.
// Per-lane counters are packed two per register, wave-reduced by DPP and flushed to LDS before 16-bit fields can overflow.
// A block that counts its chunk alone (gridDim.y = 1) stores the totals; when the reads of a long column are split over
// several blocks they go to HBM with integer atomics, which commute, so results are bit-exact and order-independent either
// way.  There is no reuse between blocks (halo columns aside), so the block -> XCD mapping does not matter here.
// pileup_planes_group_kernel runs the same body for several windows in one launch (blockIdx.z = window).
// (Rounds 1-2 counted a 4-bit-per-cell matrix with v_dot8 measurements; the planes replaced it as THE resident format in
// round 4 — every producer writes them directly — and that kernel is gone.)
#include <stdlib.h>
#include <string.h>

#include "call_eval.h"
#include "jl_internal.h"

namespace {


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

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// chunk record (host-built, capi.hip: build_chunks): x = first column, y = own columns (bits 0-3) | codon-start
// flags of the own columns (bits 4-15) | "a codon reaches into the next two columns" (bit 16)
#define JL_CHUNK_META(ncols, startf, halo) ((uint32_t)(ncols) | ((uint32_t)(startf) << 4) | ((uint32_t)(halo) << 16))

// ---------------------------------------------------------------------------------------- counting from the bit planes
// The resident matrix (jl_ctx::d_msa): per column three planes — bit k of every read's code, reads in bit order — so a cell
// costs 3 bits of HBM traffic and one instruction handles 32 reads:
//   * column counts: six popcounts per 32 reads (b0, b1, b2, b0&b1, b0&b2, b1&b2; code 7 does not occur), from which
//     T = n01, N = n02, uncovered = n12, C = n0 - n01 - n02, G = n1 - n01 - n12, '-' = n2 - n02 - n12, A = the rest;
//   * codons against the seed: six xors against the seed's bits (block-uniform all-ones / all-zero words), or-ed together
//     with the three b2 words (a code >= 4 anywhere: not a codon): reads equal to the seed codon are a popcount, the rare
//     valid mismatches are walked bit by bit into the LDS histogram as before.
// A lane takes 8 or 16 bytes = 64 / 128 reads of each plane per tile; the next tile's words are prefetched into a second
// register set while the current one is counted.
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

// FOLD (round 6): the Fisher stage of the chunk's codons in the workgroup that has just counted them — the histogram is still in
// LDS — instead of a call launch that reads it back from HBM (8 us of a 64 us window, 20 of the 72 us an eight-window launch's
// tail exposes).  Only where ONE workgroup counts a chunk (gridDim.y = 1).  Wave j & 3 takes the codon that begins at the chunk's
// column j — the positions at that column from the host's lists (jl_win_fold) — behind the kernel's last barrier: the other
// waves store the histogram and leave, so what the evaluation holds for its 2-3 us is one wave's registers, not the workgroup's.
template <int W, int NQ, bool FOLD>
__device__ __forceinline__ void pileup_planes_body(const uint8_t JL_AS1 *planes, uint64_t plane_stride, uint32_t n_cols, uint32_t n_tiles,
                                                   const uint2 JL_AS1 *chunks, const uint32_t JL_AS1 *guess32, uint32_t JL_AS1 *counts,
                                                   uint32_t JL_AS1 *hist, const jl_win_fold *fold)
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
    if (FOLD) {
        const jl_win_call &c = fold->c;
        if (blockIdx.x == 0 && tid == 0 && c.meta) {   // counters of the phasing launch that follows on the stream (call_kernel's first block did this)
            c.meta->n_occupied = 0;
            c.meta->overflow = 0;
            jl_phase_summary z = {0, 0, 0, 0, 0, 0, 0, 0};
            c.meta->summary = z;
        }
        const uint32_t wid = tid >> 6, lane = tid & 63u;
#pragma unroll
        for (int j = 0; j < W; ++j) {
            if (((uint32_t)j & 3u) != wid || !(startf & (1u << j))) continue;      // (wave-uniform)
            const uint32_t col = c0 + (uint32_t)j;
            const uint32_t h = s_hist[j][lane];
            for (uint32_t p = fold->col_head[col]; p != 0xFFFFFFFFu; p = fold->pos_next[p])
                jl_call_position<false>(c.A, p, col, h, c.pos_refcfg[p], c.pos_gene[p], c.pos_codon[p], c.drm, c.called, c.staged);
        }
    }
}

template <int W, int NQ>
__global__ __launch_bounds__(256) void pileup_planes_kernel(const uint8_t *__restrict__ planes, uint64_t plane_stride, uint32_t n_cols,
                                                            uint32_t n_tiles, const uint2 *__restrict__ chunks,
                                                            const uint32_t *__restrict__ guess32, uint32_t *__restrict__ counts,
                                                            uint32_t *__restrict__ hist)
{
    pileup_planes_body<W, NQ, false>((const uint8_t JL_AS1 *)planes, plane_stride, n_cols, n_tiles, (const uint2 JL_AS1 *)chunks,
                                     (const uint32_t JL_AS1 *)guess32, (uint32_t JL_AS1 *)counts, (uint32_t JL_AS1 *)hist, nullptr);
}

template <int W, int NQ>
__global__ __launch_bounds__(256) void pileup_planes_group_kernel(jl_pileup_group_args args)
{
    const jl_win_pileup &w = args.w[blockIdx.z];
    if (blockIdx.x >= w.n_chunks) return;
    pileup_planes_body<W, NQ, false>((const uint8_t JL_AS1 *)w.msa, w.plane_stride, w.n_cols, w.n_tiles, (const uint2 JL_AS1 *)w.chunks,
                                     (const uint32_t JL_AS1 *)w.guess32, (uint32_t JL_AS1 *)w.counts, (uint32_t JL_AS1 *)w.hist, nullptr);
}

// ... and with the Fisher stage folded in (runs whose chunks are counted by one workgroup each; gridDim.y = 1)
template <int W, int NQ>
__global__ __launch_bounds__(256) void pileup_fold_kernel(jl_win_pileup w, jl_win_fold f)
{
    pileup_planes_body<W, NQ, true>((const uint8_t JL_AS1 *)w.msa, w.plane_stride, w.n_cols, w.n_tiles, (const uint2 JL_AS1 *)w.chunks,
                                    (const uint32_t JL_AS1 *)w.guess32, (uint32_t JL_AS1 *)w.counts, (uint32_t JL_AS1 *)w.hist, &f);
}

template <int W, int NQ>
__global__ __launch_bounds__(256) void pileup_fold_group_kernel(jl_pileup_fold_group_args args)
{
    const jl_win_pileup &w = args.w[blockIdx.z];
    if (blockIdx.x >= w.n_chunks) return;
    pileup_planes_body<W, NQ, true>((const uint8_t JL_AS1 *)w.msa, w.plane_stride, w.n_cols, w.n_tiles, (const uint2 JL_AS1 *)w.chunks,
                                    (const uint32_t JL_AS1 *)w.guess32, (uint32_t JL_AS1 *)w.counts, (uint32_t JL_AS1 *)w.hist, &args.f[blockIdx.z]);
}

// Seed base per column for majority-codon mode: majority base among the first 2048 reads of the column.
// (Any value is correct; a good seed keeps the codon compare on its fast path.)
__global__ __launch_bounds__(64) void guess_kernel(const uint8_t *__restrict__ planes, uint64_t plane_stride,
                                                   uint32_t n_cols, uint8_t *__restrict__ guess)
{
    const uint32_t c = blockIdx.x;
    if (c >= n_cols) return;
    const uint32_t lane = threadIdx.x;
    uint32_t nA = 0, nC = 0, nG = 0, nT = 0;
    const uint64_t off = (uint64_t)lane * 4u;   // 64 lanes x 32 reads
    if (off < plane_stride) {
        const uint8_t *base = planes + (uint64_t)c * 3u * plane_stride + off;
        const uint32_t b0 = *reinterpret_cast<const uint32_t *>(base);
        const uint32_t b1 = *reinterpret_cast<const uint32_t *>(base + plane_stride);
        const uint32_t b2 = *reinterpret_cast<const uint32_t *>(base + 2u * plane_stride);
        nA = __popc(~b2 & ~b1 & ~b0);
        nC = __popc(~b2 & ~b1 & b0);
        nG = __popc(~b2 & b1 & ~b0);
        nT = __popc(~b2 & b1 & b0);
    }
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

// the load width of a launch (see plane_tile_bytes above); chunk tables are 3 or 6 columns wide (build_chunks)
static int planes_nq(const jl_ctx *ctx, bool grouped) { return ctx->pileup_w == 3 && (grouped || ctx->plane_stride >= 32768u) ? 4 : 2; }
static uint32_t planes_tiles(const jl_ctx *ctx, bool grouped)
{
    const uint32_t tile = plane_tile_bytes(planes_nq(ctx, grouped));
    return (uint32_t)((ctx->plane_stride + tile - 1) / tile);
}
// slot of the single-window kernel in the context's occupancy table
static int planes_slot(const jl_ctx *ctx) { return ctx->pileup_w == 3 ? (planes_nq(ctx, false) == 4 ? 0 : 1) : 2; }

void jl_launch_guess(jl_ctx *ctx, hipStream_t st)
{
    hipLaunchKernelGGL(guess_kernel, dim3(ctx->n_cols), dim3(64), 0, st, ctx->d_msa, ctx->plane_stride, ctx->n_cols, ctx->d_guess);
}

// occupancy query, once per kernel and outside any stream capture
void jl_prepare_pileup(jl_ctx *ctx)
{
    const int idx = planes_slot(ctx);
    if (ctx->pileup_blocks_per_cu[idx] > 0) return;
    // (the single-window launch: a short window alone reads 8 bytes a lane, a deep one 16)
    const void *fn = idx == 0 ? (const void *)pileup_planes_kernel<3, 4> : idx == 1 ? (const void *)pileup_planes_kernel<3, 2>
                                                                                    : (const void *)pileup_planes_kernel<6, 2>;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, 0) != hipSuccess || per_cu < 1) per_cu = 2;
    if (per_cu > 8) per_cu = 8;
    ctx->pileup_blocks_per_cu[idx] = per_cu;
}

// Launch shape: chunks x read splits.  One resident wave of blocks (CUs x the blocks the kernel's registers admit
// per CU) for short columns, four per slot for long ones; reads are split no finer than a tile.
uint32_t jl_pileup_rsplit(jl_ctx *ctx)
{
    jl_prepare_pileup(ctx);
    const uint32_t n_chunks = ctx->n_chunks ? ctx->n_chunks : 1u;
    const uint32_t n_tiles = planes_tiles(ctx, false);
    const int per_cu = ctx->pileup_blocks_per_cu[planes_slot(ctx)];
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

void jl_launch_pileup(jl_ctx *ctx, hipStream_t st)
{
    const uint32_t rsplit = jl_pileup_rsplit(ctx);
    const uint32_t nt = planes_tiles(ctx, false);
#define JL_LAUNCH_PLANES(W, NQ)                                                                                                       \
    hipLaunchKernelGGL((pileup_planes_kernel<W, NQ>), dim3(ctx->n_chunks, rsplit), dim3(256), 0, st, ctx->d_msa, ctx->plane_stride,   \
                       ctx->n_cols, nt, (const uint2 *)ctx->d_chunks, (const uint32_t *)ctx->d_guess, ctx->d_counts, ctx->d_hist)
    if (ctx->pileup_w == 6) JL_LAUNCH_PLANES(6, 2);
    else if (planes_nq(ctx, false) == 4) JL_LAUNCH_PLANES(3, 4);
    else JL_LAUNCH_PLANES(3, 2);
#undef JL_LAUNCH_PLANES
}

// (JL_NO_FOLD_CALL: the separate call launch instead, for A/B measurements; read once, like JL_NO_GRAPH)
bool jl_fold_enabled(void)
{
    static const bool on = !getenv("JL_NO_FOLD_CALL");
    return on;
}
bool jl_pileup_can_fold(jl_ctx *ctx) { return jl_fold_enabled() && ctx->P != 0 && jl_pileup_rsplit(ctx) == 1u; }

void jl_fill_win_fold(jl_ctx *ctx, const jl_win_call *call, jl_win_fold *f)
{
    f->c = *call;
    f->col_head = ctx->d_col_head;
    f->pos_next = ctx->d_pos_next;
}

void jl_launch_pileup_fold(jl_ctx *ctx, hipStream_t st, const jl_win_call *call)
{
    jl_win_pileup w;
    jl_win_fold f;
    w.msa = ctx->d_msa;
    w.plane_stride = ctx->plane_stride;
    w.n_cols = ctx->n_cols;
    w.n_tiles = planes_tiles(ctx, false);
    w.n_chunks = ctx->n_chunks;
    w.pad_ = 0;
    w.chunks = (const uint2 *)ctx->d_chunks;
    w.guess32 = (const uint32_t *)ctx->d_guess;
    w.counts = ctx->d_counts;
    w.hist = ctx->d_hist;
    jl_fill_win_fold(ctx, call, &f);
    if (ctx->pileup_w == 6) hipLaunchKernelGGL((pileup_fold_kernel<6, 2>), dim3(ctx->n_chunks, 1), dim3(256), 0, st, w, f);
    else if (planes_nq(ctx, false) == 4) hipLaunchKernelGGL((pileup_fold_kernel<3, 4>), dim3(ctx->n_chunks, 1), dim3(256), 0, st, w, f);
    else hipLaunchKernelGGL((pileup_fold_kernel<3, 2>), dim3(ctx->n_chunks, 1), dim3(256), 0, st, w, f);
}

int jl_launch_pileup_fold_group(jl_ctx *const *ctxs, uint32_t n_win, const jl_win_pileup *h_wins, const jl_win_fold *h_fold, uint32_t max_chunks, hipStream_t st)
{
    if (n_win == 0 || n_win > JL_GROUP_MAX) return JL_ERR_ARG;
    for (uint32_t k = 1; k < n_win; ++k)
        if (ctxs[k]->pileup_w != ctxs[0]->pileup_w) return JL_ERR_ARG;
    jl_pileup_fold_group_args args;
    memset(&args, 0, sizeof args);
    memcpy(args.w, h_wins, sizeof(jl_win_pileup) * n_win);
    memcpy(args.f, h_fold, sizeof(jl_win_fold) * n_win);
    if (ctxs[0]->pileup_w == 3) hipLaunchKernelGGL((pileup_fold_group_kernel<3, 4>), dim3(max_chunks, 1, n_win), dim3(256), 0, st, args);
    else hipLaunchKernelGGL((pileup_fold_group_kernel<6, 2>), dim3(max_chunks, 1, n_win), dim3(256), 0, st, args);
    return JL_OK;
}

void jl_fill_win_pileup(jl_ctx *ctx, jl_win_pileup *w)
{
    w->msa = ctx->d_msa;
    w->plane_stride = ctx->plane_stride;
    w->n_cols = ctx->n_cols;
    w->n_tiles = planes_tiles(ctx, true);
    w->n_chunks = ctx->n_chunks;
    w->pad_ = 0;
    w->chunks = (const uint2 *)ctx->d_chunks;
    w->guess32 = (const uint32_t *)ctx->d_guess;
    w->counts = ctx->d_counts;
    w->hist = ctx->d_hist;
}

// Every window of a group must use the same chunk width; each is counted by ONE block per chunk whatever its
// depth (a single run would split very long columns over several blocks).
int jl_launch_pileup_group(jl_ctx *const *ctxs, uint32_t n_win, const jl_win_pileup *h_wins, uint32_t max_chunks, hipStream_t st)
{
    if (n_win == 0 || n_win > JL_GROUP_WINDOWS_MAX) return JL_ERR_ARG;
    for (uint32_t k = 1; k < n_win; ++k)
        if (ctxs[k]->pileup_w != ctxs[0]->pileup_w) return JL_ERR_ARG;
    jl_pileup_group_args args;
    memset(&args, 0, sizeof args);
    memcpy(args.w, h_wins, sizeof(jl_win_pileup) * n_win);
    if (ctxs[0]->pileup_w == 3) hipLaunchKernelGGL((pileup_planes_group_kernel<3, 4>), dim3(max_chunks, 1, n_win), dim3(256), 0, st, args);
    else hipLaunchKernelGGL((pileup_planes_group_kernel<6, 2>), dim3(max_chunks, 1, n_win), dim3(256), 0, st, args);
    return JL_OK;
}
