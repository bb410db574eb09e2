// kernels_util.hip — device-side producers of the resident matrix (three bit planes per column, jl_internal.h); every one
// of them writes the planes directly:
//   synth_kernel              synthetic aligned CCS reads (jl_synth.h)
//   pack_rows_kernel          by-row uint8 codes (jl_msa_pack_rows)
//   nibbles_to_planes_kernel  the interchange format of jl_msa_upload, validated on the way (planes_to_nibbles_kernel: jl_msa_download)
//   expand_rows_kernel + transpose_rows_kernel   aligned BAM records
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


// ---------------------------------------------------------------------------------------- record ingest
// Aligned records -> the resident planes: doc/JULIET.md:26-27 (insertions dropped, deletions '-'), :53 (PacBio cigars
// = X I D S H N; M rejected on the host), :256-259 (filtered base = N).  Two streaming kernels:
//   expand_rows_kernel   ONE WAVE per read.  The wave scans the read's cigar once into LDS (prefix sums of reference and
//                        query lengths), then every lane produces 8 consecutive columns = one dword of the read's row
//                        in a by-row nibble matrix: a binary search finds the op that covers its first column; when
//                        all 8 columns lie inside one op (runs of '=' are tens of columns long) they are taken at
//                        once — eight packed BAM bases fetched as two dwords, converted nibble-parallel — otherwise
//                        column by column.  No divergence over reads: a wave only ever looks at one cigar.
//   transpose_rows_kernel  by-row nibbles -> the planes through a 256-read x 256-column LDS tile: a thread takes 32 reads x
//                        8 columns out of it, transposes four 8 x 8 nibble blocks in registers and splits the codes into
//                        their three bits — a dword of each plane of each of the 8 columns, 32-byte runs per quad of lanes.
// (The first build walked one read per LANE while the wave swept the columns: some lane of 64 changed its op at
// nearly every column and the kernel was bound by the cursor's instructions under divergence, 0.29 TB/s.)
// Reads past n_reads (the padding of a column up to its 128-byte stride) and columns outside a read's span are 'not covered'.
constexpr uint32_t kExpandOps = 256;    // cigar ops of a read held in LDS at a time; longer cigars go through in pieces
constexpr uint32_t kExpandSeqDw = 1024; // dwords of a read's packed bases staged in LDS (8192 bases); longer reads load from HBM

__device__ __forceinline__ bool cig_ref(uint32_t op) { return op == 2u || op == 3u || op == 7u || op == 8u; }    // D N = X
__device__ __forceinline__ bool cig_query(uint32_t op) { return op == 1u || op == 4u || op == 7u || op == 8u; }  // I S = X

// eight BAM base codes (nt16: A=1 C=2 G=4 T=8, everything else ambiguous) -> symbol codes 0..3 / 5, nibble-parallel
__device__ __forceinline__ uint32_t nt16_to_sym8(uint32_t w)
{
    const uint32_t m = 0x11111111u;
    const uint32_t b0 = w & m, b1 = (w >> 1) & m, b2 = (w >> 2) & m, b3 = (w >> 3) & m;
    const uint32_t cnt = b0 + b1 + b2 + b3;          // set bits per nibble, 0..4
    const uint32_t idx = b1 + 2u * b2 + 3u * b3;     // one-hot -> 0..3
    const uint32_t t = cnt ^ m;                      // non-zero where the nibble is not one-hot
    const uint32_t bad = (t | (t >> 1) | (t >> 2)) & m;
    return (idx & ~(bad * 15u)) | (bad * 5u);
}

__global__ __launch_bounds__(256) void expand_rows_kernel(uint64_t r0, uint64_t n_batch, uint64_t n_reads, uint32_t n_cols,
                                                           uint32_t win_begin, const int32_t *__restrict__ pos,
                                                           const uint32_t *__restrict__ cigar,
                                                           const uint64_t *__restrict__ cig_off,
                                                           const uint8_t *__restrict__ seq4,
                                                           const uint64_t *__restrict__ seq_off,
                                                           const uint8_t *__restrict__ qual,
                                                           const uint64_t *__restrict__ qual_off, uint32_t min_qv,
                                                           uint32_t *__restrict__ rows4, uint32_t row_dwords,
                                                           uint32_t ops_cap, uint32_t seq_cap)
{
    // LDS per wave, sized by the launch for the longest cigar (up to kExpandOps) and read (up to kExpandSeqDw dwords) of
    // the input: short reads leave room for every wave slot of the CU
    extern __shared__ uint32_t s_dyn[];
    const uint32_t wid = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t *s_base = s_dyn + (size_t)wid * (2u * ops_cap + seq_cap + 2u);
    // RUNS of the cigar: '=' and 'X' alternate in a PacBio cigar while reference and query advance together, so a
    // stretch of them is ONE run of aligned bases; D and N are runs of their own; I / S / H / P only end a run.  A CCS
    // read is a handful of runs of hundreds of columns, where its cigar has an op every few dozen.
    uint32_t *s_rbeg = s_base;                    // [ops_cap] reference offset (relative to the read's first base) of run i
    uint32_t *s_rq = s_base + ops_cap;            // [ops_cap] its first query offset (28 bits) | kind << 28 (1 bases, 2 '-', 3 skip)
    uint32_t *s_seq = s_base + 2u * ops_cap;      // [seq_cap + 2] the read's packed bases from the aligned dword that holds its first one
    const uint64_t b = (uint64_t)blockIdx.x * 4u + wid;   // read of this wave, within the batch
    if (b >= n_batch) return;
    const uint64_t r = r0 + b;
    uint32_t *row = rows4 + b * (uint64_t)row_dwords;
    if (r >= n_reads) {   // padding read: not covered anywhere
        for (uint32_t d = lane; d < row_dwords; d += 64u) row[d] = 0x66666666u;
        return;
    }
    const uint64_t c_beg = cig_off[r], c_end = cig_off[r + 1];
    const uint64_t so = seq_off[r];
    const uint64_t n_bases = (seq_off[r + 1] - so) * 2u;
    const uint8_t *ql = (qual && min_qv) ? qual + qual_off[r] : nullptr;
    const uint64_t n_qual = ql ? qual_off[r + 1] - qual_off[r] : 0u;
    const int64_t rel0 = (int64_t)win_begin - (int64_t)pos[r];   // reference offset (relative to the read) of window column 0
    const bool single = c_end - c_beg <= ops_cap;
    // the first piece of the cigar: all loads issued before anything waits for one
    uint32_t cw_first[kExpandOps / 64u];
#pragma unroll
    for (uint32_t i = 0; i < kExpandOps / 64u; ++i) {
        const uint64_t k = c_beg + i * 64u + lane;
        cw_first[i] = (i * 64u < ops_cap && k < c_end) ? cigar[k] : 0u;
    }
    // ---- the read's bases into LDS: every later fetch is an LDS access, not a dependent trip to HBM
    const uint64_t so_al = so & ~(uint64_t)3;                      // aligned start
    const uint32_t seq_dw = (uint32_t)((seq_off[r + 1] - so_al + 3u) / 4u) + 1u;   // the arrays are padded by 16 bytes
    const bool seq_lds = seq_dw <= seq_cap + 2u;
    const uint32_t *seq_g = reinterpret_cast<const uint32_t *>(seq4 + so_al);
    if (seq_lds)
        for (uint32_t i = lane; i < seq_dw; i += 64u) s_seq[i] = seq_g[i];
    const uint32_t lb0 = (uint32_t)(so - so_al);                   // byte of the first base within the staged dwords
    if (!single)
        for (uint32_t d = lane; d < row_dwords; d += 64u) row[d] = 0x66666666u;
    uint32_t ref_carry = 0, q_carry = 0;
    for (uint64_t base = c_beg; base < c_end || base == c_beg; base += ops_cap) {
        const uint32_t m = (uint32_t)min<uint64_t>(ops_cap, c_end - base);
        const uint32_t ref_first = ref_carry;
        // ---- prefix sums of this piece of the cigar, and its runs
        uint32_t n_runs = 0, prev_kind = 0;   // kind of the op before this chunk's first (a piece starts a new run)
        for (uint32_t k0 = 0; k0 < m; k0 += 64u) {
            const uint32_t k = k0 + lane;
            uint32_t cw = 0u;
            if (base == c_beg) {
#pragma unroll
                for (uint32_t i = 0; i < kExpandOps / 64u; ++i)
                    if (k0 == i * 64u) cw = cw_first[i];
            } else if (k < m) cw = cigar[base + k];
            const uint32_t op = cw & 15u, len = cw >> 4;
            const bool live = k < m;
            const uint32_t rl = (live && cig_ref(op)) ? len : 0u, qlx = (live && cig_query(op)) ? len : 0u;
            uint32_t ri = rl, qi = qlx;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t ur = __shfl_up(ri, o, 64), uq = __shfl_up(qi, o, 64);
                if ((int)lane >= o) { ri += ur; qi += uq; }
            }
            const uint32_t kind = !live ? 0u : (op == 7u || op == 8u) ? 1u : op == 2u ? 2u : op == 3u ? 3u : 0u;
            uint32_t before = __shfl_up(kind, 1, 64);
            if (lane == 0) before = prev_kind;
            const bool starts = kind != 0u && len != 0u && !(kind == 1u && before == 1u);
            const uint64_t bal = __ballot(starts);
            if (starts) {
                const uint32_t idx = n_runs + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                s_rbeg[idx] = ref_carry + ri - rl;
                s_rq[idx] = ((q_carry + qi - qlx) & 0x0FFFFFFFu) | (kind << 28);
            }
            n_runs += (uint32_t)__popcll(bal);
            prev_kind = __shfl(kind, 63, 64);
            ref_carry += __shfl(ri, 63, 64);
            q_carry += __shfl(qi, 63, 64);
        }
        __builtin_amdgcn_wave_barrier();
        const uint32_t ref_last = ref_carry;   // this piece covers reference offsets [ref_first, ref_last)
        // eight bases starting at query offset q, as nt16 codes in nibble order
        auto fetch8 = [&](uint64_t q) -> uint32_t {
            const uint32_t lb = lb0 + (uint32_t)(q >> 1);          // byte within the staged dwords
            const uint32_t dw = lb >> 2;
            uint64_t v;
            if (seq_lds) v = (uint64_t)s_seq[dw] | ((uint64_t)s_seq[dw + 1u] << 32);
            else v = (uint64_t)seq_g[dw] | ((uint64_t)seq_g[dw + 1u] << 32);
            v >>= 8u * (lb & 3u);
            v = ((v & 0x0F0F0F0F0F0F0F0Full) << 4) | ((v >> 4) & 0x0F0F0F0F0F0F0F0Full);   // base order = nibble order
            v >>= 4u * (uint32_t)(q & 1u);
            return (uint32_t)v;
        };
        // ---- the row, one dword (8 columns) per lane and step
        for (uint32_t d = lane; d < row_dwords; d += 64u) {
            const int64_t x0 = rel0 + (int64_t)d * 8;
            // columns of this dword the piece can say something about
            if (!single && (x0 + 8 <= (int64_t)ref_first || x0 >= (int64_t)ref_last)) continue;
            uint32_t out = single ? 0x66666666u : row[d];
            if (x0 + 8 > (int64_t)ref_first && x0 < (int64_t)ref_last && n_runs) {
                const uint32_t xs = x0 > (int64_t)ref_first ? (uint32_t)x0 : ref_first;   // first offset to look up
                // last run that starts at or before xs
                uint32_t lo = 0, hi = n_runs;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (s_rbeg[mid] <= xs) lo = mid + 1;
                    else hi = mid;
                }
                uint32_t i = lo ? lo - 1u : 0u;
                // (scratch dwords wholly past the last column hold no column at all)
                const uint32_t ncol = (uint64_t)d * 8u + 8u <= n_cols ? 8u : ((uint64_t)d * 8u < n_cols ? n_cols - d * 8u : 0u);
                uint32_t j = (uint32_t)((int64_t)xs - x0);   // first column of the dword this piece covers
                if (s_rbeg[i] > xs) j = ncol;                // (only a piece that begins with ops without reference)
                while (j < ncol && i < n_runs) {
                    const uint32_t x = (uint32_t)(x0 + (int64_t)j);
                    const uint32_t rend = i + 1u < n_runs ? s_rbeg[i + 1u] : ref_last;
                    if (x >= rend) { ++i; continue; }
                    const uint32_t rq = s_rq[i], kind = rq >> 28;
                    const uint32_t cnt = min(ncol - j, rend - x);            // columns of this run inside the dword
                    const uint32_t keep = cnt >= 8u ? 0xFFFFFFFFu : ((1u << (4u * cnt)) - 1u);
                    uint32_t sym;
                    if (kind == 2u) sym = 0x44444444u;
                    else if (kind == 3u) sym = 0x66666666u;
                    else {
                        const uint64_t q = (uint64_t)(rq & 0x0FFFFFFFu) + (x - s_rbeg[i]);
                        sym = nt16_to_sym8(fetch8(q));
                        if (q + cnt > n_bases)   // malformed input: never past the read's own bases
                            for (uint32_t t = 0; t < cnt; ++t)
                                if (q + t >= n_bases) sym = (sym & ~(15u << (4u * t))) | (5u << (4u * t));
                        if (ql)
                            for (uint32_t t = 0; t < cnt; ++t) {
                                const uint8_t qv = ql[min(q + t, n_qual ? n_qual - 1u : 0u)];
                                if (qv != 0xFFu && qv < min_qv) sym = (sym & ~(15u << (4u * t))) | ((uint32_t)JL_SYM_MASK << (4u * t));
                            }
                    }
                    const uint32_t mask = keep << (4u * j);
                    out = (out & ~mask) | ((sym << (4u * j)) & mask);
                    j += cnt;
                }
            }
            row[d] = out;
        }
        __builtin_amdgcn_wave_barrier();
        if (c_end == c_beg) break;
    }
}

// by-row nibbles of a batch of reads -> the column-packed matrix.  Tile = 128 reads x 256 columns through LDS.
// 8 x 8 nibbles held as 8 dwords (row i = m[i], element j at bits 4j) -> their transpose
__device__ __forceinline__ void transpose_nibbles_8x8(uint32_t (&m)[8])
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t a = m[i], b = m[i + 4];
        m[i] = (a & 0x0000FFFFu) | (b << 16);
        m[i + 4] = (a >> 16) | (b & 0xFFFF0000u);
    }
#pragma unroll
    for (int h = 0; h < 8; h += 4)
#pragma unroll
        for (int i = h; i < h + 2; ++i) {
            const uint32_t a = m[i], b = m[i + 2];
            m[i] = (a & 0x00FF00FFu) | ((b & 0x00FF00FFu) << 8);
            m[i + 2] = ((a >> 8) & 0x00FF00FFu) | (b & 0xFF00FF00u);
        }
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const uint32_t a = m[i], b = m[i + 1];
        m[i] = (a & 0x0F0F0F0Fu) | ((b & 0x0F0F0F0Fu) << 4);
        m[i + 1] = ((a >> 4) & 0x0F0F0F0Fu) | (b & 0xF0F0F0F0u);
    }
}

// 32 reads x 8 columns, R[i] = the 8 codes (nibbles) of read i -> out[j][k] = plane k of column j, bit i = read i.
// Four 8 x 8 nibble transposes — block g holds reads g, g + 4, ..., g + 28, so that after it nibble n of M[g][j] is read
// 4 n + g at column j — then bit k of the four blocks' nibbles interleaves into the 32 read bits with four and-or steps.
__device__ __forceinline__ void nibble_rows_to_plane_words(const uint32_t (&R)[32], uint32_t (&out)[8][3])
{
    uint32_t M[4][8];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int i = 0; i < 8; ++i) M[g][i] = R[4 * i + g];
        transpose_nibbles_8x8(M[g]);
    }
    constexpr uint32_t m = 0x11111111u;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (uint32_t k = 0; k < 3u; ++k)
            out[j][k] = ((M[0][j] >> k) & m) | (((M[1][j] >> k) & m) << 1) | (((M[2][j] >> k) & m) << 2) | (((M[3][j] >> k) & m) << 3);
}

// LDS tile of 256 reads x 32 dwords: dword d of read r at (r & 7) * kTrOct + (r >> 3) * 33 + d.  Rows go in with one read
// per half-wave (banks d); a thread takes 32 reads of one dword out.
constexpr uint32_t kTrOct = 1060;
static_assert(kTrOct % 32u == 4u && kTrOct >= 32u * 33u, "LDS tile layout");

__global__ __launch_bounds__(256) void transpose_rows_kernel(const uint32_t *__restrict__ rows4, uint32_t row_dwords,
                                                              uint64_t r0, uint64_t n_batch, uint32_t n_cols,
                                                              uint8_t *__restrict__ msa, uint64_t plane_stride)
{
    __shared__ uint32_t s_t[8u * kTrOct];
    const uint32_t tid = threadIdx.x;
    const uint64_t rb = (uint64_t)blockIdx.x * 256u;          // first read of the tile, within the batch
    const uint32_t d0 = blockIdx.y * 32u;                     // first dword of the tile's columns
    // 16 bytes per lane and load, all eight of a thread in flight before the first is used (dword loads left the kernel
    // at 1.7 TB/s: too few bytes in flight); the scratch rows are 128-byte aligned, so a piece never leaves its row
    uint4 v[8];
#pragma unroll
    for (uint32_t k = 0; k < 8u; ++k) {
        const uint32_t i = tid + 256u * k, rr = i >> 3, q = i & 7u;
        v[k] = make_uint4(0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u);
        if (rb + rr < n_batch && d0 + 4u * q < row_dwords)
            v[k] = *reinterpret_cast<const uint4 *>(rows4 + (rb + rr) * (uint64_t)row_dwords + d0 + 4u * q);
    }
#pragma unroll
    for (uint32_t k = 0; k < 8u; ++k) {
        const uint32_t i = tid + 256u * k, rr = i >> 3, q = i & 7u;
        uint32_t *dst = s_t + (rr & 7u) * kTrOct + (rr >> 3) * 33u + 4u * q;
        dst[0] = v[k].x; dst[1] = v[k].y; dst[2] = v[k].z; dst[3] = v[k].w;
    }
    __syncthreads();
    // 8 groups of 32 reads x 32 dwords of 8 columns, one per thread: a dword of each plane of each column out; the lanes of
    // a quad... of eight lanes write 32 consecutive bytes of one plane
    const uint32_t G = tid & 7u, dwi = tid >> 3;
    uint32_t R[32];
#pragma unroll
    for (uint32_t i = 0; i < 32u; ++i) {
        const uint32_t rr = 32u * G + i;
        R[i] = s_t[(rr & 7u) * kTrOct + (rr >> 3) * 33u + dwi];
    }
    uint32_t out[8][3];
    nibble_rows_to_plane_words(R, out);
    const uint64_t byte = (r0 + rb) / 8u + (uint64_t)G * 4u;
    if (byte < plane_stride) {
#pragma unroll
        for (uint32_t j = 0; j < 8u; ++j) {
            const uint32_t c = (d0 + dwi) * 8u + j;
            if (c < n_cols) {
#pragma unroll
                for (uint32_t k = 0; k < 3u; ++k)
                    *reinterpret_cast<uint32_t *>(msa + ((uint64_t)c * 3u + k) * plane_stride + byte) = out[j][k];
            }
        }
    }
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

// rows4: scratch of jl_ingest_batch_reads(ctx) x jl_ingest_row_dwords(ctx) dwords
// a by-row scratch row holds 8 columns per dword; rows start on 128-byte lines, so that a transpose tile's 128-byte
// pieces of 256 rows are whole lines (unaligned they straddle two: 1.5x the read traffic)
uint32_t jl_ingest_row_dwords(const jl_ctx *ctx) { return ((ctx->n_cols + 7u) / 8u + 31u) & ~31u; }
uint64_t jl_ingest_batch_reads(const jl_ctx *ctx)
{
    const uint64_t pad = ctx->col_stride * 2u;   // reads incl. the padding of a column: a multiple of 256
    return pad < (1ull << 20) ? pad : (1ull << 20);
}

// max_ops / max_seq_bytes: the longest cigar and the most packed-base bytes of any read (they size the waves' LDS)
void jl_launch_ingest(jl_ctx *ctx, const int32_t *d_pos, const uint32_t *d_cigar, const uint64_t *d_cig_off,
                      const uint8_t *d_seq4, const uint64_t *d_seq_off, const uint8_t *d_qual,
                      const uint64_t *d_qual_off, uint32_t min_qv, uint32_t *d_rows4, uint64_t max_ops, uint64_t max_seq_bytes)
{
    const uint64_t pad = ctx->col_stride * 2u, batch = jl_ingest_batch_reads(ctx);
    const uint32_t row_dwords = jl_ingest_row_dwords(ctx);
    const uint32_t ops_cap = (uint32_t)std::min<uint64_t>(kExpandOps, std::max<uint64_t>(64, (max_ops + 63u) / 64u * 64u));
    const uint32_t seq_cap = (uint32_t)std::min<uint64_t>(kExpandSeqDw, (max_seq_bytes + 3u) / 4u + 2u);
    const uint32_t lds = 4u * (2u * ops_cap + seq_cap + 2u) * 4u;
    for (uint64_t r0 = 0; r0 < pad; r0 += batch) {
        const uint64_t nb = pad - r0 < batch ? pad - r0 : batch;
        hipLaunchKernelGGL(expand_rows_kernel, dim3((uint32_t)((nb + 3u) / 4u)), dim3(256), lds, ctx->stream, r0, nb, ctx->n_reads,
                           ctx->n_cols, ctx->win_begin, d_pos, d_cigar, d_cig_off, d_seq4, d_seq_off, d_qual, d_qual_off, min_qv,
                           d_rows4, row_dwords, ops_cap, seq_cap);
        hipLaunchKernelGGL(transpose_rows_kernel, dim3((uint32_t)((nb + 255u) / 256u), (row_dwords + 31u) / 32u), dim3(256), 0,
                           ctx->stream, (const uint32_t *)d_rows4, row_dwords, r0, nb, ctx->n_cols, ctx->d_msa, ctx->plane_stride);
    }
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
