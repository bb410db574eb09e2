// kernels_ingest.hip — aligned BAM records -> the resident bit planes, entirely on the device.
//
// Behaviour: doc/JULIET.md:26-27 (insertions dropped, deletions '-'), :53 (PacBio cigars = X I D S H N; M is rejected on the
// host), :256-259 (a QV-filtered base shows up as N).  Reads past n_reads (the padding of a plane up to its stride) and
// columns outside a read's span are 'not covered' (code 6).
//
// Three launches, no by-row scratch in HBM (rounds 1-3 expanded every read into a by-row nibble matrix, transposed that
// into a column-packed one and made the planes from it: 870 MB moved for the 270 MB that are needed):
//   cigar_runs_kernel   one wave per read: prefix sums over the cigar -> the read's RUNS (stretches of '=' / 'X' merge into one
//                       run of aligned bases; D and N are runs of their own; I / S / H / P only end a run), 8 bytes each, plus
//                       for every column sweep of the window the index of the run that contains its first column.
//   ingest_planes_kernel  one workgroup = 256 reads x one sweep of 224 columns.  It loads the handful of runs its reads have
//                       in the sweep into LDS, then works INPUT-driven: a lane takes 16 aligned bytes = 32 bases of a read
//                       straight from HBM (coalesced, all of a wave's loads in flight before the first is used), converts
//                       them to symbol codes nibble-parallel, finds the run(s) they belong to and XORs them — shifted to
//                       their columns — into a by-row nibble tile in LDS that starts out as 'not covered' (a cell is
//                       written by exactly one run, so XOR against the initial code stores it; deletions are XORed in by a
//                       pass over the D runs; reference skips need nothing).  The tile then leaves as planes: a thread
//                       takes 32 reads x 8 columns, transposes four 8 x 8 nibble blocks in registers, splits the codes into
//                       their three bits and stores a dword of each plane of each column; the four workgroups that share
//                       the 128-byte lines of a sweep run on one XCD next to each other (blockIdx mapping), so the lines
//                       are completed in that XCD's L2.
//   ingest_slow_kernel  the (read, sweep) pairs a workgroup could not take — more runs in one sweep than its LDS list
//                       holds (a deletion every other column), a huge insertion inside a sweep: column by column from the
//                       runs in HBM, bits flipped with atomics.  Empty for real CCS data.
#include <stdlib.h>

#include <algorithm>

#include "jl_internal.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr uint32_t kSweep = JL_INGEST_SWEEP;        // columns per workgroup
constexpr uint32_t kSweepDw = kSweep / 8u;          // dwords of 8 columns in a tile row
#ifndef JL_INGEST_TILE
#define JL_INGEST_TILE 128
#endif
constexpr uint32_t kTileReads = JL_INGEST_TILE;     // reads per workgroup: 128 (22.7 KB of LDS: seven workgroups per CU) or 256
constexpr uint32_t kTileGroups = kTileReads / 32u;  // groups of 32 reads = dwords of a plane the tile writes per column
constexpr uint32_t kSubTiles = 1024u / kTileReads;  // tiles that share the 128-byte lines of the planes
// tile row of read r at (r & 31) * kRowI + (r >> 5) * kSweepDw: the eight reads a wave expands together lie an odd number of
// banks apart, the 32-read groups of the transposing step four banks apart
constexpr uint32_t kRowI = kTileGroups * kSweepDw + 1u;
constexpr uint32_t kTileDw = 32u * kRowI;
#ifndef JL_INGEST_ENT_PER_READ
#define JL_INGEST_ENT_PER_READ 4
#endif
// run entries of the workgroup's reads in its sweep (LDS).  Four per read on average (CCS reads need 2.6: a run, half a deletion,
// the end entry) keep the workgroup below 22.8 KB of LDS: seven workgroups per CU instead of six (136 against 141 us); a tile
// that needs more leaves its last reads to the slow kernel
constexpr uint32_t kEntCap = JL_INGEST_ENT_PER_READ * kTileReads;
constexpr uint32_t kRunMask = 0x3FFFFFFFu;          // reference offset of a run; kind in the two bits above
constexpr uint32_t kMaxPieces = 1023u;              // pieces of a read in a sweep (10 bits; the deferral list holds 12)
// 32-byte pieces (64 bases) of a read that a sweep takes from a 16-byte boundary on, insertions aside: a power of two of lanes
constexpr uint32_t kPiecesPerRead = (kSweep + 31u + 63u) / 64u <= 4u ? 4u : 8u;
static_assert(kSweep % 8u == 0 && kSweep + 31u <= 64u * kPiecesPerRead, "a sweep is at most eight pieces of a read wide");
// threads of a planes workgroup: two per read of the tile (a thread per read in the prologue, the waves share the expansion)
#ifndef JL_INGEST_THREADS
#define JL_INGEST_THREADS 256
#endif
constexpr uint32_t kThreads = JL_INGEST_THREADS, kWaves = kThreads / 64u;
static_assert(kThreads % 64u == 0 && kThreads >= kTileReads && kThreads <= 256u, "whole waves, a thread per read at least");
static_assert((JL_INGEST_TILE * kPiecesPerRead) % kThreads == 0 && JL_INGEST_TILE % 32 == 0, "whole rounds of the waves; whole plane dwords");
static_assert(kThreads / kTileGroups >= kSweepDw, "a thread per 32 reads x 8 columns in the transposing phase");

// ---------------------------------------------------------------------------------------- runs
// inclusive prefix sum over the 64 lanes by DPP (four shifts within rows of 16, two row broadcasts)
__device__ __forceinline__ uint32_t wave_scan(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2, 3
    return v;
}
// two independent sums scanned in step (a DPP read of a register needs two wait states behind the write: each chain fills
// the other's)
__device__ __forceinline__ void wave_scan2(uint32_t &a, uint32_t &b)
{
#define JL_SCAN_STEP(ctrl, rows, bc)                                                            \
    {                                                                                           \
        const uint32_t ta = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, ctrl, rows, 0xF, bc); \
        const uint32_t tb = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b, ctrl, rows, 0xF, bc); \
        a += ta;                                                                                \
        b += tb;                                                                                \
    }
    JL_SCAN_STEP(0x111, 0xF, true)
    JL_SCAN_STEP(0x112, 0xF, true)
    JL_SCAN_STEP(0x114, 0xF, true)
    JL_SCAN_STEP(0x118, 0xF, true)
    JL_SCAN_STEP(0x142, 0xA, false)
    JL_SCAN_STEP(0x143, 0xC, false)
#undef JL_SCAN_STEP
}

// runs[cig_off[r] + r + i] = {reference offset of run i relative to the read's first base | kind << 30, query offset};
// kind 1 aligned bases, 2 deletion, 3 reference skip; entry n_runs = {the read's reference length, its query length}.
// first_run[r][s], s = 0 .. n_sweeps: the number of runs i >= 1 (the end entry included) that begin at or before window
// column s * kSweep = the index of the run that contains that column (0 before the read, n_runs behind it).
// A wave takes four consecutive reads, a lane two consecutive cigar ops (128 ops a step: a CCS read's cigar in one or two
// steps); the offsets and the first step's cigar words of all four reads are requested before anything waits — with one
// read per wave the kernel was a chain of three trips to HBM per wave and nothing else (51 us for 100k reads).
constexpr uint32_t kRunsReadsPerWave = 4u;
// The records are untrusted: a cigar with an 'M' (forbidden in PacBio BAM, doc/JULIET.md:53) or one that consumes more bases
// (or qualities) than the record holds is reported — *bad = min over such reads of (read << 8 | code), code 1 'M', 2 bases,
// 3 qualities — and the read is treated as covering nothing, so no later kernel follows its offsets anywhere.
__global__ __launch_bounds__(256) void cigar_runs_kernel(uint64_t n_reads, const int32_t *__restrict__ pos, const uint32_t *__restrict__ cigar,
                                                         const uint64_t *__restrict__ cig_off, const uint64_t *__restrict__ seq_off,
                                                         const uint64_t *__restrict__ qual_off, uint32_t win_begin, uint32_t n_cols,
                                                         uint32_t n_sweeps, uint2 *__restrict__ runs, uint32_t *__restrict__ nruns,
                                                         uint32_t *__restrict__ first_run, unsigned long long *__restrict__ bad)
{
    extern __shared__ uint32_t s_dyn[];
    const uint32_t wid = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint64_t r0 = ((uint64_t)blockIdx.x * 4u + wid) * kRunsReadsPerWave;
    if (r0 >= n_reads) return;
    uint32_t *hist = s_dyn + (size_t)wid * (n_sweeps + 2u);
    // lanes 0..4: the cigar offsets of the wave's reads (one more than reads), lanes 0..3 their positions
    const uint64_t rl_ = r0 + lane;
    const uint64_t co_l = (lane <= kRunsReadsPerWave && rl_ <= n_reads) ? cig_off[rl_] : 0u;
    const int32_t pos_l = (lane < kRunsReadsPerWave && rl_ < n_reads) ? pos[rl_] : 0;
    uint64_t cb[kRunsReadsPerWave + 1u];
#pragma unroll
    for (uint32_t q = 0; q <= kRunsReadsPerWave; ++q)
        cb[q] = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(co_l >> 32), (int)q) << 32) |
                (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)co_l, (int)q);   // (a lane known at compile time: no trip through the LDS crossbar)
    uint32_t cw0[kRunsReadsPerWave][2];
#pragma unroll
    for (uint32_t q = 0; q < kRunsReadsPerWave; ++q) {
        const uint64_t k = cb[q] + 2u * lane;
        const bool live = r0 + q < n_reads;
        cw0[q][0] = (live && k < cb[q + 1u]) ? cigar[k] : 0u;
        cw0[q][1] = (live && k + 1u < cb[q + 1u]) ? cigar[k + 1u] : 0u;
    }
#pragma unroll
    for (uint32_t q = 0; q < kRunsReadsPerWave; ++q) {
        const uint64_t r = r0 + q;
        if (r >= n_reads) break;
        for (uint32_t s = lane; s < n_sweeps + 2u; s += 64u) hist[s] = 0;
        const uint64_t c_beg = cb[q], c_end = cb[q + 1u];
        uint2 *out = runs + c_beg + r;
        const int64_t base = (int64_t)__builtin_amdgcn_readlane(pos_l, (int)q) - (int64_t)win_begin;
        // first sweep whose first column is at or behind the run's start (32-bit arithmetic: the offset is clamped to the
        // window first — a 64-bit division by the sweep width was a quarter of this kernel's instructions)
        const int64_t lim = (int64_t)n_sweeps * kSweep;
        auto sweep_of = [&](uint32_t rb) -> uint32_t {
            const int64_t w = base + (int64_t)rb;
            if (w <= 0) return 0u;
            if (w > lim) return n_sweeps + 1u;
            return ((uint32_t)w + kSweep - 1u) / kSweep;
        };
        uint32_t n_runs = 0, prev_kind = 0, ref_carry = 0, q_carry = 0;
        bool has_m = false;
        // (op indices relative to the read's first, in 32 bits: a record's cigar has fewer than 2^32 ops)
        const uint32_t n_ops = (uint32_t)min(c_end - c_beg, (uint64_t)0xFFFFFF00u);
        const uint32_t *cig = cigar + c_beg;
        for (uint32_t k0 = 0; k0 < n_ops; k0 += 128u) {
            uint32_t cw[2] = {cw0[q][0], cw0[q][1]};
            if (k0 != 0u) {
                const uint32_t k = k0 + 2u * lane;
                cw[0] = k < n_ops ? cig[k] : 0u;
                cw[1] = k + 1u < n_ops ? cig[k + 1u] : 0u;
            }
            uint32_t kind[2], rl[2], ql[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                // what an op does, from three constants indexed by its code (a chain of comparisons per property was a fifth of
                // this kernel): D N = X consume the reference, I S = X the query; kind 1 for = X, 2 for D, 3 for N
                const uint32_t op = cw[t] & 15u, len = cw[t] >> 4;   // (a missing op is the word 0: length 0, which is nothing)
                constexpr uint32_t kRefOps = (1u << 2) | (1u << 3) | (1u << 7) | (1u << 8);
                constexpr uint32_t kQueryOps = (1u << 1) | (1u << 4) | (1u << 7) | (1u << 8);
                constexpr uint32_t kKinds = (2u << 4) | (3u << 6) | (1u << 14) | (1u << 16);   // two bits per op
                rl[t] = ((kRefOps >> op) & 1u) ? len : 0u;
                ql[t] = ((kQueryOps >> op) & 1u) ? len : 0u;
                kind[t] = len == 0u ? 0u : (kKinds >> (2u * op)) & 3u;
                has_m = has_m || (op == 0u && (k0 + 2u * lane + (uint32_t)t) < n_ops);
            }
            uint32_t ri = rl[0] + rl[1], qi = ql[0] + ql[1];
            wave_scan2(ri, qi);   // inclusive, per lane pair
            // the kind of the op before this lane's first one: the previous lane's second op
            uint32_t before = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)kind[1], 0x138, 0xF, 0xF, false);   // wave_shr:1
            if (lane == 0) before = prev_kind;
            const bool st0 = kind[0] != 0u && !(kind[0] == 1u && before == 1u);
            const bool st1 = kind[1] != 0u && !(kind[1] == 1u && kind[0] == 1u);
            const uint64_t b0 = __ballot(st0), b1 = __ballot(st1);
            // run starts in the lanes before this one (v_mbcnt: the set bits of a mask below the lane, with an addend)
            const uint32_t ahead = __builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1,
                                   __builtin_amdgcn_mbcnt_hi((uint32_t)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b0, 0u))));
            const uint32_t rbeg0 = ref_carry + ri - rl[0] - rl[1], qbeg0 = q_carry + qi - ql[0] - ql[1];
            if (st0) {
                const uint32_t idx = n_runs + ahead;
                const uint32_t rb = rbeg0 & kRunMask;
                out[idx] = make_uint2(rb | (kind[0] << 30), qbeg0);
                if (idx >= 1u) atomicAdd(&hist[sweep_of(rb)], 1u);
            }
            if (st1) {
                const uint32_t idx = n_runs + ahead + (st0 ? 1u : 0u);
                const uint32_t rb = (rbeg0 + rl[0]) & kRunMask;
                out[idx] = make_uint2(rb | (kind[1] << 30), qbeg0 + ql[0]);
                if (idx >= 1u) atomicAdd(&hist[sweep_of(rb)], 1u);
            }
            n_runs += (uint32_t)__popcll(b0) + (uint32_t)__popcll(b1);
            prev_kind = (uint32_t)__builtin_amdgcn_readlane((int)kind[1], 63);   // (only a full step has a successor)
            ref_carry += (uint32_t)__builtin_amdgcn_readlane((int)ri, 63);
            q_carry += (uint32_t)__builtin_amdgcn_readlane((int)qi, 63);
        }
        uint32_t code = __ballot(has_m) != 0ull ? 1u : 0u;
        if (!code) {
            const uint64_t so0 = seq_off[r], so1 = seq_off[r + 1];
            if ((uint64_t)q_carry > 2u * (so1 - so0)) code = 2u;
            else if (qual_off && (uint64_t)q_carry > qual_off[r + 1] - qual_off[r]) code = 3u;
        }
        if (code) {
            n_runs = 0;
            if (lane == 0) atomicMin(bad, ((unsigned long long)r << 8) | code);
        }
        if (lane == 0) {
            out[n_runs] = make_uint2(ref_carry & kRunMask, q_carry);
            nruns[r] = n_runs;
            if (n_runs) atomicAdd(&hist[sweep_of(ref_carry & kRunMask)], 1u);
        }
        __builtin_amdgcn_wave_barrier();
        uint32_t carry = 0;
        for (uint32_t s0 = 0; s0 <= n_sweeps; s0 += 64u) {
            const uint32_t s = s0 + lane;
            const uint32_t v = wave_scan((s <= n_sweeps && !code) ? hist[s] : 0u);
            if (s <= n_sweeps) first_run[r * (uint64_t)(n_sweeps + 1u) + s] = carry + v;
            carry += (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------- the planes of one sweep
struct read_info {     // what the expansion needs of one read of the tile (LDS)
    int32_t base;      // window column of the read's first reference base
    uint32_t ent;      // first entry in s_ent (bits 0-11) | entries (12-21) | pieces (22-31)
    uint32_t p0_lo, p0_hi;   // byte offset, within the packed bases, of the read's first 16-byte piece in this sweep
    int32_t q0;        // query offset of that piece's first base (>= -30: a piece may begin inside the previous read)
    uint32_t pad_;
};

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
// 4 n + g at column j — then the bits of the four blocks' nibbles change places (below).
__device__ __forceinline__ void nibble_rows_to_plane_words(const uint32_t (&R)[32], uint32_t (&out)[8][3])
{
    uint32_t M[4][8];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int i = 0; i < 8; ++i) M[g][i] = R[4 * i + g];
        transpose_nibbles_8x8(M[g]);
    }
    // Plane k of column j wants, in nibble n, bit k of the four blocks' nibbles n (reads 4 n .. 4 n + 3): a 4 x 4 bit transpose
    // between the four words, in every nibble at once — two rounds of masked swaps, 24 operations a column where pulling
    // each bit out and shifting it into place took 33.
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        uint32_t a0 = M[0][j], a1 = M[1][j], a2 = M[2][j], a3 = M[3][j], t;
        t = ((a0 >> 1) ^ a1) & 0x55555555u; a1 ^= t; a0 ^= t << 1;
        t = ((a2 >> 1) ^ a3) & 0x55555555u; a3 ^= t; a2 ^= t << 1;
        t = ((a0 >> 2) ^ a2) & 0x33333333u; a2 ^= t; a0 ^= t << 2;
        t = ((a1 >> 2) ^ a3) & 0x33333333u; a3 ^= t; a1 ^= t << 2;
        out[j][0] = a0;
        out[j][1] = a1;
        out[j][2] = a2;   // (a3 would be bit 3 of the codes: always zero)
    }
}

struct ingest_args {
    uint64_t n_reads;
    uint32_t n_cols, n_sweeps, n_groups, n_pairs;   // n_groups: groups of 1024 reads (the tiles that share lines); n_pairs = n_groups * n_sweeps
    uint32_t win_begin, min_qv;
    const int32_t *pos;
    const uint64_t *cig_off;
    const uint8_t *seq4;
    const uint64_t *seq_off;
    const uint8_t *qual;         // null: no QV masking
    const uint64_t *qual_off;
    const uint2 *runs;
    const uint32_t *nruns, *first_run;
    uint32_t *slow_count;
    uint2 *slow_list;            // {read, sweep}
    uint8_t *msa;
    uint64_t plane_stride;
    uint32_t skip;               // tuning builds: bit 0 no bases, bit 1 no deletions, bit 2 no stores, bit 3 no second pass, bit 4 no transposing
};
#ifdef JL_TUNING
#define JL_ING_SKIP(a, bit) (((a).skip >> (bit)) & 1u)
#else
#define JL_ING_SKIP(a, bit) 0u
#endif

constexpr int kPiece = 64;     // bases a lane expands at a time: 32 bytes of packed bases (two 16-byte loads)

// the nibbles [lo, hi) of the 64 a piece holds, as four 64-bit masks (part k = nibbles 16 k .. 16 k + 15)
__device__ __forceinline__ uint64_t range_mask16(int lo, int hi, int k)
{
    const int l = min(max(lo - 16 * k, 0), 16), h = min(max(hi - 16 * k, 0), 16);
    if (h <= l) return 0ull;
    const uint64_t upto = h == 16 ? ~0ull : ((1ull << (4 * h)) - 1ull);
    return upto & ~((1ull << (4 * l)) - 1ull);
}

// 32 bytes of packed bases (BAM order: first base in the high nibble) -> 64 symbol codes, base b in nibble b & 7 of S[b >> 3];
// QV: bases whose quality is below min_qv become N.  Q = query offset of the piece's base 0, lo_v = its first base that is
// the read's own.
template <bool QV>
__device__ __forceinline__ void piece_syms(const ingest_args &a, const uint4 &v0, const uint4 &v1, int Q, int lo_v, uint64_t qual_base,
                                           uint32_t (&S)[8])
{
    const uint32_t w8[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    // BAM's 4-bit base -> symbol code by table: A C G T (1 2 4 8) -> 0..3, everything else (N = 15, '=' = 0, the IUPAC
    // ambiguity codes) -> N, a filtered base.  v_perm_b32 looks four bytes up in an 8-byte table: the low three bits of a
    // base select in the table of the codes 0..7 and in that of 8..15, bit 3 picks between the two results.  The first base
    // of a byte is its HIGH nibble and goes to the lower column: the two halves are put together the other way round.
    // (Counting the set bits of every nibble, mapping one-hot to index and patching the rest was 29 instructions a dword,
    // and two pieces in three hold an N; this is 18.)
    constexpr uint32_t kLo03 = 0x05010005u, kLo47 = 0x05050502u;   // codes of 0..3 (bytes 0..3), of 4..7
    constexpr uint32_t kHi03 = 0x05050503u, kHi47 = 0x05050505u;   // codes of 8..11, of 12..15
    auto lookup4 = [&](uint32_t x) -> uint32_t {   // four bases, one per byte (0..15) -> four codes
        const uint32_t sel = x & 0x07070707u;
        const uint32_t lo = __builtin_amdgcn_perm(kLo47, kLo03, sel), hi = __builtin_amdgcn_perm(kHi47, kHi03, sel);
        // bytes of ones where bit 3 is set: 8 << 5 minus 8 >> 3 within each byte (a 32-bit multiply is a quarter-rate instruction)
        const uint32_t b3 = x & 0x08080808u, pick = (b3 << 5) - (b3 >> 3);
        return (hi & pick) | (lo & ~pick);
    };
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t first = lookup4((w8[k] >> 4) & 0x0F0F0F0Fu), second = lookup4(w8[k] & 0x0F0F0F0Fu);
        S[k] = first | (second << 4);
    }
    if (QV) {
        // qualities of the piece's bases, one byte each, from the aligned dwords around them; a base below min_qv
        // becomes N (0xFF = absent never does)
        const uint64_t addr = qual_base + (uint64_t)(Q + lo_v);    // first quality wanted
        const uint32_t *qp = reinterpret_cast<const uint32_t *>(a.qual + (addr & ~(uint64_t)3));
        const uint32_t sh = (uint32_t)(addr & 3u);
        uint32_t qw[17];
#pragma unroll
        for (int i = 0; i < 17; ++i) qw[i] = qp[i];
        const uint32_t T = a.min_qv * 0x01010101u;
        uint64_t flags = 0;   // bit b: base lo_v + b is masked
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t x = __builtin_amdgcn_alignbyte(qw[i + 1], qw[i], sh);
            const uint32_t lt = ~((x | 0x80808080u) - T) & 0x80808080u;       // bytes below min_qv (and below 128)
            uint32_t f = lt >> 7;
            f = (f | (f >> 7)) & 0x00030003u;
            f = (f | (f >> 14)) & 0xFu;
            flags |= (uint64_t)f << (4 * i);
        }
        flags <<= lo_v;   // by the piece's own base index (bases past 64 drop out)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint32_t b = (uint32_t)(flags >> (8 * k)) & 0xFFu;
            b = (b | (b << 12)) & 0x000F000Fu;
            b = (b | (b << 6)) & 0x03030303u;
            b = (b | (b << 3)) & 0x11111111u;
            const uint32_t mk = b * 15u;
            S[k] = (S[k] & ~mk) | ((uint32_t)JL_SYM_MASK * 0x11111111u & mk);
        }
    }
}

// The part of a piece (bases Q .. Q + 63 of the read, the first lo_v of them not its own) that lies in ONE run of aligned
// bases {rb, q, len} and inside the sweep goes into the tile row: XOR against 'not covered', shifted to its columns.
// s_row may be XORed up to eight dwords before and behind the row's own: the payload there is zero (the guard dwords of
// the tile take what would fall outside it).
__device__ __forceinline__ void emit_run(const uint32_t (&S)[8], int Q, int lo_v, int rb, int q, int len, int base, int X, int Xend, uint32_t *s_row)
{
    const int qa = max(q, Q + lo_v), qe = min(q + len, Q + kPiece);
    if (qa >= qe) return;
    const int col_a = base + rb + (qa - q);                 // window column of base qa
    const int ca = max(col_a, X), cb = min(col_a + (qe - qa), Xend);
    if (ca >= cb) return;
    const int lo = qa - Q + (ca - col_a), hi = lo + (cb - ca);   // the piece's bases [lo, hi) go to tile columns ca - X ...
    const int delta = (ca - X) - lo;                              // tile column of the piece's base 0
    const uint32_t s4 = 4u * (uint32_t)(delta & 7);
    const int dd = delta >> 3;
    uint32_t P[10];
    P[0] = 0; P[9] = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint64_t mk = range_mask16(lo, hi, k);
        P[2 * k + 1] = (S[2 * k] ^ 0x66666666u) & (uint32_t)mk;
        P[2 * k + 2] = (S[2 * k + 1] ^ 0x66666666u) & (uint32_t)(mk >> 32);
    }
    // Dword j of the row's part takes {P[j + 1], P[j]} >> (32 - s4).  v_alignbit_b32 does that in one instruction for shifts
    // below 32; with s4 = 0 it returns P[j] — the value that belongs one dword further down — so the destination moves
    // instead (P[0] = 0 lands in the guard).  (As 64-bit shifts these were two instructions and two moves each.)
    uint32_t *dst = s_row + dd - (s4 == 0u ? 1 : 0);
    const uint32_t sh = (32u - s4) & 31u;
#pragma unroll
    for (int j = 0; j < 9; ++j) atomicXor(&dst[j], __builtin_amdgcn_alignbit(P[j + 1], P[j], sh));
}

// every run from entry i on that the piece reaches (the general form: pieces that cross a run boundary, or begin in a gap)
__device__ __forceinline__ void emit_rest(const uint32_t (&S)[8], int Q, int lo_v, const uint2 *ent, uint32_t i, uint32_t cnt, int base, int X,
                                          int Xend, uint32_t *s_row)
{
    for (; i < cnt; ++i) {
        const uint2 e = ent[i];
        const int q = (int)e.y;
        if (q >= Q + kPiece) break;
        if ((e.x >> 30) != 1u) continue;
        const int rb = (int)(e.x & kRunMask);
        emit_run(S, Q, lo_v, rb, q, (int)(ent[i + 1u].x & kRunMask) - rb, base, X, Xend, s_row);
    }
}

constexpr uint32_t kListCap = 2u * kTileReads;     // pieces a workgroup defers to its second pass

template <bool QV>
__global__ __launch_bounds__(kThreads) void ingest_planes_kernel(ingest_args a)
{
    __shared__ uint32_t s_tile_g[kTileDw + 24u];   // guard dwords: 8 in front, 16 behind (see emit_run)
    __shared__ uint2 s_ent[kEntCap];
    __shared__ read_info s_info[kTileReads];
    __shared__ uint32_t s_qlo[QV ? kTileReads : 1], s_qhi[QV ? kTileReads : 1];   // qual_off of every read
    __shared__ uint32_t s_list[kListCap];
    __shared__ uint32_t s_wsum[4], s_nlist;
    uint32_t *s_tile = s_tile_g + 8;
    const uint32_t tid = threadIdx.x, wid = tid >> 6, lane = tid & 63u;
    // block -> (read tile, sweep).  Blocks b, b + 8, b + 16, ... are dealt to the same XCD one after the other; an XCD takes
    // whole groups of 1024 reads (group = xcd, xcd + 8, ...), and of a group all sweeps in turn, the tiles of the group
    // innermost.  So (a) the tiles that share the 128-byte lines of a sweep's planes meet in one L2, and (b) everything the
    // sweeps of a group read again — the reads' offsets, run entries and per-sweep run indices, and the 128-byte lines of
    // packed bases that straddle two sweeps (a sweep is 112 bytes of a read: 2.1 x the bases were fetched) — is in that L2
    // when the next sweep asks for it: the prologue's scattered loads become L2 hits.
    const uint32_t b = blockIdx.x, xcd = b & 7u, jb = b >> 3, sub = jb % kSubTiles, q = jb / kSubTiles;
    const uint32_t group = xcd + 8u * (q / a.n_sweeps), sweep = q % a.n_sweeps;
    if (group >= a.n_groups) return;
    const uint32_t tile = kSubTiles * group + sub;
    const int X = (int)(sweep * kSweep), Xend = (int)min(a.n_cols, sweep * kSweep + kSweep);
    const uint64_t r = (uint64_t)tile * kTileReads + tid;
    const bool real = tid < kTileReads && r < a.n_reads;

    // ---- 0. what this read has in the sweep (the loads fly while the tile is set to 'not covered')
    uint32_t f0 = 0, f1 = 0, nr = 0;
    uint64_t co = 0, so = 0;
    int32_t p = 0;
    if (real) {
        // (the two indices in one request: every scattered request of the prologue shows in the kernel's time)
        const uint32_t *fr = a.first_run + r * (uint64_t)(a.n_sweeps + 1u) + sweep;
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2), aligned(4)));
        const u32x2 f01 = *reinterpret_cast<const u32x2 *>(fr);
        f0 = f01.x;
        f1 = f01.y;
        nr = a.nruns[r];
        co = a.cig_off[r];
        so = a.seq_off[r];
        p = a.pos[r];
        if (QV) {
            const uint64_t qo = a.qual_off[r];
            s_qlo[tid] = (uint32_t)qo;
            s_qhi[tid] = (uint32_t)(qo >> 32);
        }
    }
    for (uint32_t i = tid; i < kTileDw + 24u; i += kThreads) s_tile_g[i] = 0x66666666u;
    if (tid == 0) s_nlist = 0;
    uint32_t cnt = 0;
    if (real && nr && f0 < nr) cnt = min(f1, nr - 1u) - f0 + 1u;
    uint32_t need = cnt ? cnt + 1u : 0u;   // + the entry behind the last run: its end
    // exclusive scan of `need` over the workgroup
    const uint32_t inc = wave_scan(need);
    if (lane == 63u) s_wsum[wid] = inc;
    __syncthreads();
    uint32_t off = inc - need;
    for (uint32_t w = 0; w < wid; ++w) off += s_wsum[w];
    bool slow = need != 0u && (cnt > 1022u || off + need > kEntCap);
    const uint2 *src = a.runs + co + r + f0;
    if (need && !slow) {
        // (the first four entries in two 16-byte requests that go out together — a load-store loop pays a trip to HBM per
        // entry; entries past `need` belong to the next read or to the array's slack)
        typedef uint32_t u32x4a8 __attribute__((ext_vector_type(4), aligned(8)));
        const u32x4a8 e01 = *reinterpret_cast<const u32x4a8 *>(src), e23 = *reinterpret_cast<const u32x4a8 *>(src + 2);
        const uint2 e4[4] = {make_uint2(e01.x, e01.y), make_uint2(e01.z, e01.w), make_uint2(e23.x, e23.y), make_uint2(e23.z, e23.w)};
#pragma unroll
        for (uint32_t i = 0; i < 4u; ++i)
            if (i < need) s_ent[off + i] = e4[i];
        for (uint32_t i = 4; i < need; ++i) s_ent[off + i] = src[i];
    }
    // the bases the sweep takes of this read: query range -> 16-byte pieces of its packed bases
    const int base = (int)((int64_t)p - (int64_t)a.win_begin);
    uint32_t q_lo = 0xFFFFFFFFu, q_hi = 0;
    if (need && !slow) {
        for (uint32_t i = 0; i < cnt; ++i) {
            const uint2 e = s_ent[off + i];
            if ((e.x >> 30) != 1u) continue;
            const int rb = (int)(e.x & kRunMask), len = (int)(s_ent[off + i + 1u].x & kRunMask) - rb;
            const int W = base + rb;
            const int ca = max(W, X), cb = min(W + len, Xend);
            if (ca >= cb) continue;
            q_lo = min(q_lo, e.y + (uint32_t)(ca - W));
            q_hi = max(q_hi, e.y + (uint32_t)(cb - W));
        }
    }
    uint32_t np = 0;
    uint64_t p0 = 0;
    int32_t q0 = 0;
    if (q_lo < q_hi) {
        const uint64_t byte_lo = so + (q_lo >> 1), byte_hi = so + ((uint64_t)q_hi + 1u) / 2u;
        p0 = byte_lo & ~(uint64_t)15;
        const uint64_t n = (byte_hi - p0 + 31u) >> 5;                // pieces of 32 bytes from a 16-byte boundary on
        q0 = (int32_t)(2 * ((int64_t)p0 - (int64_t)so));
        if (n > kMaxPieces) slow = true;   // (an insertion of tens of thousands of bases inside the sweep)
        else np = (uint32_t)n;
    }
    if (slow) {
        const uint32_t at = atomicAdd(a.slow_count, 1u);
        a.slow_list[at] = make_uint2((uint32_t)r, sweep);
        cnt = 0;
        np = 0;
    }
    read_info ri;
    ri.base = base;
    ri.ent = (off & 0xFFFu) | (cnt << 12) | (np << 22);
    ri.p0_lo = (uint32_t)p0;
    ri.p0_hi = (uint32_t)(p0 >> 32);
    ri.q0 = q0;
    ri.pad_ = 0;
    if (tid < kTileReads) s_info[tid] = ri;
    __syncthreads();

    // ---- 1. deletions: every D run of this read inside the sweep ('-' = 4 = 'not covered' ^ 2)
    if (tid < kTileReads && !JL_ING_SKIP(a, 1)) {
        uint32_t *row = s_tile + (tid & 31u) * kRowI + (tid >> 5) * kSweepDw;
        for (uint32_t i = 0; i < cnt; ++i) {
            const uint2 e = s_ent[off + i];
            if ((e.x >> 30) != 2u) continue;
            const int rb = (int)(e.x & kRunMask), len = (int)(s_ent[off + i + 1u].x & kRunMask) - rb;
            const int W = base + rb;
            const int ta = max(W, X) - X, tb = min(W + len, Xend) - X;
            for (int d = ta >> 3; ta < tb && d <= (tb - 1) >> 3; ++d) {
                const int l = max(ta - 8 * d, 0), h = min(tb - 8 * d, 8);
                const uint32_t m = (h - l == 8) ? 0xFFFFFFFFu : (((1u << (4 * (h - l))) - 1u) << (4 * l));
                atomicXor(&row[d], 0x22222222u & m);
            }
        }
    }

    // ---- 2. bases.  A wave takes sixteen reads at a time, four 32-byte pieces (64 bases) each — a sweep of 224 columns is
    // at most four such pieces of a read from a 16-byte boundary on, insertions aside; the loads of a wave's rounds are all
    // in flight before the first is used.  A lane puts the part of its piece that lies in the FIRST run it touches into the
    // tile; the pieces that go on into another run (one in seven) are listed and finished in a second pass with full lanes
    // — in line, the loop over a piece's runs made every wave walk two or three runs for them.
    if (!JL_ING_SKIP(a, 0)) {
        constexpr uint32_t kReadsPerRound = 64u / kPiecesPerRead;              // reads a wave expands at a time
        constexpr uint32_t kRounds = kTileReads / (kWaves * kReadsPerRound);   // rounds of a wave
        const uint32_t slot = lane / kPiecesPerRead, piece = lane % kPiecesPerRead;
        auto fetch = [&](uint32_t j, uint32_t pc, uint4 &v0, uint4 &v1) {
            const read_info &q = s_info[j];
            v0 = make_uint4(0, 0, 0, 0);
            v1 = v0;
            if (pc < (q.ent >> 22)) {
                const uint64_t at = (((uint64_t)q.p0_hi << 32) | q.p0_lo) + 32u * pc;
                // (plain loads: the two halves of a piece and the neighbouring lanes' pieces share lines, which a
                // non-temporal load does not keep — 2.3 x the bytes crossed the HBM with them)
                const u32x4 t0 = *reinterpret_cast<const u32x4 *>(a.seq4 + at);
                const u32x4 t1 = *reinterpret_cast<const u32x4 *>(a.seq4 + at + 16u);
                v0 = make_uint4(t0.x, t0.y, t0.z, t0.w);
                v1 = make_uint4(t1.x, t1.y, t1.z, t1.w);
            }
        };
        auto expand = [&](uint32_t j, uint32_t pc, const uint4 &v0, const uint4 &v1) {
            const read_info q = s_info[j];
            const uint32_t e_off = q.ent & 0xFFFu, cn = (q.ent >> 12) & 0x3FFu;
            if (pc >= (q.ent >> 22) || cn == 0u) return;
            const int Q = q.q0 + kPiece * (int)pc;
            const int lo_v = Q < 0 ? -Q : 0;
            uint32_t S[8];
            const uint64_t qb = QV ? (((uint64_t)s_qhi[j] << 32) | s_qlo[j]) : 0u;
            if (JL_ING_SKIP(a, 6)) { S[0] = v0.x; S[1] = v0.y; S[2] = v0.z; S[3] = v0.w; S[4] = v1.x; S[5] = v1.y; S[6] = v1.z; S[7] = v1.w; }
            else piece_syms<QV>(a, v0, v1, Q, lo_v, qb, S);
            if (JL_ING_SKIP(a, 5)) {   // (probe: the codes are used, nothing is placed)
                atomicXor(&s_tile[(j & 31u) * kRowI], S[0] ^ S[1] ^ S[2] ^ S[3] ^ S[4] ^ S[5] ^ S[6] ^ S[7]);
                return;
            }
            // the last entry whose query offset is at or before the piece's first base: three entries' offsets at once (a read
            // has one or two runs in a sweep, rarely more), the rest one by one
            const uint2 *ent = s_ent + e_off;
            const int Qs = Q + lo_v;
            uint32_t qy[4];
#pragma unroll
            for (uint32_t t = 1; t < 4u; ++t) qy[t] = ent[min(t, cn - 1u)].y;
            uint32_t i = 0;
#pragma unroll
            for (uint32_t t = 1; t < 4u; ++t) i += (t < cn && (int)qy[t] <= Qs) ? 1u : 0u;
            if (i == 3u)
                for (uint32_t t = 4; t < cn && (int)ent[t].y <= Qs; ++t) i = t;
            const uint2 e = ent[i], nx = ent[i + 1u];
            uint32_t *row = s_tile + (j & 31u) * kRowI + (j >> 5) * kSweepDw;
            const int rb = (int)(e.x & kRunMask);
            if ((e.x >> 30) == 1u) emit_run(S, Q, lo_v, rb, (int)e.y, (int)(nx.x & kRunMask) - rb, q.base, X, Xend, row);
            if (i + 1u < cn && (int)nx.y < Q + kPiece) {       // the piece reaches the next entry: later
                const uint32_t at = atomicAdd(&s_nlist, 1u);
                if (at < kListCap) s_list[at] = (j << 24) | (pc << 12) | (i + 1u);
                else emit_rest(S, Q, lo_v, ent, i + 1u, cn, q.base, X, Xend, row);
            }
        };
        uint4 va[kRounds], vb[kRounds];
#pragma unroll
        for (uint32_t it = 0; it < kRounds; ++it) fetch((wid + kWaves * it) * kReadsPerRound + slot, piece, va[it], vb[it]);
#pragma unroll
        for (uint32_t it = 0; it < kRounds; ++it) {
            const uint32_t j = (wid + kWaves * it) * kReadsPerRound + slot;
            expand(j, piece, va[it], vb[it]);
            // a read with more pieces in the sweep (insertions)
            const uint32_t npj = s_info[j].ent >> 22;
            for (uint32_t pp = piece + kPiecesPerRead; __ballot(pp < npj) != 0ull; pp += kPiecesPerRead) {
                uint4 w0, w1;
                fetch(j, pp, w0, w1);
                expand(j, pp, w0, w1);
            }
        }
    }
    __syncthreads();
    if (!JL_ING_SKIP(a, 3)) {   // the listed pieces, a lane each
        const uint32_t nl = min(s_nlist, kListCap);
        for (uint32_t k = tid; k < nl; k += kThreads) {
            const uint32_t it = s_list[k], j = it >> 24, pc = (it >> 12) & 0xFFFu;
            const read_info q = s_info[j];
            const uint64_t at = (((uint64_t)q.p0_hi << 32) | q.p0_lo) + 32u * pc;
            const uint4 v0 = *reinterpret_cast<const uint4 *>(a.seq4 + at), v1 = *reinterpret_cast<const uint4 *>(a.seq4 + at + 16u);
            const int Q = q.q0 + kPiece * (int)pc;
            const int lo_v = Q < 0 ? -Q : 0;
            uint32_t S[8];
            const uint64_t qb = QV ? (((uint64_t)s_qhi[j] << 32) | s_qlo[j]) : 0u;
            piece_syms<QV>(a, v0, v1, Q, lo_v, qb, S);
            emit_rest(S, Q, lo_v, s_ent + (q.ent & 0xFFFu), it & 0xFFFu, (q.ent >> 12) & 0x3FFu, q.base, X, Xend,
                      s_tile + (j & 31u) * kRowI + (j >> 5) * kSweepDw);
        }
    }
    __syncthreads();

    // ---- 3. the tile as planes: thread = 32 reads x 8 columns; neighbouring lanes write consecutive dwords of a plane
    {
        const uint32_t G = tid % kTileGroups, dwi = tid / kTileGroups;
        if (dwi < kSweepDw && X + 8 * (int)dwi < Xend && !JL_ING_SKIP(a, 4)) {
            uint32_t R[32];
#pragma unroll
            for (uint32_t i = 0; i < 32u; ++i) R[i] = s_tile[i * kRowI + G * kSweepDw + dwi];
            uint32_t out[8][3];
            nibble_rows_to_plane_words(R, out);
            const uint64_t byte = (uint64_t)tile * (kTileReads / 8u) + (uint64_t)G * 4u;
            if (byte < a.plane_stride && (!JL_ING_SKIP(a, 2) || out[0][0] == 0x12345u)) {
                // (one 64-bit multiply for the first plane row, then a stride at a time: twenty-four of them were a tenth of the phase)
                uint8_t *row = a.msa + (uint64_t)(((uint32_t)X + 8u * dwi) * 3u) * a.plane_stride + byte;
#pragma unroll
                for (uint32_t j = 0; j < 8u; ++j) {
                    const uint32_t c = (uint32_t)X + 8u * dwi + j;
#pragma unroll
                    for (uint32_t k = 0; k < 3u; ++k) {
                        if ((int)c < Xend) *reinterpret_cast<uint32_t *>(row) = out[j][k];
                        row += a.plane_stride;
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------- what the tiles left out
// One wave per (read, sweep) pair: a lane per column looks its run up in HBM and flips the bits in which the symbol differs
// from 'not covered' (which is what the tile's workgroup stored for the read).
__global__ __launch_bounds__(256) void ingest_slow_kernel(ingest_args a, const int32_t *__restrict__ pos_, const uint32_t cap)
{
    const uint32_t wid = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t n = *a.slow_count;
    if (n > cap) n = cap;
    for (uint32_t it = blockIdx.x * 4u + wid; it < n; it += gridDim.x * 4u) {
        const uint2 pr = a.slow_list[it];
        const uint64_t r = pr.x;
        const uint32_t sweep = pr.y;
        const int X = (int)(sweep * kSweep), Xend = (int)min(a.n_cols, sweep * kSweep + kSweep);
        const uint2 *runs = a.runs + a.cig_off[r] + r;
        const uint32_t nr = a.nruns[r];
        const int64_t base = (int64_t)pos_[r] - (int64_t)a.win_begin;
        const uint64_t so = a.seq_off[r];
        const uint64_t qo = a.qual ? a.qual_off[r] : 0u;
        for (int c = X + (int)lane; c < Xend; c += 64) {
            const int64_t x = (int64_t)c - base;
            if (x < 0 || nr == 0u) continue;
            uint32_t lo = 0, hi = nr + 1u;   // entries 0 .. nr (the end entry); the last one with rbeg <= x
            while (hi - lo > 1u) {
                const uint32_t mid = (lo + hi) >> 1;
                if ((int64_t)(runs[mid].x & kRunMask) <= x) lo = mid;
                else hi = mid;
            }
            if (lo >= nr) continue;          // behind the read
            const uint2 e = runs[lo];
            const uint32_t kind = e.x >> 30;
            uint32_t sym = 6u;
            if (kind == 2u) sym = 4u;
            else if (kind == 1u) {
                const uint64_t q = (uint64_t)e.y + (uint64_t)(x - (int64_t)(e.x & kRunMask));
                const uint32_t by = a.seq4[so + (q >> 1)];
                const uint32_t b16 = (q & 1u) ? (by & 15u) : (by >> 4);
                sym = (uint32_t)((0x5555555355525105ull >> (4u * b16)) & 15ull);   // A=1 C=2 G=4 T=8 -> 0..3, else 5
                if (a.qual) {
                    const uint32_t qv = a.qual[qo + q];
                    if (qv != 0xFFu && qv < a.min_qv) sym = JL_SYM_MASK;
                }
            }
            const uint32_t flip = sym ^ 6u;
            const uint32_t bit = 1u << (uint32_t)(r & 31u);
#pragma unroll
            for (uint32_t k = 0; k < 3u; ++k)
                if ((flip >> k) & 1u)
                    atomicXor(reinterpret_cast<uint32_t *>(a.msa + ((uint64_t)c * 3u + k) * a.plane_stride + (r >> 5) * 4u), bit);
        }
    }
}

}  // namespace

uint32_t jl_ingest_sweeps(uint32_t n_cols) { return (n_cols + kSweep - 1u) / kSweep; }

// d_runs: n_cig + n_reads + 1 entries; d_nruns: n_reads; d_first: n_reads x (sweeps + 1); d_slow: n_reads x sweeps pairs behind
// one counter word (zeroed here); d_slow_count[2..3] = the 64-bit word of the first malformed record (all ones: none).
// Everything is enqueued on ctx->stream; nothing waits.
void jl_launch_ingest(jl_ctx *ctx, const int32_t *d_pos, const uint32_t *d_cigar, const uint64_t *d_cig_off,
                      const uint8_t *d_seq4, const uint64_t *d_seq_off, const uint8_t *d_qual,
                      const uint64_t *d_qual_off, uint32_t min_qv, uint2 *d_runs, uint32_t *d_nruns, uint32_t *d_first,
                      uint32_t *d_slow_count, uint2 *d_slow)
{
    hipStream_t st = ctx->stream;
    const uint32_t ns = jl_ingest_sweeps(ctx->n_cols);
    hipMemsetAsync(d_slow_count, 0, 4, st);
    hipMemsetAsync(d_slow_count + 2, 0xFF, 8, st);
    if (ctx->n_reads)
        hipLaunchKernelGGL(cigar_runs_kernel, dim3((uint32_t)((ctx->n_reads + 4u * kRunsReadsPerWave - 1u) / (4u * kRunsReadsPerWave))), dim3(256), 4u * (ns + 2u) * 4u, st, ctx->n_reads, d_pos,
                           d_cigar, d_cig_off, d_seq_off, d_qual ? d_qual_off : nullptr, ctx->win_begin, ctx->n_cols, ns, d_runs, d_nruns, d_first,
                           reinterpret_cast<unsigned long long *>(d_slow_count + 2));
    ingest_args a;
    a.n_reads = ctx->n_reads;
    a.n_cols = ctx->n_cols;
    a.n_sweeps = ns;
    const uint64_t reads_pad = ctx->plane_stride * 8u;                 // a multiple of 1024: whole line groups of tiles
    a.n_groups = (uint32_t)(reads_pad / 1024u);
    a.n_pairs = a.n_groups * ns;
    a.win_begin = ctx->win_begin;
    a.min_qv = std::min<uint32_t>(min_qv, 127u);   // (the byte-parallel compare of the QV path; BAM qualities end at 93)
    a.pos = d_pos;
    a.cig_off = d_cig_off;
    a.seq4 = d_seq4;
    a.seq_off = d_seq_off;
    const bool qv = d_qual != nullptr && min_qv != 0u;
    a.qual = qv ? d_qual : nullptr;
    a.qual_off = qv ? d_qual_off : nullptr;
    a.runs = d_runs;
    a.nruns = d_nruns;
    a.first_run = d_first;
    a.slow_count = d_slow_count;
    a.slow_list = d_slow;
    a.msa = ctx->d_msa;
    a.plane_stride = ctx->plane_stride;
    a.skip = 0;
#ifdef JL_TUNING
    if (const char *e = getenv("JL_ING_SKIP")) a.skip = (uint32_t)atoi(e);
#endif
    const uint32_t grid = (a.n_groups + 7u) / 8u * ns * 8u * kSubTiles;   // (groups per XCD, rounded up) x sweeps x 8 XCDs x tiles of a group
    if (qv) hipLaunchKernelGGL(ingest_planes_kernel<true>, dim3(grid), dim3(kThreads), 0, st, a);
    else hipLaunchKernelGGL(ingest_planes_kernel<false>, dim3(grid), dim3(kThreads), 0, st, a);
    const uint64_t cap = (uint64_t)ctx->n_reads * ns;
    hipLaunchKernelGGL(ingest_slow_kernel, dim3(256), dim3(256), 0, st, a, d_pos, (uint32_t)std::min<uint64_t>(cap, 0xFFFFFFFFu));
}
