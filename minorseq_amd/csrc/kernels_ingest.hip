// kernels_ingest.hip — aligned BAM records -> the resident bit planes, entirely on the device.
//
// Behaviour: doc/JULIET.md:26-27 (insertions dropped, deletions '-'), :53 (PacBio cigars = X I D S H N; M is refused),
// :256-259 (a QV-filtered base shows up as N).  Reads past n_reads (the padding of a plane up to its stride) and columns
// outside a read's span are 'not covered' (code 6).
//
// Four launches at most, three for a CCS sample (the upload looks at the cigars of the few reads that could be long ones —
// jl_ingest_read_is_long — and cigar_runs_kernel is launched only if there may be one; the planes kernel's second size finds nothing
// to do on CCS reads); no by-row scratch in HBM:
//   cigar_walk_kernel   (round 6) a workgroup takes 64 reads.  One thread a read walks its cigar, sixteen ops a step in registers ->
//                       the read's RUNS in WINDOW columns (stretches of '=' / 'X' merge into one run of aligned bases; D and N are
//                       runs of their own; I / S / H / P only end a run), 8 bytes each, between a leading 'not covered from column
//                       0' entry and two trailing ones (the read's end; 'never'), so that every column of the window lies in exactly
//                       one entry's interval — into LDS.  Then a thread a (read, sweep) pair: ONE 16-byte descriptor per read and
//                       column sweep: where the sweep's entries are, how many, the dword the first 16-byte piece of packed bases
//                       the sweep needs begins on, how many pieces, the query offset of that piece — everything
//                       ingest_planes_kernel needs to ask for its input in one round trip.  Reads of more than 192 ops or 37 runs
//                       are left to
//   cigar_runs_kernel   a row of sixteen lanes a read, prefix sums over the cigar, 512 entries a read in LDS and what is beyond
//                       read back from HBM (rounds 4-5: also the first launch, with 64 entries a read).
//   ingest_planes_kernel  a workgroup = 128 reads x one sweep of 256 columns.  One request per read (the descriptor), then
//                       the entries and the pieces together.  The pieces become nibbles in QUERY order in LDS with plain
//                       16-byte stores — no run search, no masks.  Meanwhile a thread per read fills a table: for every block
//                       of 8 columns, the LDS nibble address its eight codes begin at — inside the read's bases when one
//                       aligned run covers the block, a constant dword of '-' or of 'not covered' when a deletion / nothing
//                       does; the few blocks with a run boundary inside are put together into a side dword, a thread per
//                       entry, and the table points there.  GATHER AT THE TRANSPOSE: a thread takes 32 reads x 8 columns,
//                       picks each read's dword by the table (two LDS reads + one v_alignbit), transposes four 8 x 8 nibble
//                       blocks in registers, turns BAM's base codes into symbol codes as bit planes and stores a dword of
//                       each plane of each column; the eight workgroups that share the 128-byte lines of a sweep run on one
//                       XCD next to each other (blockIdx mapping), so the lines are completed in that XCD's L2.
//                       The first launch has room for four entries a read on average; a (tile, sweep) unit with more is
//                       handed on to a second launch of the same kernel with room for sixteen (three workgroups a CU
//                       instead of five).  A read the workgroup has no room for at all — more inserted bases in a sweep than
//                       a staging row holds, more than 254 entries in a sweep, entries beyond the second size — is written
//                       column by column from its entries in HBM by a wave of the SAME workgroup behind its stores (slow_pair).
//                       With qualities a workgroup of the first size takes TWO sibling tiles one after the other and stores
//                       their plane words together, 32 bytes of a line a request (round 6: the L2 takes a write request per row
//                       and tile, 7 x 10^6 a window — the stores were 34 us of the launch's 111, 70 of 184 with qualities).
//                       (Rounds 1-3 expanded every read into a by-row matrix in HBM: 870 MB moved for the 316 MB needed;
//                       round 4 scattered the codes into a by-row LDS tile with masked, shifted XORs: 137 us.)
#include <stdlib.h>

#include <algorithm>

#include "jl_internal.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4a4 __attribute__((ext_vector_type(4), aligned(4)));

constexpr uint32_t kSweep = JL_INGEST_SWEEP;        // columns per workgroup
constexpr uint32_t kBlocks = kSweep / 8u;           // blocks of 8 columns = dwords of 8 codes
#ifndef JL_INGEST_TILE
#define JL_INGEST_TILE 128
#endif
constexpr uint32_t kTileReads = JL_INGEST_TILE;     // reads per workgroup
constexpr uint32_t kTileGroups = kTileReads / 32u;  // groups of 32 reads = dwords of a plane the tile writes per column
constexpr uint32_t kSubTiles = 1024u / kTileReads;  // tiles that share the 128-byte lines of the planes
constexpr uint32_t kThreads = 2u * kTileReads;      // a thread per read in the prologue, two in the table fill
constexpr uint32_t kRunMask = 0x3FFFFFFFu;          // window column of an entry; kind in the two bits above
static_assert(kSweep % 16u == 0 && kTileReads % 32u == 0 && kThreads <= 1024u, "whole blocks for both table threads, whole plane dwords");
static_assert(kTileGroups * kBlocks <= kThreads, "a thread per 32 reads x 8 columns in the transposing phase");

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
// Entries of read r: runs[cig_off[r] + 3 r + i], i = 0 .. n_runs + 2, each {window column | kind << 30, query offset}; entry i
// says what the read shows in the columns [its column, the next entry's column): kind 1 aligned bases (column c holds the
// base at query offset + c - column), 2 deletion, 3 nothing ('not covered': before the read, a reference skip, behind it).
// Entry 0 = {0, nothing}; entries 1 .. n_runs the read's runs, columns clamped to [0, n_cols] (a run that begins before the
// window begins at column 0 with its query offset moved along); entry n_runs + 1 = the read's end {column, nothing, query
// length}; entry n_runs + 2 = {kRunMask, nothing}: never reached.  The columns never decrease.
constexpr uint32_t kRunsReadsPerWave = 4u;   // a wave takes four reads a turn
// entries of a read kept in LDS for the descriptors: 64 — 8 KB a workgroup — in the launch that takes every read (a CCS read has
// a dozen), which lists the reads with more; 512 — 64 KB, two workgroups a CU — in the launch that takes those (and reads
// what is beyond even that back from HBM, a trip an entry: 1 ms instead of 0.12 for 100k reads with 1 % of indels, 236 ops
// a read, when they all did)
#ifndef JL_RUNS_LDS_SMALL
#define JL_RUNS_LDS_SMALL 64u
#endif
constexpr uint32_t kRunsLdsLarge = 512u;
constexpr uint32_t kRunsDeferred = 0xFFFFFFFFu;   // nruns[r] of a read the first launch (cigar_walk_kernel) leaves to the second
constexpr uint32_t kRunsLongGrid = 1024u;     // workgroups of the second launch at most: its waves take 64 reads at a time, in turns
constexpr uint32_t kDescSweeps = 15u;       // sweeps a row of sixteen lanes describes per pass (it needs sixteen bounds)
constexpr uint32_t kDescMax = 255u;         // "more than the planes kernel takes": pieces or entries of a (read, sweep)

// Descriptor of (sweep s, read r) at desc[s * n_reads + r]:
//   x  where the first 16-byte piece of packed bases begins (byte offset / 4: on a dword), low 32 bits
//   y  index of the sweep's first entry in runs[], low 32 bits
//   z  query offset of the first piece's first base (>= -6: the piece may begin before the read's bases)
//   w  x's bits 32-39 | entry index bits 32-39 << 8 | pieces << 16 | entries << 24   (255 = too many)
// The entries of a sweep: from the one that contains its first column to the first one that begins behind its last column.
// Two entries = ONE entry covers the whole sweep, the common case: then `entries` reads 1, y = the query offset of the sweep's
// first column and bits 8-9 of w = the entry's kind — the planes kernel never asks for the entries.

// The records are untrusted: a cigar with an 'M' (forbidden in PacBio BAM, doc/JULIET.md:53), one that consumes more bases
// (or qualities) than the record holds, or one that spans 2^30 reference bases or more is reported — *bad = min over such reads
// of (read << 8 | code), code 1 'M', 2 bases, 3 qualities, 4 span (5: see cigar_walk_kernel) — and the read is treated as covering nothing, so no later
// kernel follows its offsets anywhere.  Lengths add up in 64 bits (a step that holds an op of 2^24 bases or more is summed
// exactly, lane by lane), so no crafted cigar wraps a sum back into range.
#ifndef JL_RUNS_WAVES
#define JL_RUNS_WAVES 1
#endif
// The first launch takes every read, four a wave, with room for kRunsLdsSmall entries each in LDS; a read with more gets its
// entries and its count there, but not its descriptors.  The second launch (LONG) takes those reads — it finds them by their
// counts, sixty-four reads a wave at a time — with room for kRunsLdsLarge.
template <uint32_t kRunsLds, bool LONG>
__global__ __launch_bounds__(256, JL_RUNS_WAVES) void cigar_runs_kernel(uint64_t n_reads, const int32_t *__restrict__ pos, const uint32_t *__restrict__ cigar,
                                                         const uint64_t *__restrict__ cig_off, const uint64_t *__restrict__ seq_off,
                                                         const uint64_t *__restrict__ qual_off, uint32_t win_begin, uint32_t n_cols,
                                                         uint32_t n_sweeps, uint2 *__restrict__ runs, uint32_t *__restrict__ nruns,
                                                         uint4 *__restrict__ desc, unsigned long long *__restrict__ bad)
{
    // A ROW OF SIXTEEN LANES PER READ — a wave takes four reads at once — and EIGHT CONSECUTIVE OPS PER LANE (128 ops a step: a CCS
    // read's cigar in one or two steps): a lane adds its eight ops up itself, the sixteen lanes' sums are scanned within the DPP
    // row (four row_shr steps, no row broadcast), and a second pass over the lane's ops puts the entries out.  (One read at a
    // time over the whole wave, two ops a lane, was 450 wave instructions a read; the scans, the ballots and the bookkeeping
    // per step are now shared by four reads and eight ops a lane.  Several batches of four a wave, the next one's words asked
    // for ahead, measured slower: 63 against 51 us.)
    __shared__ uint2 s_run_all[4][kRunsReadsPerWave][kRunsLds];
    const uint32_t wid = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint2 (*s_run)[kRunsLds] = s_run_all[wid];
    const uint32_t q = lane >> 4, sl = lane & 15u;
    struct head_t { uint64_t cb, c_end, so, so1, qlen; int32_t pos; };
    auto ask_head = [&](uint64_t rr) -> head_t {     // (the sixteen lanes of a row ask for the same words: one request)
        head_t h = {0, 0, 0, 0, ~0ull, 0};
        if (rr < n_reads) {
            h.cb = cig_off[rr];
            h.c_end = cig_off[rr + 1];
            h.so = seq_off[rr];
            h.so1 = seq_off[rr + 1];
            if (qual_off) h.qlen = qual_off[rr + 1] - qual_off[rr];
            h.pos = pos[rr];
        }
        return h;
    };
    auto ask_words = [&](const head_t &h, uint32_t k, u32x4a4 &wa, u32x4a4 &wb) {
        const uint32_t n = (uint32_t)min(h.c_end - h.cb, (uint64_t)0xFFFFFF00u);
        const u32x4a4 z = {0, 0, 0, 0};
        wa = k < n ? *reinterpret_cast<const u32x4a4 *>(cigar + h.cb + k) : z;           // (words past the read's ops are masked where they are used)
        wb = k + 4u < n ? *reinterpret_cast<const u32x4a4 *>(cigar + h.cb + k + 4u) : z;
    };
    const uint64_t turn = (uint64_t)gridDim.x * 4u * (LONG ? 64u : kRunsReadsPerWave);
    for (uint64_t i0 = ((uint64_t)blockIdx.x * 4u + wid) * (LONG ? 64u : kRunsReadsPerWave); i0 < n_reads; i0 += turn) {
    // LONG: which of the sixty-four reads from i0 on are long ones, taken four at a time
    uint64_t todo = 1;
    if (LONG) todo = __ballot(i0 + lane < n_reads && nruns[min(i0 + lane, n_reads - 1u)] == kRunsDeferred);   // (left to this launch by cigar_walk_kernel)
    while (todo) {
    uint64_t r = i0 + q;
    if (LONG) {
        uint32_t at[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            at[t] = todo ? (uint32_t)__builtin_ctzll(todo) : 64u;
            todo &= todo - 1u;
        }
        const uint32_t mine = q == 0u ? at[0] : q == 1u ? at[1] : q == 2u ? at[2] : at[3];
        r = mine < 64u ? i0 + mine : n_reads;
    } else todo = 0;
    const bool live = r < n_reads;
    const head_t h = ask_head(r);
    u32x4a4 wa_first, wb_first;
    ask_words(h, 8u * sl, wa_first, wb_first);
    const uint64_t my_cb = h.cb, c_end = h.c_end, my_so = h.so, so1 = h.so1, qlen = h.qlen;
    const int64_t base = (int64_t)h.pos - (int64_t)win_begin;
    const uint64_t ent0 = my_cb + 3u * r;     // index of the read's entry 0 in runs[]
    uint2 *out = runs + ent0;
    const uint32_t *cig = cigar + my_cb;
    const uint32_t n_ops = (uint32_t)min(c_end - my_cb, (uint64_t)0xFFFFFF00u);   // (a record's cigar has fewer than 2^32 ops)
    uint32_t max_ops = n_ops;                  // of the wave's four reads
#pragma unroll
    for (int t = 0; t < 4; ++t) max_ops = max(max_ops, (uint32_t)__builtin_amdgcn_readlane((int)n_ops, 16 * t));
    // inclusive scan within the row of sixteen lanes
    auto row_scan = [](uint32_t v) -> uint32_t {
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);   // row_shr:1
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);   // row_shr:2
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);   // row_shr:4
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);   // row_shr:8
        return v;
    };
    auto row_last = [&](uint32_t v) -> uint32_t {   // lane 15 of the row's value, in all of its lanes
        return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(4u * (lane | 15u)), (int)v);
    };
    // window column of reference offset `ref`, clamped to the window; `before`: how far in front of it
    auto window_col = [&](uint64_t ref, uint32_t &before) -> uint32_t {
        const int64_t w = base + (int64_t)ref;      // (below 2^60: 28-bit lengths, fewer than 2^32 ops)
        before = w >= 0 ? 0u : (-w > (int64_t)0xFFFFFFFFll ? 0xFFFFFFFFu : (uint32_t)(-w));
        return w < 0 ? 0u : (w > (int64_t)n_cols ? n_cols : (uint32_t)w);
    };
    auto put = [&](uint32_t idx, uint2 e) {
        out[idx] = e;
        if (idx < kRunsLds) s_run[q][idx] = e;
    };
    uint64_t ref_total = 0, q_total = 0;       // (the row's: the same in its sixteen lanes)
    uint32_t n_runs = 0, prev_kind = 0;
    bool has_m = false;
    for (uint32_t k0 = 0; k0 < max_ops; k0 += 128u) {
        const uint32_t k = k0 + 8u * sl;
        u32x4a4 wa = wa_first, wb = wb_first;
        if (k0 != 0u) {
            const u32x4a4 z = {0, 0, 0, 0};
            wa = k < n_ops ? *reinterpret_cast<const u32x4a4 *>(cig + k) : z;           // (words past the read's ops are masked below)
            wb = k + 4u < n_ops ? *reinterpret_cast<const u32x4a4 *>(cig + k + 4u) : z;
        }
        const uint32_t w8[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
        // what an op does, from three constants indexed by its code: D N = X consume the reference, I S = X the query; kind 1
        // for = X, 2 for D, 3 for N
        constexpr uint32_t kRefOps = (1u << 2) | (1u << 3) | (1u << 7) | (1u << 8);
        constexpr uint32_t kQueryOps = (1u << 1) | (1u << 4) | (1u << 7) | (1u << 8);
        constexpr uint32_t kKinds = (2u << 4) | (3u << 6) | (1u << 14) | (1u << 16);   // two bits per op
        uint32_t kinds = 0, rsum = 0, qsum = 0, big = 0;      // kinds: two bits an op
#pragma unroll
        for (uint32_t t = 0; t < 8u; ++t) {
            const bool in = k + t < n_ops;
            const uint32_t op = w8[t] & 15u, len = in ? w8[t] >> 4 : 0u;
            kinds |= (len == 0u ? 0u : (kKinds >> (2u * op)) & 3u) << (2u * t);
            has_m = has_m || (op == 0u && in);
            rsum += ((kRefOps >> op) & 1u) ? len : 0u;
            qsum += ((kQueryOps >> op) & 1u) ? len : 0u;
            big |= len;
        }
        // the kind of the op before this lane's first one: the lane before's last (the row's first lane: the step before's)
        uint32_t before = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(kinds >> 14), 0x111, 0xF, 0xF, true);   // row_shr:1
        if (sl == 0u) before = prev_kind;
        // bit 2 t: op t begins an entry — it has a kind, and not both it and the op before are aligned bases
        const uint32_t prevs = (kinds << 2) | before;
        const uint32_t any = (kinds | (kinds >> 1)) & 0x5555u, both1 = (kinds & ~(kinds >> 1)) & (prevs & ~(prevs >> 1)) & 0x5555u;
        const uint32_t starts = any & ~both1;
        const uint32_t n_st = (uint32_t)__popc(starts);
        const uint32_t ri = row_scan(rsum), qi = row_scan(qsum), ni = row_scan(n_st);
        // the row's sums so far, in 64 bits; a step that holds an op of 2^24 bases or more is summed exactly, lane by lane (a
        // 32-bit scan of sixteen lanes of eight such ops could wrap)
        uint64_t r_step = row_last(ri), q_step = row_last(qi);
        if (__ballot(big >= (1u << 24)) != 0ull) {
            r_step = 0;
            q_step = 0;
            for (uint32_t l = 0; l < 16u; ++l) {
                r_step += (uint32_t)__builtin_amdgcn_ds_bpermute((int)(4u * ((lane & 48u) | l)), (int)rsum);
                q_step += (uint32_t)__builtin_amdgcn_ds_bpermute((int)(4u * ((lane & 48u) | l)), (int)qsum);
            }
        }
        // the entries
        uint64_t ref_at = ref_total + (ri - rsum), q_at = q_total + (qi - qsum);
        uint32_t idx = 1u + n_runs + (ni - n_st);
#pragma unroll
        for (uint32_t t = 0; t < 8u; ++t) {
            const uint32_t op = w8[t] & 15u, len = k + t < n_ops ? w8[t] >> 4 : 0u;
#if defined(JL_RUNS_PROBE) && JL_RUNS_PROBE >= 2     // (tuning: ... and without the entries)
            if (false)
#else
            if ((starts >> (2u * t)) & 1u)
#endif
            {
                uint32_t bf;
                const uint32_t w = window_col(ref_at, bf);
                put(idx, make_uint2(w | (((kinds >> (2u * t)) & 3u) << 30), (uint32_t)q_at + bf));
                ++idx;
            }
            ref_at += ((kRefOps >> op) & 1u) ? len : 0u;
            q_at += ((kQueryOps >> op) & 1u) ? len : 0u;
        }
        n_runs += row_last(ni);
        prev_kind = row_last(kinds >> 14);   // (only a full step has a successor)
        ref_total += r_step;
        q_total += q_step;
    }
    // M anywhere in the row?  (a ballot's sixteen bits)
    const uint64_t bm = __ballot(has_m);
    uint32_t code = ((bm >> (16u * q)) & 0xFFFFull) != 0ull ? 1u : 0u;
    if (!code) {
        if (q_total > 2u * (so1 - my_so) || q_total > 0x7FFFFFFFull) code = 2u;
        else if (q_total > qlen) code = 3u;
        else if (ref_total > (uint64_t)kRunMask) code = 4u;
    }
    if (!live) code = 0;
    uint32_t end_col = 0;
    if (code) {
        n_runs = 0;
        if (sl == 0u) atomicMin(bad, ((unsigned long long)r << 8) | code);
    } else {
        uint32_t bf;
        end_col = window_col(ref_total, bf);
    }
    // a read with more entries than the LDS copy holds: for the second launch (its entries are written here all the same)
    const bool is_long = !LONG && live && n_runs + 3u > kRunsLds;
    if (live && sl < 3u) {
        const uint32_t idx = sl == 0u ? 0u : n_runs + sl;
        const uint2 e = sl == 0u ? make_uint2(3u << 30, 0u) : sl == 1u ? make_uint2(end_col | (3u << 30), (uint32_t)q_total) : make_uint2(kRunMask | (3u << 30), 0u);
        put(idx, e);
        if (sl == 0u) nruns[r] = n_runs;
    }
#if defined(JL_RUNS_PROBE) && JL_RUNS_PROBE >= 1     // (tuning: the kernel without its descriptors — wrong results by design)
    if (n_reads) continue;
#endif
    // ---- the descriptors: a lane per sweep (fifteen sweeps a pass: a sweep needs the bound of the next one too).  Entries
    // beyond the LDS copy are read back from HBM: past this wave's own stores.
    const uint32_t n_mine = is_long ? 0u : n_runs;
    uint32_t n_max = n_mine;
#pragma unroll
    for (int t = 0; t < 4; ++t) n_max = max(n_max, (uint32_t)__builtin_amdgcn_readlane((int)n_mine, 16 * t));
    if (LONG && n_max + 3u > kRunsLds) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_wave_barrier();
    const uint32_t n_ent_all = n_runs + 3u;          // entries of this lane's read
    uint32_t top = 1u;                         // the highest power of two not above the longest list
    while (2u * top <= n_max + 3u) top *= 2u;
    // (`entry`: the read's i-th entry — out of the wave's LDS copy, or, for the lists that are longer than it, read back)
    auto describe = [&](auto entry) {
        for (uint32_t s0 = 0; s0 < n_sweeps; s0 += kDescSweeps) {
            const uint32_t s = s0 + sl;
            // f = the number of entries that begin at or before the sweep's first column (entry 0 always does, the last never)
            const uint32_t X = min(s, n_sweeps) * kSweep;   // (lanes past the last bound repeat it)
            // The bound BEHIND the window's last sweep is the window's last column, not n_sweeps * kSweep: every run beyond the
            // window is clamped to column n_cols, and counted up to there all of a long read's runs behind the window were
            // entries of its last sweep (six of them and the unit went to the second size, 255 and the read to slow_pair: right
            // cells, but `juliet --windows N` paid for it in every window).  Up to n_cols - 1 the first clamped entry closes the list.
            const uint32_t X_bound = s >= n_sweeps ? n_cols - 1u : X;
            uint32_t f = 0;
            for (uint32_t step = top; step; step >>= 1) {
                const uint32_t t = f + step;
                if (live && !is_long && t <= n_ent_all && (entry(t - 1u).x & kRunMask) <= X_bound) f = t;
            }
            const uint32_t f_next = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)f, 0x101, 0xF, 0xF, false);   // row_shl:1
            if (!live || is_long || sl == kDescSweeps || s >= n_sweeps) continue;
            const uint32_t Xend = min(n_cols, X + kSweep);
            const uint32_t lo = f - 1u;
            uint32_t n_ent = f_next - f + 2u;
            uint32_t q_lo = 0xFFFFFFFFu, q_hi = 0;
            uint2 e = entry(lo);
            const uint2 first = e;
            if (n_ent >= kDescMax) n_ent = kDescMax;
            else {
                for (uint32_t i = lo; i + 1u < lo + n_ent; ++i) {
                    const uint2 nx = entry(i + 1u);
                    if ((e.x >> 30) == 1u) {
                        const uint32_t W = e.x & kRunMask, ca = max(W, X), cbv = min(nx.x & kRunMask, Xend);
                        if (ca < cbv) {
                            q_lo = min(q_lo, e.y + (ca - W));
                            q_hi = max(q_hi, e.y + (cbv - W));
                        }
                    }
                    e = nx;
                }
            }
            // the 16-byte pieces of packed bases the sweep takes: from the DWORD that holds its first base on — at most seven bases
            // before it, so that a row of the planes kernel (a sweep + 32 bases) has room for 25 inserted ones (byte offsets are
            // even in bases: all of this in 32 bits relative to the read's first byte, the one 64-bit sum at the end)
            uint64_t piece = 0;
            uint32_t np = 0;
            int32_t q0 = 0;
            if (q_lo < q_hi) {
                const uint32_t al = (uint32_t)my_so & 3u;                        // the read's first byte within its dword
                const uint32_t b_lo = al + (q_lo >> 1), b_hi = al + (q_hi + 1u) / 2u;   // bytes from that dword's first on
                const uint32_t d_lo = b_lo >> 2;
                piece = (my_so >> 2) + d_lo;
                np = min((b_hi + 15u - 4u * d_lo) >> 4, kDescMax);
                q0 = 2 * ((int32_t)(4u * d_lo) - (int32_t)al);
            }
            const uint64_t e_idx = ent0 + lo;
            uint4 d;
            d.x = (uint32_t)piece;
            d.z = (uint32_t)q0;
            if (n_ent == 2u) {
                // ONE entry covers the whole sweep (three reads in four of a CCS sample): the planes kernel needs no entries for it —
                // what the entry is, and the query offset of the sweep's first column
                d.y = first.y + (X - (first.x & kRunMask));
                d.w = (uint32_t)((piece >> 32) & 0xFFu) | ((first.x >> 30) << 8) | (np << 16) | (1u << 24);
            } else {
                d.y = (uint32_t)e_idx;
                d.w = (uint32_t)((piece >> 32) & 0xFFu) | ((uint32_t)((e_idx >> 32) & 0xFFu) << 8) | (np << 16) | (n_ent << 24);
            }
            desc[(uint64_t)s * n_reads + r] = d;
        }
    };
    const uint2 *s_mine = s_run[q];
    if (!LONG || n_max + 3u <= kRunsLds) describe([&](uint32_t i) -> uint2 { return s_mine[i]; });
    else
        describe([&](uint32_t i) -> uint2 {
            if (i < kRunsLds) return s_mine[i];
            const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(runs + ent0 + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
        });
    __builtin_amdgcn_wave_barrier();      // (the next four reads write the LDS copy these ones' descriptors were made from)
    }
    if (!LONG) break;                     // (the first launch has a wave for every four reads)
    }
}

// ---------------------------------------------------------------------------------------- runs, a thread a read
// The first launch of a build (round 6; it replaces cigar_runs_kernel<64, false>, which gave a row of sixteen lanes to every read:
// 24 x 10^6 wave instructions and 47-50 us for 100k reads of 127 ops, 36 us for the ten ops of a `ccs --richQVs` read — the scans,
// the LDS copy of the entries and a binary search per sweep done by every row, for lists a dozen entries long).  A workgroup takes
// 64 reads:
//   1  ONE THREAD a read (the first wave): the ops, sixteen a step in registers (four 16-byte loads in flight, the next step's
//      asked for before this one's are looked at) -> the read's ENTRIES in LDS.  No store to HBM in this pass: vmcnt counts loads
//      and stores in one order, and forms of this kernel that stored entries while walking waited for every store's
//      acknowledgement at each new load of ops (32-61 us for ten ops, 144-720 us for 127).
//   2  a thread a (read, sweep) pair, all four waves — the 64 lanes of a wave write one sweep's descriptors of the 64 reads, 1 KB a
//      store: the last entry that begins at or before the sweep's first column by bisection, then the sweep's entries up to the
//      first that begins behind its last column.  Then the entries go out.
// (One thread a read for BOTH passes — a dozen descriptors one after the other — was 25 us for ten ops: a wave is alone on its
// SIMD then, 1563 waves for 1024 SIMDs, and the length of the dependent chain is all that counts.)
// A read with more than kWalkOps ops or more than kWalkEnt entries is left, whole, to the launch with a row of lanes a read:
// nruns[r] = kRunsDeferred.
constexpr uint32_t kWalkOps = 192u;
#ifndef JL_WALK_ENT
#define JL_WALK_ENT 40
#endif
// entries of a read in LDS: a CCS read has a dozen.  With 24 two reads in a thousand went to the other launch; with 32 two in 100 000
// (fifteen deletions) — enough for the upload to have to ask for that launch in every build; with 40 none of a CCS sample does
// (48: 26 KB of LDS a workgroup, six a CU — not all 1563 of a 100k-read build at once: 28 us instead of 22)
constexpr uint32_t kWalkEnt = JL_WALK_ENT;
constexpr uint32_t kWalkReads = 64u;    // reads of a workgroup

__global__ __launch_bounds__(256) void cigar_walk_kernel(uint64_t n_reads, const int32_t *__restrict__ pos, const uint32_t *__restrict__ cigar,
                                                         const uint64_t *__restrict__ cig_off, const uint64_t *__restrict__ seq_off,
                                                         const uint64_t *__restrict__ qual_off, uint32_t win_begin, uint32_t n_cols,
                                                         uint32_t n_sweeps, uint2 *__restrict__ runs, uint32_t *__restrict__ nruns,
                                                         uint4 *__restrict__ desc, unsigned long long *__restrict__ bad,
                                                         uint32_t *__restrict__ count, uint32_t second)
{
    // second = 0: the host has looked at the cigars (jl_ingest_read_is_long, at the upload) and found no read for the other launch, which
    // is then not made (6 us of a build).  Should this kernel find one all the same, the read covers nothing and the build fails with
    // code 5 — never descriptors nobody wrote.
    // (the build's counters — [0] pairs listed, [1] units handed on — begin at zero: the planes kernels, which count, come behind this
    // launch on the stream; a launch of its own for this was 4 us of a build)
    if (blockIdx.x == 0 && threadIdx.x < 2u) count[threadIdx.x] = 0u;
    __shared__ uint2 s_ent[kWalkEnt * kWalkReads];      // entry i of read j at [i * 64 + j]
    __shared__ uint32_t s_nruns[kWalkReads];            // kRunsDeferred: no descriptors from here
    __shared__ uint64_t s_ent0[kWalkReads], s_so[kWalkReads];
    const uint32_t tid = threadIdx.x, j = tid & (kWalkReads - 1u);
    const uint64_t r = (uint64_t)blockIdx.x * kWalkReads + j;
    if (tid < kWalkReads) {
        uint32_t n_runs = kRunsDeferred;
        if (r < n_reads) {
            const uint64_t my_cb = cig_off[r], c_end = cig_off[r + 1], my_so = seq_off[r], so1 = seq_off[r + 1];
            const uint64_t qlen = qual_off ? qual_off[r + 1] - qual_off[r] : ~0ull;
            const int64_t base = (int64_t)pos[r] - (int64_t)win_begin;
            const uint32_t n_ops = (uint32_t)min(c_end - my_cb, (uint64_t)0xFFFFFF00u);
            s_ent0[j] = my_cb + 3u * r;
            s_so[j] = my_so;
            if (n_ops > kWalkOps && !second) {
                atomicMin(bad, ((unsigned long long)r << 8) | 5u);
                n_runs = 0;
                s_ent[j] = make_uint2(3u << 30, 0u);
                s_ent[kWalkReads + j] = make_uint2(3u << 30, 0u);
                s_ent[2u * kWalkReads + j] = make_uint2(kRunMask | (3u << 30), 0u);
            }
            if (n_ops <= kWalkOps) {
                const uint32_t *cig = cigar + my_cb;
                auto window_col = [&](uint64_t ref, uint32_t &before) -> uint32_t {
                    const int64_t w = base + (int64_t)ref;
                    before = w >= 0 ? 0u : (-w > (int64_t)0xFFFFFFFFll ? 0xFFFFFFFFu : (uint32_t)(-w));
                    return w < 0 ? 0u : (w > (int64_t)n_cols ? n_cols : (uint32_t)w);
                };
                constexpr uint32_t kRefOps = (1u << 2) | (1u << 3) | (1u << 7) | (1u << 8);
                constexpr uint32_t kQueryOps = (1u << 1) | (1u << 4) | (1u << 7) | (1u << 8);
                constexpr uint32_t kKinds = (2u << 4) | (3u << 6) | (1u << 14) | (1u << 16);   // two bits per op
                // (the loads take sixteen words from the step's first op on whatever the read's length: words past its ops are the
                // next read's or the array's slack, and are not looked at)
                struct ops16 { u32x4a4 a, b, c, d; };
                auto ask = [&](uint32_t k0) -> ops16 {
                    ops16 o;
                    const uint32_t *p = cig + k0;
                    o.a = *reinterpret_cast<const u32x4a4 *>(p);
                    o.b = *reinterpret_cast<const u32x4a4 *>(p + 4);
                    o.c = *reinterpret_cast<const u32x4a4 *>(p + 8);
                    o.d = *reinterpret_cast<const u32x4a4 *>(p + 12);
                    return o;
                };
                uint64_t ref_at = 0, q_at = 0;
                uint32_t prev_kind = 0;
                bool has_m = false;
                n_runs = 0;
                s_ent[j] = make_uint2(3u << 30, 0u);
                ops16 cur = ask(0);
                for (uint32_t k0 = 0; k0 < n_ops; k0 += 16u) {
                    const ops16 nxt = ask(k0 + 16u < n_ops ? k0 + 16u : 0u);
                    const uint32_t w16[16] = {cur.a.x, cur.a.y, cur.a.z, cur.a.w, cur.b.x, cur.b.y, cur.b.z, cur.b.w,
                                              cur.c.x, cur.c.y, cur.c.z, cur.c.w, cur.d.x, cur.d.y, cur.d.z, cur.d.w};
#pragma unroll
                    for (uint32_t t = 0; t < 16u; ++t) {
                        const bool in = k0 + t < n_ops;
                        const uint32_t op = w16[t] & 15u, len = in ? w16[t] >> 4 : 0u;
                        const uint32_t kind = len == 0u ? 0u : (kKinds >> (2u * op)) & 3u;
                        has_m = has_m || (in && op == 0u);
                        if (kind != 0u && !(kind == 1u && prev_kind == 1u)) {     // an entry begins: not both this op and the one before are aligned bases
                            uint32_t bf;
                            const uint32_t col = window_col(ref_at, bf);
                            ++n_runs;
                            if (n_runs < kWalkEnt) s_ent[n_runs * kWalkReads + j] = make_uint2(col | (kind << 30), (uint32_t)q_at + bf);
                        }
                        if (in) prev_kind = kind;
                        ref_at += ((kRefOps >> op) & 1u) ? len : 0u;
                        q_at += ((kQueryOps >> op) & 1u) ? len : 0u;
                    }
                    cur = nxt;
                }
                if (n_runs + 3u > kWalkEnt && second) n_runs = kRunsDeferred;
                else {
                    uint32_t code = n_runs + 3u > kWalkEnt ? 5u : has_m ? 1u : 0u;
                    if (!code) {
                        if (q_at > 2u * (so1 - my_so) || q_at > 0x7FFFFFFFull) code = 2u;
                        else if (q_at > qlen) code = 3u;
                        else if (ref_at > (uint64_t)kRunMask) code = 4u;
                    }
                    uint32_t end_col = 0, bf;
                    if (code) {      // a malformed record covers nothing (see cigar_runs_kernel)
                        atomicMin(bad, ((unsigned long long)r << 8) | code);
                        n_runs = 0;
                    } else end_col = window_col(ref_at, bf);
                    s_ent[(n_runs + 1u) * kWalkReads + j] = make_uint2(end_col | (3u << 30), (uint32_t)q_at);
                    s_ent[(n_runs + 2u) * kWalkReads + j] = make_uint2(kRunMask | (3u << 30), 0u);
                }
            }
            nruns[r] = n_runs;
        }
        s_nruns[j] = n_runs;
    }
    __syncthreads();
    const uint32_t n_runs = s_nruns[j];
    if (n_runs == kRunsDeferred) return;       // (no read, or one for the other launch; no barrier follows)
    const uint32_t n_ent_all = n_runs + 3u;
    const uint64_t ent0 = s_ent0[j], my_so = s_so[j];
    auto ent = [&](uint32_t i) -> uint2 { return s_ent[i * kWalkReads + j]; };
    // ---- 2: the descriptors, sweeps wave, wave + 4, ... of the 64 reads (see cigar_runs_kernel's `describe`)
    for (uint32_t s = tid >> 6; s < n_sweeps; s += 4u) {
        const uint32_t X = s * kSweep, Xend = min(n_cols, X + kSweep);
        // (the window's last sweep ends at its last column: the runs behind the window are all clamped to column n_cols)
        const uint32_t bound = s + 1u < n_sweeps ? X + kSweep : n_cols - 1u;
        // lo: the last entry that begins at or before X (entry 0 does, the last one never)
        uint32_t lo = 0, hi = n_ent_all - 1u;
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if ((ent(mid).x & kRunMask) <= X) lo = mid;
            else hi = mid;
        }
        const uint2 first = ent(lo);
        uint2 e = first;
        uint32_t q_lo = 0xFFFFFFFFu, q_hi = 0, i = lo;
        for (;;) {
            ++i;
            const uint2 nx = ent(i);
            if ((e.x >> 30) == 1u) {
                const uint32_t W = e.x & kRunMask, ca = max(W, X), cbv = min(nx.x & kRunMask, Xend);
                if (ca < cbv) {
                    q_lo = min(q_lo, e.y + (ca - W));
                    q_hi = max(q_hi, e.y + (cbv - W));
                }
            }
            if ((nx.x & kRunMask) > bound) break;      // it begins behind the sweep: the closing entry
            e = nx;
        }
        const uint32_t n_ent = i - lo + 1u;      // (< kWalkEnt < kDescMax)
        uint64_t piece = 0;
        uint32_t np = 0;
        int32_t q0 = 0;
        if (q_lo < q_hi) {
            const uint32_t al = (uint32_t)my_so & 3u;
            const uint32_t b_lo = al + (q_lo >> 1), b_hi = al + (q_hi + 1u) / 2u;
            const uint32_t d_lo = b_lo >> 2;
            piece = (my_so >> 2) + d_lo;
            np = min((b_hi + 15u - 4u * d_lo) >> 4, kDescMax);
            q0 = 2 * ((int32_t)(4u * d_lo) - (int32_t)al);
        }
        const uint64_t at = ent0 + lo;
        uint4 d;
        d.x = (uint32_t)piece;
        d.z = (uint32_t)q0;
        if (n_ent == 2u) {      // ONE entry covers the whole sweep: what it is, and the query offset of the sweep's first column
            d.y = first.y + (X - (first.x & kRunMask));
            d.w = (uint32_t)((piece >> 32) & 0xFFu) | ((first.x >> 30) << 8) | (np << 16) | (1u << 24);
        } else {
            d.y = (uint32_t)at;
            d.w = (uint32_t)((piece >> 32) & 0xFFu) | ((uint32_t)((at >> 32) & 0xFFu) << 8) | (np << 16) | (n_ent << 24);
        }
        desc[(uint64_t)s * n_reads + r] = d;
    }
    // ---- the entries: thread (read, part) writes entries part, part + 4, ...
    uint2 *out = runs + ent0;
    for (uint32_t i = tid >> 6; i < n_ent_all; i += 4u) out[i] = ent(i);
}

// ---------------------------------------------------------------------------------------- the planes of one sweep
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

// 32 reads x 8 columns, R[i] = the 8 nibbles of read i -> out[j][k] = plane k of column j, bit i = read i.
// Four 8 x 8 nibble transposes — block g holds reads g, g + 4, ..., g + 28, so that after it nibble n of M[g][j] is read
// 4 n + g at column j — then the bits of the four blocks' nibbles change places: the four bit planes of the nibbles.
// A read's nibbles are symbol codes already where its bit in `codes` is set (planes 0-2 are the answer); everywhere else they
// are BAM's 4-bit bases, one plane per letter A C G T, and the symbol code is made HERE, 32 reads an instruction: exactly one
// letter -> its index, anything else (N, '=', the ambiguity codes) -> 5, a filtered base.
__device__ __forceinline__ void nibble_rows_to_plane_words(const uint32_t (&R)[32], uint32_t codes, uint32_t (&out)[8][3])
{
    uint32_t M[4][8];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int i = 0; i < 8; ++i) M[g][i] = R[4 * i + g];
        transpose_nibbles_8x8(M[g]);
    }
    // Plane k of column j wants, in nibble n, bit k of the four blocks' nibbles n (reads 4 n .. 4 n + 3): a 4 x 4 bit transpose
    // between the four words, in every nibble at once — two rounds of masked swaps.
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        uint32_t a0 = M[0][j], a1 = M[1][j], a2 = M[2][j], a3 = M[3][j], t;
        t = ((a0 >> 1) ^ a1) & 0x55555555u; a1 ^= t; a0 ^= t << 1;
        t = ((a2 >> 1) ^ a3) & 0x55555555u; a3 ^= t; a2 ^= t << 1;
        t = ((a0 >> 2) ^ a2) & 0x33333333u; a2 ^= t; a0 ^= t << 2;
        t = ((a1 >> 2) ^ a3) & 0x33333333u; a3 ^= t; a1 ^= t << 2;
        // (three-input boolean functions are one v_bitop3 each)
        const uint32_t one3 = (a0 ^ a1 ^ a2) & ~(a0 & a1 & a2), none3 = ~(a0 | a1 | a2);   // exactly one / none of A C G
        const uint32_t letter = (one3 & ~a3) | (none3 & a3);                               // exactly one of A C G T
        const uint32_t c0 = ~letter | a1 | a3, c1 = letter & (a2 | a3);                    // C T N odd; G T have bit 1; N has bit 2
        out[j][0] = (codes & a0) | (~codes & c0);
        out[j][1] = (codes & a1) | (~codes & c1);
        out[j][2] = (codes & a2) | (~codes & ~letter);
    }
}

struct ingest_args {
    uint64_t n_reads;
    uint32_t n_cols, n_sweeps, n_groups;   // n_groups: groups of 1024 reads (the tiles that share lines)
    uint32_t min_qv;
    const uint64_t *cig_off;     // (slow_pair: where a read's entries begin)
    const uint8_t *seq4;
    const uint64_t *seq_off;     // (slow_pair)
    const uint8_t *qual;         // null: no QV masking
    const uint64_t *qual_off;
    const uint2 *runs;
    const uint32_t *nruns;
    const uint4 *desc;
    uint32_t *slow_count;        // the 64-byte block of counters: [1] units handed on
    uint32_t *big_list;          // the units (block numbers) handed on to the kernel's second size; their number in slow_count[1]
    uint8_t *msa;
    uint64_t plane_stride;
    uint64_t seq_bytes, n_entries;   // (tuning builds check every address a descriptor leads to against these and report in dbg[])
    uint32_t *dbg;
    unsigned long long *stamps;  // (tuning builds)
    uint32_t skip;               // tuning builds: bit 0 no bases, bit 1 no table, bit 2 no stores, bit 3 no general pass, bit 4 no transposing, bit 6 no conversion
};
#ifdef JL_TUNING
#define JL_ING_SKIP(a, bit) (((a).skip >> (bit)) & 1u)
// an address that would leave its array: counted in dbg[code], the offending value kept in dbg[5 + code]; the access is redirected
#define JL_ING_CHECK(a, ok, code, value, fix) \
    if (!(ok)) {                               \
        atomicAdd(&(a).dbg[code], 1u);         \
        (a).dbg[5 + (code)] = (uint32_t)(value); \
        fix;                                   \
    }
// wall-clock stamps (10 ns) of every 61st workgroup's waves at the phase boundaries, in the (empty) slow list's memory
#define JL_ING_STAMP(a, k)                                                                                        \
    if ((a).stamps && blockIdx.x % 61u == 0u && blockIdx.x / 61u < 160u && (threadIdx.x & 63u) == 0u)           \
        (a).stamps[((blockIdx.x / 61u) * 4u + (threadIdx.x >> 6)) * 12u + (k)] = wall_clock64();
#else
#define JL_ING_SKIP(a, bit) 0u
#define JL_ING_CHECK(a, ok, code, value, fix)
#define JL_ING_STAMP(a, k)
#endif

// LDS of a planes workgroup.  The staging area holds dwords of eight codes; a NIBBLE address into it fits 16 bits:
//   dwords 0-1 'not covered' twice, 2-3 '-' twice (what a table entry of a block nothing / a deletion covers points at),
//   4 .. 4 + kEntCap   the blocks with a boundary inside, put together (a dword for every entry: the block it begins inside),
//   kRowBase ..        the reads' codes in query order: a row of kRowPieces 16-byte pieces per read — a sweep's bases from a
//                      16-byte boundary on, + 8 dwords between the 32-read groups, so that the four reads a wave gathers from
//                      at a time lie 8 banks apart.
#ifndef JL_INGEST_ENT_PER_READ
#define JL_INGEST_ENT_PER_READ 4
#endif
#ifndef JL_INGEST_ENT_PER_READ_BIG
#define JL_INGEST_ENT_PER_READ_BIG 16
#endif
constexpr uint32_t kReadWaves = kTileReads / 64u;     // waves of read threads, with a part of the entry area each
#ifndef JL_INGEST_MIN_WGS
#define JL_INGEST_MIN_WGS 4               // workgroups a CU must be able to hold: 128 registers for the first size (two tiles a workgroup with qualities: 129 without the bound)
#endif
#ifndef JL_INGEST_NT
#define JL_INGEST_NT 1                    // sibling tiles a workgroup of the planes kernel's first size takes (1, 2, 4); with qualities:
#endif
#ifndef JL_INGEST_NT_QV
#define JL_INGEST_NT_QV 2
#endif
#ifndef JL_INGEST_QUAL_AHEAD
#define JL_INGEST_QUAL_AHEAD 3            // pieces whose qualities a thread has asked for ahead of their turn (8 registers each; with all
                                          // seven ahead one tile a workgroup is 5 us faster, two tiles a workgroup 11 us slower)
#endif
#ifndef JL_INGEST_ROW_EXTRA
#define JL_INGEST_ROW_EXTRA 0             // pieces of a row beyond a sweep's own (room for inserted bases: 32 a piece)
#endif
constexpr uint32_t kRowPieces = (kSweep + 31u) / 32u + 1u + JL_INGEST_ROW_EXTRA, kRowDw = 4u * kRowPieces, kGroupPadDw = 8u;
// The kernel comes in two sizes of its entry area (entries of the reads that need them, 4 bytes each in LDS: column - sweep's first
// (0..256: 9 bits) | kind << 9 | (query offset there - the row's first) << 11): four entries a read on average — a CCS read has 4
// in a sweep with an indel — with five workgroups to a CU, and sixteen, with three: for the (tile, sweep) units whose reads have
// more (an indel every 50 columns), which the first size hands on (ingest_args::big_list) instead of leaving read after read
// to the column-by-column kernel (7 ms for 100k reads with 1 % of indels).
template <uint32_t EPR>
struct planes_shape {
    static constexpr uint32_t kEntCap = EPR * kTileReads, kEntCapWave = kEntCap / kReadWaves;
    static constexpr uint32_t kRowBase = 4u + kEntCap;     // (a side dword per entry: an entry begins inside one block at most)
    static constexpr uint32_t kStageDw = kRowBase + kRowDw * kTileReads + kGroupPadDw * kTileGroups + 4u;
    static __device__ __forceinline__ uint32_t row_dw(uint32_t j) { return kRowBase + kRowDw * j + kGroupPadDw * (j >> 5); }
    static_assert(kRowBase % 4u == 0 && kStageDw * 8u <= 65536u, "16-byte pieces; 16-bit nibble addresses");
};
// the table: 32 entries (16 bits) a read — rows on 8-byte boundaries for the four-entries-at-a-time stores —
// and 8 dwords of padding per 32 reads: the four 32-read groups of a wave's lanes read it 8 banks apart
constexpr uint32_t kTabRow = kBlocks > 16u ? 32u : 16u, kTabGroupPad = 16u;
static_assert(kBlocks <= kTabRow && kBlocks % 4u == 0, "a sweep is at most 32 blocks wide, whole chunks of four");
constexpr uint32_t kTabSize = kTileReads * kTabRow + kTileGroups * kTabGroupPad;
__device__ __forceinline__ uint32_t tab_row(uint32_t j) { return j * kTabRow + (j >> 5) * kTabGroupPad; }
// ... and within a row the chunks of four blocks change places by the read's number: the rows are 16 dwords apart, so that sixty-four
// reads storing the same chunk of their rows met in two bank phases (8 b64 stores of a one-entry row took a sixteenth of their rate:
// 13 of the kernel's 21 x 10^6 LDS cycles were bank conflicts)
__device__ __forceinline__ uint32_t tab_at(uint32_t j, uint32_t blk) { return tab_row(j) + ((((blk >> 2) ^ j) & (kTabRow / 4u - 1u)) << 2) + (blk & 3u); }
static_assert((kTabRow / 4u & (kTabRow / 4u - 1u)) == 0, "chunks of a row: a power of two");
static_assert(kTileReads % 64u == 0 && kThreads == 2u * kTileReads && kBlocks <= 32u, "whole waves of read threads; a block index fits five bits");
// the pieces of a tile: the upper half of the threads take kPieceRoundsB rounds of kTileReads pieces, the read threads — who have
// their rows of the table to make as well — the rest
constexpr uint32_t kPieceRoundsA = kRowPieces >= 8u ? 2u : 1u, kPieceRoundsB = kRowPieces - kPieceRoundsA;


// Eight of BAM's 4-bit bases, one per nibble -> eight symbol codes: A C G T (1 2 4 8) -> 0..3, everything else (N = 15, '=' = 0, the
// IUPAC ambiguity codes) -> N, a filtered base.  v_perm_b32 looks four bytes up in an 8-byte table: the low three bits of a
// base select in the table of the codes 0..7 and in that of 8..15, bit 3 picks between the two results.  (Only the few blocks
// with a run boundary inside go through this: everything else is converted as bit planes, behind the transposition.)
__device__ __forceinline__ uint32_t codes_of_bases8(uint32_t x)
{
    constexpr uint32_t kLo03 = 0x05010005u, kLo47 = 0x05050502u;   // codes of 0..3 (bytes 0..3), of 4..7
    constexpr uint32_t kHi03 = 0x05050503u, kHi47 = 0x05050505u;   // codes of 8..11, of 12..15
    auto lookup4 = [&](uint32_t y) -> uint32_t {   // four bases, one per byte (0..15) -> four codes
        const uint32_t sel = y & 0x07070707u;
        const uint32_t lo = __builtin_amdgcn_perm(kLo47, kLo03, sel), hi = __builtin_amdgcn_perm(kHi47, kHi03, sel);
        const uint32_t b3 = y & 0x08080808u, pick = (b3 << 5) - (b3 >> 3);   // bytes of ones where bit 3 is set
        return (hi & pick) | (lo & ~pick);
    };
    return lookup4(x & 0x0F0F0F0Fu) | (lookup4((x >> 4) & 0x0F0F0F0Fu) << 4);
}

// 16 bytes of packed bases (BAM order: first base in the high nibble) -> the same 32 bases in QUERY order, base b in nibble
// b & 7 of S[b >> 3] — BAM's codes as they are: they become symbol codes as bit planes, behind the transposition, at an
// eighth of the price per base.  QV: bases whose quality is below min_qv become 15 (N).  Q = query offset of the piece's base 0
// (negative: the first -Q bases are not the read's own).
// The qualities of a piece's 32 bases, one byte each: two 16-byte loads at the BYTE address of the first (gfx950 takes unaligned
// dwordx4 loads; Q < 0 — the piece begins up to six bases before the read's own — reads the read before's last bytes, or the 16
// bytes in front of the first read's (jl_records_begin): those bases are nobody's, whatever mask they get).
struct piece_quals { u32x4 a, b; };
__device__ __forceinline__ piece_quals ask_quals(const ingest_args &a, int Q, uint64_t qual_base)
{
    typedef uint32_t u32x4a1 __attribute__((ext_vector_type(4), aligned(1)));
    const uint8_t *qp = a.qual + (int64_t)qual_base + (int64_t)Q;
    piece_quals q;
#ifdef JL_TUNING
    if (JL_ING_SKIP(a, 10)) {      // (probe: streaming loads)
        q.a = __builtin_nontemporal_load(reinterpret_cast<const u32x4a1 *>(qp));
        q.b = __builtin_nontemporal_load(reinterpret_cast<const u32x4a1 *>(qp + 16));
        return q;
    }
#endif
    q.a = *reinterpret_cast<const u32x4a1 *>(qp);
    q.b = *reinterpret_cast<const u32x4a1 *>(qp + 16);
    return q;
}

// 32 bases in query order (S) and their qualities (pq): a base is masked — becomes 15, N — when its quality is below min_qv (0xFF =
// absent, and anything above 127, never masks).  Eight bases a step, the byte-parallel compare done on the even and on the odd
// bases' bytes apart (v_perm), so that the two results interleave into one flag per NIBBLE — the bases' own layout — with one
// shift: 13 instructions for eight bases (round 5: nine dword loads, eight v_alignbyte, flags gathered into bits and spread again: 30).
__device__ __forceinline__ void mask_low_quals(const ingest_args &a, const piece_quals &pq, uint32_t (&S)[4])
{
    const uint32_t q8[8] = {pq.a.x, pq.a.y, pq.a.z, pq.a.w, pq.b.x, pq.b.y, pq.b.z, pq.b.w};
    const uint32_t T = a.min_qv * 0x01010101u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t ev = __builtin_amdgcn_perm(q8[2 * k + 1], q8[2 * k], 0x06040200u);   // qualities of bases 0 2 4 6 of the eight
        const uint32_t od = __builtin_amdgcn_perm(q8[2 * k + 1], q8[2 * k], 0x07050301u);   // of bases 1 3 5 7
        const uint32_t lt_e = ~((ev | 0x80808080u) - T) & ~ev & 0x80808080u;                // bit 7 of byte i: base 2 i is below min_qv
        const uint32_t lt_o = ~((od | 0x80808080u) - T) & ~od & 0x80808080u;
        const uint32_t c = lt_o | (lt_e >> 4);          // bit 3 of nibble j: base j is masked
        S[k] |= c | (c - (c >> 3));                     // 8 -> 15 (N) in those nibbles
    }
}

template <bool QV>
__device__ __forceinline__ void piece_bases(const ingest_args &a, const u32x4 &v, const piece_quals &pq, uint32_t (&S)[4])
{
    const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) S[k] = ((w4[k] >> 4) & 0x0F0F0F0Fu) | ((w4[k] & 0x0F0F0F0Fu) << 4);
    if (QV) mask_low_quals(a, pq, S);
}

__device__ __forceinline__ uint32_t ent_col(uint32_t e) { return e & 511u; }
__device__ __forceinline__ uint32_t ent_kind(uint32_t e) { return (e >> 9) & 3u; }
// staging nibble address of column c (relative to the sweep's first) in entry e of a read whose row begins at nibble row8
__device__ __forceinline__ uint32_t ent_addr(uint32_t e, uint32_t row8, uint32_t c)
{
    const uint32_t kind = ent_kind(e);
    return kind == 1u ? row8 + (e >> 11) + (c - ent_col(e)) : kind == 2u ? 16u : 0u;
}

// ---------------------------------------------------------------------------------------- what a tile cannot take
// A (read, sweep) pair with more pieces or entries than a workgroup has room for, by ONE WAVE of the tile's own workgroup,
// behind its stores: a lane per column looks its entry up in HBM and flips the bits in which the symbol differs from 'not
// covered' — which is what the workgroup has stored for the read (the words are the tile's own: 32 reads of it each).
__device__ __forceinline__ void slow_pair(const ingest_args &a, uint64_t r, uint32_t sweep, uint32_t lane)
{
    const int X = (int)(sweep * kSweep), Xend = (int)min(a.n_cols, sweep * kSweep + kSweep);
    const uint2 *runs = a.runs + a.cig_off[r] + 3u * r;
    const uint32_t ne = a.nruns[r] + 3u;
    const uint64_t so = a.seq_off[r];
    const uint64_t qo = a.qual ? a.qual_off[r] : 0u;
    for (int c = X + (int)lane; c < Xend; c += 64) {
        uint32_t lo = 0, hi = ne - 1u;   // the last entry that begins at or before c (entry 0 does, the last one never)
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if ((int)(runs[mid].x & kRunMask) <= c) lo = mid;
            else hi = mid;
        }
        const uint2 e = runs[lo];
        const uint32_t kind = e.x >> 30;
        uint32_t sym = 6u;
        if (kind == 2u) sym = 4u;
        else if (kind == 1u) {
            const uint64_t q = (uint64_t)e.y + (uint64_t)(c - (int)(e.x & kRunMask));
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

// A workgroup = 128 reads x one sweep, four waves with two jobs.  Nothing a thread asks HBM for depends on another thread: every
// thread reads the descriptors of the reads whose pieces it takes itself (neighbouring lanes share them), so the two trips —
// descriptor, then pieces / entries — are the only waits before the first barrier, and there are two barriers in all.
//   waves 2, 3  pieces: thread t takes pieces t, t + 128, ... of the tile's 128 x kRowPieces (7 of 9 rounds), 16 bytes = 32 bases
//               each -> codes -> the staging area in query order.
//   waves 0, 1  a thread per read: its descriptor -> the row of the table (one entry covers the sweep: at once; several: an entry
//               at a time when they have arrived), the boundary blocks listed; then the other 2 rounds of pieces.
//   barrier; the listed blocks put together, a thread each; barrier; gather at the transpose (waves 0, 1).
// (Device stamps of the form before this one — the read waves made the table while the others waited, then everybody staged:
// descriptor 0.75 us, prologue 0.9, table 4.7 of which 1.5 a third trip for the reads with more than four entries, staging 1.8,
// general 1.1, transposing 2.6; waves 2, 3 waited 5.2 us of 14.6 at the first barrier.)
// DEFER (the first size, several sibling tiles a workgroup: ingest_planes_kernel): the unit's plane words are handed back in `keep`
// instead of being stored — returns whether the unit made any (false: outside the window's groups, or handed on) — and a read
// the workgroup has no room for hands the unit on instead of going through slow_pair (whose bit flips need the stores done).
template <bool QV, uint32_t EPR, bool BIG, bool DEFER = false>
__device__ __forceinline__ bool planes_unit(const ingest_args &a, const uint32_t b, uint32_t (&keep)[8][3])
{
    using shape = planes_shape<EPR>;
    constexpr uint32_t kEntCap = shape::kEntCap, kEntCapWave = shape::kEntCapWave, kRowBase = shape::kRowBase, kStageDw = shape::kStageDw;
#ifndef JL_INGEST_LDS_PAD
#define JL_INGEST_LDS_PAD 0               // (tuning: bytes of LDS a workgroup holds on top of what it uses — fewer workgroups per CU)
#endif
    __shared__ __attribute__((aligned(16))) uint32_t s_stage[kStageDw + JL_INGEST_LDS_PAD / 4];
    __shared__ __attribute__((aligned(16))) uint16_t s_tab[kTabSize];
    __shared__ uint32_t s_ent[kEntCap];           // the entries of the reads with several in the sweep, a half per read wave
    __shared__ uint8_t s_own[kEntCap];            // whose: the read | 0x80 for its last one
    static_assert(kTileReads <= 128u, "a read of the tile and a flag in a byte");
    __shared__ uint32_t s_nent[2u * kReadWaves];  // entries in each part; [kReadWaves + w]: wave w hands the unit on
    __shared__ uint32_t s_nslow;                  // reads left to slow_pair
    __shared__ uint8_t s_slow[kTileReads];
    uint32_t tid_ = threadIdx.x;
    if (DEFER) asm volatile("" : "+v"(tid_));      // (the units of a workgroup run in a loop: what follows from the thread's number is not to be
                                                   // kept in registers across it — hoisted, it was 110 of them)
    const uint32_t tid = tid_, wid = tid >> 6, lane = tid & 63u;
    // block -> (read tile, sweep).  Blocks b, b + 8, b + 16, ... are dealt to the same XCD one after the other; an XCD takes
    // whole groups of 1024 reads (group = xcd, xcd + 8, ...), and of a group all sweeps in turn, the tiles of the group
    // innermost.  So (a) the tiles that share the 128-byte lines of a sweep's planes meet in one L2, and (b) what the
    // sweeps of a group read again — the reads' entries, the 128-byte lines of packed bases that straddle two sweeps — is
    // in that L2 when the next sweep asks for it.
    const uint32_t xcd = b & 7u, jb = b >> 3, sub = jb % kSubTiles, qq = jb / kSubTiles;
    const uint32_t group = xcd + 8u * (qq / a.n_sweeps), sweep = qq % a.n_sweeps;
    if (group >= a.n_groups) return false;
    const uint32_t tile = kSubTiles * group + sub;
    const uint32_t X = sweep * kSweep, Xend = min(a.n_cols, X + kSweep), width = Xend - X;
    const uint4 *desc = a.desc + (uint64_t)sweep * a.n_reads;
    const uint64_t r0 = (uint64_t)tile * kTileReads;
    JL_ING_STAMP(a, 0)

    // the pieces p0, p0 + 128, ... (K of them): the descriptors of their reads, then the pieces, all of a thread's requests in
    // flight together; a piece that is not there (the read has fewer) asks for the read's first one again — the same number of
    // requests in every lane, so that a wait for something asked earlier does not wait for these
    struct piece_t { u32x4 v; uint32_t dst; int32_t Q; uint64_t qb; };    // dst: dword in the staging area, 0 = none
    // (every load is asked for whatever the read's number — a read past the last asks for the first one's, and is set to nothing
    // afterwards: loads inside a branch made the compiler wait for each before it asked for the next)
    auto ask_descs = [&](uint32_t p0, uint32_t K, uint4 (&d)[kPieceRoundsB], uint64_t (&qo)[kPieceRoundsB]) {
        bool in[kPieceRoundsB];
#pragma unroll
        for (uint32_t k = 0; k < kPieceRoundsB; ++k) {
            if (k >= K) break;
            const uint32_t j = (p0 + kTileReads * k) / kRowPieces;
            in[k] = r0 + j < a.n_reads;
            const uint64_t at = in[k] ? r0 + j : 0u;
            d[k] = desc[at];
            qo[k] = QV ? a.qual_off[at] : 0u;
        }
#pragma unroll
        for (uint32_t k = 0; k < kPieceRoundsB; ++k) {
            if (k >= K) break;
            if (!in[k]) {
                d[k] = make_uint4(0, 0, 0, 0);
                qo[k] = 0;
            }
        }
    };
    auto ask_pieces = [&](uint32_t p0, uint32_t K, const uint4 (&d)[kPieceRoundsB], const uint64_t (&qo)[kPieceRoundsB], piece_t (&pc)[kPieceRoundsB]) {
#pragma unroll
        for (uint32_t k = 0; k < kPieceRoundsB; ++k) {
            if (k >= K) break;
            const uint32_t p = p0 + kTileReads * k, j = p / kRowPieces, i = p - kRowPieces * j;
            uint32_t np = (d[k].w >> 16) & 0xFFu;
            if (np > kRowPieces || (d[k].w >> 24) == kDescMax) np = 0;      // (slow_pair's)
            uint64_t at = ((((uint64_t)d[k].w & 0xFFu) << 32) | d[k].x) + (i < np ? 4u * i : 0u);     // (in dwords)
            JL_ING_CHECK(a, 4u * at + 16u <= a.seq_bytes + 64u, 1, at, at = 0)
            // (plain loads: neighbouring lanes' pieces share lines, and so do the sweeps of a read)
#ifdef JL_TUNING
            if (JL_ING_SKIP(a, 9)) pc[k].v = __builtin_nontemporal_load(reinterpret_cast<const u32x4a4 *>(a.seq4 + 4u * at));   // (probe)
            else
#endif
            pc[k].v = *reinterpret_cast<const u32x4a4 *>(a.seq4 + 4u * at);
            pc[k].dst = i < np ? shape::row_dw(j) + 4u * i : 0u;
            pc[k].Q = (int32_t)d[k].z + (i < np ? 32 * (int32_t)i : 0);      // (a piece that is not there: its read's first one's qualities)
            pc[k].qb = qo[k];
        }
    };
    // (QV: a piece's qualities are asked for kQualAhead pieces ahead of its turn, outside any branch — the loads of a piece that is
    // not there ask for its read's first piece's again.  Asked for inside the piece's own turn, as round 5 had it, every piece
    // of a thread waited for its own trip to HBM, seven in a row: 184 us against 111 without qualities.)
    constexpr uint32_t kQualAhead = JL_INGEST_QUAL_AHEAD;
    auto stage_pieces = [&](uint32_t K, const piece_t (&pc)[kPieceRoundsB]) {
        piece_quals pq[kPieceRoundsB];
        if (QV) {
#pragma unroll
            for (uint32_t k = 0; k < kPieceRoundsB; ++k)
                if (k < K && k < kQualAhead) pq[k] = ask_quals(a, pc[k].Q, pc[k].qb);
        }
#pragma unroll
        for (uint32_t k = 0; k < kPieceRoundsB; ++k) {
            if (k >= K) break;
            if (QV && k + kQualAhead < K && k + kQualAhead < kPieceRoundsB) pq[k + kQualAhead] = ask_quals(a, pc[k + kQualAhead].Q, pc[k + kQualAhead].qb);
            if (pc[k].dst) {
                uint32_t S[4];
                if (JL_ING_SKIP(a, 6)) { S[0] = pc[k].v.x; S[1] = pc[k].v.y; S[2] = pc[k].v.z; S[3] = pc[k].v.w; }
                else piece_bases<QV>(a, pc[k].v, pq[k], S);
                u32x4 o = {S[0], S[1], S[2], S[3]};
                *reinterpret_cast<u32x4 *>(&s_stage[pc[k].dst]) = o;
            }
        }
    };

    bool slow_read = false;
    // WITHOUT QUALITIES a wave that still has its requests to HBM to make and its rows to stage goes before the waves that transpose
    // (s_setprio: the CU's arbiter picks by it among the waves ready to issue).  Without it the transposing waves of the CU's other
    // workgroups — three quarters of all instructions — hold up the few instructions that put the next tile's loads in flight: the
    // launch 110-113 -> 100-107 us, a whole build 171 -> 164 us on one stream and 159 -> 152 with two builds overlapping.  With
    // qualities the launch alone gains 3 us of 169 and the build LOSES 3-7 (two streams: 191 -> 198; in bench.py's once_through_qv
    // 185 -> 202): its raised waves also go before the waves of whatever else is on the device — the next build's cigar walk, the
    // pileup and the phasing of the windows before — so there it stays off.  (JL_INGEST_PRIO=0 for A/B.  Raised only until the loads
    // are out: the same; raised for the transposing instead: 116 / 176.)
#ifndef JL_INGEST_PRIO
#define JL_INGEST_PRIO 1
#endif
    constexpr bool kPrio = JL_INGEST_PRIO && !QV;
    if (kPrio) __builtin_amdgcn_s_setprio(3);
    if (tid >= kTileReads) {
        // ---- waves 2, 3: pieces
        uint4 d[kPieceRoundsB];
        uint64_t qo[kPieceRoundsB];
        piece_t pc[kPieceRoundsB];
        const uint32_t p0 = tid - kTileReads;
        ask_descs(p0, kPieceRoundsB, d, qo);
        ask_pieces(p0, kPieceRoundsB, d, qo, pc);
        JL_ING_STAMP(a, 3)
        if (!JL_ING_SKIP(a, 0)) stage_pieces(kPieceRoundsB, pc);
        JL_ING_STAMP(a, 6)
    } else {
        // ---- waves 0, 1: a thread per read
        typedef uint32_t u32x4a8 __attribute__((ext_vector_type(4), aligned(8)));
        uint4 dp[kPieceRoundsB];
        uint64_t qo[kPieceRoundsB];
        piece_t pc[kPieceRoundsB];
        const uint32_t p0 = kTileReads * kPieceRoundsB + tid;
        const uint64_t r = r0 + tid;
        uint4 d = make_uint4(0, 0, 0, 3u << 8 | 1u << 24);   // (no read: one entry of nothing)
        if (r < a.n_reads) d = desc[r];
        ask_descs(p0, kPieceRoundsA, dp, qo);
#ifdef JL_TUNING
        if (a.stamps) { asm volatile("" ::"v"(d.w)); JL_ING_STAMP(a, 1) }   // (the descriptor has arrived)
#endif
        if (tid < 4u) s_stage[tid] = tid < 2u ? 0x66666666u : 0x44444444u;
        if (tid == 4u) s_nslow = 0;
        if (lane == 0) s_nent[wid] = 0, s_nent[kReadWaves + wid] = 0;
        const uint32_t np = (d.w >> 16) & 0xFFu;
        uint32_t n_ent = d.w >> 24;
        bool slow = np > kRowPieces || n_ent == kDescMax;
        const bool simple = n_ent == 1u;
        if (simple || slow) n_ent = 0;
        // exclusive scan of the entries over the wave: each read wave has its own half of the entry area
        const uint32_t inc = wave_scan(n_ent);
        uint32_t off_e = inc - n_ent;
        if (n_ent && off_e + n_ent > kEntCapWave) {
            // more entries than the wave's part holds: the first size hands the whole unit on to the second (nothing of it is
            // written here, and no read of it goes to the slow list: the second size lists its own); the second leaves the
            // reads past the part's end to slow_pair
            if (!BIG) s_nent[kReadWaves + wid] = 1u;
            else slow = true;
            n_ent = 0;
        }
        if (DEFER && slow) {
            s_nent[kReadWaves + wid] = 1u;
            slow = false;
        }
        JL_ING_CHECK(a, !slow || r < a.n_reads, 3, r, slow = false)
        slow_read = slow;
        off_e += wid * kEntCapWave;
        // the sweep's entries of this read: the first eight in four 16-byte requests that go out together (entries past the
        // read's own belong to the next read or to the array's slack) — two deletions in a sweep are six entries, and a
        // wave in which ONE read needs a ninth makes another trip for it
        uint64_t src_at = n_ent ? ((((uint64_t)(d.w >> 8) & 0xFFu) << 32) | d.y) : 0u;
        JL_ING_CHECK(a, src_at + (n_ent > 4u ? max(n_ent, 8u) : 4u) <= a.n_entries, 2, src_at, src_at = 0)
        const uint2 *src = a.runs + src_at;
        u32x4a8 e01 = {0, 0, 0, 0}, e23 = e01, e45 = e01, e67 = e01;
        if (n_ent) {
            e01 = *reinterpret_cast<const u32x4a8 *>(src);
            e23 = *reinterpret_cast<const u32x4a8 *>(src + 2);
        }
        if (n_ent > 4u) {
            e45 = *reinterpret_cast<const u32x4a8 *>(src + 4);
            e67 = *reinterpret_cast<const u32x4a8 *>(src + 6);
        }
        ask_pieces(p0, kPieceRoundsA, dp, qo, pc);
        const int32_t q0 = (int32_t)d.z;
        const uint32_t row8 = 8u * shape::row_dw(tid);
        if (!n_ent) {
            // one entry covers the sweep: its blocks' addresses rise by eight codes a block (aligned bases) or stay (the
            // dword of '-', of 'not covered'), four blocks a store
            const uint32_t kind = (simple && !slow) ? (d.w >> 8) & 3u : 3u;
            const uint32_t a0 = kind == 1u ? row8 + (d.y - (uint32_t)q0) : kind == 2u ? 16u : 0u, st = kind == 1u ? 8u : 0u;
            uint32_t lo = a0 | ((a0 + st) << 16), hi = lo + 2u * (st | st << 16);
            const uint32_t step = 4u * (st | st << 16);
#pragma unroll
            for (uint32_t k = 0; k < kBlocks / 4u; ++k) {
                *reinterpret_cast<uint2 *>(s_tab + tab_at(tid, 4u * k)) = make_uint2(lo, hi);
                lo += step;
                hi += step;
            }
        }
        JL_ING_STAMP(a, 2)
        // several entries: -> LDS, four bytes each, with their owner; everybody makes table rows of them behind the barrier
        if (n_ent) {
            auto pack = [&](uint32_t x, uint32_t y) -> uint32_t {
                const uint32_t W = x & kRunMask, kind = x >> 30;
                const uint32_t wr = W <= X ? 0u : min(W - X, 256u);
                const uint32_t yq = y + (W < X ? X - W : 0u);      // query offset at the sweep's first column of the entry
                const uint32_t yr = (kind == 1u && wr < 256u) ? (yq - (uint32_t)q0) & 0x1FFFu : 0u;
                return wr | (kind << 9) | (yr << 11);
            };
            const uint32_t e8[8] = {pack(e01.x, e01.y), pack(e01.z, e01.w), pack(e23.x, e23.y), pack(e23.z, e23.w),
                                    pack(e45.x, e45.y), pack(e45.z, e45.w), pack(e67.x, e67.y), pack(e67.z, e67.w)};
            uint32_t *ent = s_ent + off_e;
            uint8_t *own = s_own + off_e;
#pragma unroll
            for (uint32_t i = 0; i < 8u; ++i)
                if (i < n_ent) {
                    ent[i] = e8[i];
                    own[i] = (uint8_t)(tid | (i + 1u == n_ent ? 0x80u : 0u));
                }
            for (uint32_t i = 8; i < n_ent; ++i) {
                const uint2 g = src[i];
                ent[i] = pack(g.x, g.y);
                own[i] = (uint8_t)(tid | (i + 1u == n_ent ? 0x80u : 0u));
            }
            atomicMax(&s_nent[wid], off_e - wid * kEntCapWave + n_ent);
        }
        JL_ING_STAMP(a, 4)
        if (!JL_ING_SKIP(a, 0)) stage_pieces(kPieceRoundsA, pc);
        JL_ING_STAMP(a, 6)
    }
    __syncthreads();
    JL_ING_STAMP(a, 7)
    if (!BIG) {
        uint32_t handed = 0;
#pragma unroll
        for (uint32_t w = 0; w < kReadWaves; ++w) handed |= s_nent[kReadWaves + w];
        if (handed) {
            if (tid == 0) a.big_list[atomicAdd(a.slow_count + 1, 1u)] = b;
            return false;
        }
    }
    if (slow_read) s_slow[atomicAdd(&s_nslow, 1u)] = (uint8_t)tid;      // (its row of the table says 'not covered')

    // ---- the table rows of the reads with several entries, everybody: a thread an entry.  Entry i
    // of a read covers the columns [its column, the next entry's column) and the read's last one only ends the one before
    // it: the blocks that lie WHOLLY inside are the entry's — addresses that rise by eight codes a block, or the dword of '-' /
    // of 'not covered' — and every block is wholly inside one entry or has an entry that begins inside it; those are put
    // together, by the thread of the first such entry, in that entry's side dword.  No list, no search, no loop per block.
    if (!JL_ING_SKIP(a, 1)) {
        uint32_t n_e = 0;
#pragma unroll
        for (uint32_t w = 0; w < kReadWaves; ++w) n_e += s_nent[w];
        for (uint32_t si = tid; si < n_e; si += kThreads) {
            uint32_t slot = 0, rem = si;      // the si-th entry of the parts one behind the other
            bool found = false;
#pragma unroll
            for (uint32_t w = 0; w < kReadWaves; ++w) {
                const uint32_t nw = s_nent[w];
                if (!found && rem < nw) {
                    slot = w * kEntCapWave + rem;
                    found = true;
                }
                if (!found) rem -= nw;
            }
            const uint32_t own = s_own[slot], e = s_ent[slot], nx = s_ent[slot + 1u];
            const uint32_t wr = ent_col(e);
            if ((own & 0x80u) || wr >= width) continue;
            const uint32_t row8 = 8u * shape::row_dw(own);
            // (a) the entry's whole blocks [bf, be): singly up to a multiple of four, four a store, singly again
            const uint32_t wn = ent_col(nx);
            const uint32_t be = wn >= width ? kBlocks : wn >> 3;
            uint32_t bf = (wr + 7u) >> 3;
            if (bf < be) {
                const uint32_t st = ent_kind(e) == 1u ? 8u : 0u;
                uint32_t av = ent_addr(e, row8, 8u * bf);
#pragma unroll
                for (uint32_t k = 0; k < 3u; ++k)
                    if ((bf & 3u) && bf < be) {
                        s_tab[tab_at(own, bf)] = (uint16_t)av;
                        av += st;
                        ++bf;
                    }
                uint32_t p01 = av | ((av + st) << 16);
                const uint32_t p_st = st | st << 16;
                for (; bf + 4u <= be; bf += 4u) {
                    *reinterpret_cast<uint2 *>(s_tab + tab_at(own, bf)) = make_uint2(p01, p01 + 2u * p_st);
                    p01 += 4u * p_st;
                }
                av = p01 & 0xFFFFu;
#pragma unroll
                for (uint32_t k = 0; k < 3u; ++k)
                    if (bf < be) {
                        s_tab[tab_at(own, bf)] = (uint16_t)av;
                        av += st;
                        ++bf;
                    }
            }
            // (b) the block it begins inside, if it is the first entry to do so
            const uint32_t bb = wr >> 3;
            if (!(wr & 7u) || JL_ING_SKIP(a, 3)) continue;
            uint32_t ee = s_ent[slot - 1u];        // (an entry that begins inside a block is not its read's first)
            if (ent_col(ee) > 8u * bb) continue;
            const uint32_t c0 = 8u * bb, c1 = c0 + 8u;
            uint32_t bases = 0, m_al = 0, m_del = 0;      // the block's aligned bases (BAM's codes), where they are, where a deletion is
            for (uint32_t kk = slot;; ++kk) {
                const uint32_t nn = s_ent[kk];
                const uint32_t W = ent_col(ee), Wn = ent_col(nn);
                const uint32_t ca = max(W, c0), cb = min(Wn, c1);
                const uint32_t kind = ent_kind(ee);
                if (ca < cb && kind != 3u) {
                    const uint32_t m = (cb - ca == 8u ? 0xFFFFFFFFu : ((1u << (4u * (cb - ca))) - 1u)) << (4u * (ca - c0));
                    if (kind == 1u) {
                        const uint32_t A = ent_addr(ee, row8, ca);
                        bases |= (__builtin_amdgcn_alignbit(s_stage[(A >> 3) + 1u], s_stage[A >> 3], 4u * (A & 7u)) << (4u * (ca - c0))) & m;
                        m_al |= m;
                    } else m_del |= m;
                }
                if (Wn >= c1 || (s_own[kk] & 0x80u)) break;     // (the read's last entry is nothing: 'not covered' stays)
                ee = nn;
            }
            const uint32_t R = (codes_of_bases8(bases) & m_al) | (0x44444444u & m_del) | (0x66666666u & ~(m_al | m_del));
            s_stage[4u + slot] = R;
            s_tab[tab_at(own, bb)] = (uint16_t)(8u * (4u + slot));
        }
    }
    JL_ING_STAMP(a, 8)
    __syncthreads();
    JL_ING_STAMP(a, 9)
    const uint32_t n_slow = s_nslow;

    // ---- gather at the transpose: thread = 32 reads x 8 columns; neighbouring lanes write consecutive dwords of a plane
    if (kPrio) __builtin_amdgcn_s_setprio(0);
    {
        const uint32_t G = tid % kTileGroups, blk = tid / kTileGroups;
        const bool act = blk < kBlocks && 8u * blk < width && !JL_ING_SKIP(a, 4);
        uint32_t out[8][3];
        if (act) {
            // (the dwords in front of the reads' rows — 'not covered', '-', the boundary blocks — hold symbol codes already; `codes`
            // collects, a bit a read, whose dword is one of those: v_alignbit shifts the sign of address - first row in)
            uint32_t R[32], codes = 0;
            // (the block's place in the rows of the reads i, i + 8, ...: eight offsets)
            uint32_t at[8];
#pragma unroll
            for (uint32_t i = 0; i < 8u; ++i) at[i] = tab_at(32u * G + i, blk);
#pragma unroll
            for (int i = 31; i >= 0; --i) {
                const uint32_t A = s_tab[at[i & 7] + (uint32_t)(i & ~7) * kTabRow];
                R[i] = __builtin_amdgcn_alignbit(s_stage[(A >> 3) + 1u], s_stage[A >> 3], 4u * (A & 7u));
                codes = __builtin_amdgcn_alignbit(codes, A - 8u * kRowBase, 31u);
            }
            nibble_rows_to_plane_words(R, codes, out);
        }
        if (DEFER) {
#pragma unroll
            for (uint32_t jj = 0; jj < 8u; ++jj)
#pragma unroll
                for (uint32_t k = 0; k < 3u; ++k) keep[jj][k] = out[jj][k];
        } else if (act) {
            const uint64_t byte = (uint64_t)tile * (kTileReads / 8u) + (uint64_t)G * 4u;
#ifdef JL_TUNING
            if (JL_ING_SKIP(a, 7) || JL_ING_SKIP(a, 8)) {   // (probes, wrong data by design: 7 the same bytes in 16-byte stores, a quarter of the requests; 8 non-temporal stores)
                uint8_t *row = a.msa + (uint64_t)((X + 8u * blk) * 3u) * a.plane_stride + (uint64_t)tile * (kTileReads / 8u) + (JL_ING_SKIP(a, 8) ? (uint64_t)G * 4u : 0u);
                for (uint32_t jj = 0; jj < 8u; ++jj)
                    for (uint32_t k = 0; k < 3u; ++k) {
                        if (JL_ING_SKIP(a, 8)) __builtin_nontemporal_store(out[jj][k], reinterpret_cast<uint32_t *>(row));
                        else if (G == 0) { u32x4 o = {out[jj][k], out[jj][0], out[jj][1], out[jj][2]}; *reinterpret_cast<u32x4 *>(row) = o; }
                        row += a.plane_stride;
                    }
            } else if (JL_ING_SKIP(a, 11)) {   // (probe, wrong data by design: the same bytes as 64-byte requests — a quarter of the rows, four tiles wide)
                uint8_t *row = a.msa + (uint64_t)((X + 8u * blk) * 3u) * a.plane_stride + (uint64_t)(tile & ~3u) * (kTileReads / 8u) + (uint64_t)G * 16u;
                for (uint32_t jj = 0; jj < 8u; ++jj)
                    for (uint32_t k = 0; k < 3u; ++k) {
                        if (((jj * 3u + k) & 3u) == (sub & 3u)) { u32x4 o = {out[jj][k], out[jj][0], out[jj][1], out[jj][2]}; *reinterpret_cast<u32x4 *>(row) = o; }
                        row += a.plane_stride;
                    }
            } else
#endif
            if (byte < a.plane_stride && (!JL_ING_SKIP(a, 2) || out[0][0] == 0x12345u)) {
                // (one 64-bit multiply for the first plane row, then a stride at a time; a block of eight whole columns — all but
                // the window's last — stores without a question per column: a predicate per store was a sixth of this phase)
                uint8_t *row = a.msa + (uint64_t)((X + 8u * blk) * 3u) * a.plane_stride + byte;
                if (8u * blk + 8u <= width) {
#pragma unroll
                    for (uint32_t jj = 0; jj < 8u; ++jj)
#pragma unroll
                        for (uint32_t k = 0; k < 3u; ++k) {
                            *reinterpret_cast<uint32_t *>(row) = out[jj][k];
                            row += a.plane_stride;
                        }
                } else {
                    for (uint32_t jj = 0; 8u * blk + jj < width; ++jj)
                        for (uint32_t k = 0; k < 3u; ++k) {
                            *reinterpret_cast<uint32_t *>(row) = jj == 0u ? out[0][k] : jj == 1u ? out[1][k] : jj == 2u ? out[2][k] : jj == 3u ? out[3][k] : jj == 4u ? out[4][k] : jj == 5u ? out[5][k] : out[6][k];
                            row += a.plane_stride;
                        }
                }
            }
        }
    }
    JL_ING_STAMP(a, 10)
    // ---- the reads left out (none, in a CCS sample): column by column behind the workgroup's own stores, a wave a read
    if (!DEFER && n_slow) {       // (the same in every thread)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (uint32_t i = wid; i < n_slow; i += kThreads / 64u) slow_pair(a, r0 + s_slow[i], sweep, lane);
    }
    return true;
}

// The first size: a workgroup a unit (blockIdx -> unit: planes_unit).  The second: the units the first handed on, in the
// order they came, taken in turns by a grid that fills the chip once (an empty list costs a launch of 768 workgroups).
constexpr uint32_t kBigGrid = 768u;
// NT (the first size): sibling tiles a workgroup takes one after the other, their plane words kept in registers and stored
// TOGETHER at the end — NT x 16 bytes of a line a request instead of 16.  The L2 takes a write request per (row, tile): 7 x 10^6 a
// window, more than all its read requests; the same bytes as 64-byte requests (a probe: JL_ING_SKIP bit 11) took 19 us off the
// launch's 113 and 29 off the 180 with qualities.  The tiles' words change places between the lanes of a DPP row (16 lanes = four
// blocks of 8 columns x four 32-read groups): the lane at position u of a run of NT blocks ends up with tile u's words of all NT
// blocks (log2 NT rounds of two DPP moves a word), and 4 NT neighbouring lanes write NT x 16 contiguous bytes of a row.
template <bool QV, uint32_t EPR, bool BIG, uint32_t NT = 1u>
__global__ __launch_bounds__(kThreads, (BIG ? 1 : JL_INGEST_MIN_WGS)) void ingest_planes_kernel(ingest_args a)
{
    static_assert(NT == 1u || NT == 2u || NT == 4u, "a DPP row holds four blocks");
    if (BIG) {
        uint32_t none[8][3];
        const uint32_t n = a.slow_count[1];
        for (uint32_t u = blockIdx.x; u < n; u += gridDim.x) {
            planes_unit<QV, EPR, true>(a, a.big_list[u], none);
            __syncthreads();      // (the next unit's LDS)
        }
    } else if (NT == 1u) {
        uint32_t none[8][3];
        planes_unit<QV, EPR, false>(a, blockIdx.x, none);
    } else {
        // workgroup w -> the units (planes_unit's block numbers) of NT consecutive tiles of one (group, sweep), on w's XCD
        const uint32_t w = blockIdx.x, xcd = w & 7u, jw = w >> 3, h = jw % (kSubTiles / NT), qq = jw / (kSubTiles / NT);
        const uint32_t b0 = xcd + 8u * (qq * kSubTiles + NT * h);
        uint32_t D[NT][8][3];
        uint32_t ok = 0;
#pragma unroll 1
        for (uint32_t i = 0; i < NT; ++i) {
            // (the words kept so far move down a place: after the last tile D[t] is tile t's)
#pragma unroll
            for (uint32_t t = 0; t + 1u < NT; ++t)
#pragma unroll
                for (uint32_t jj = 0; jj < 8u; ++jj)
#pragma unroll
                    for (uint32_t k = 0; k < 3u; ++k) D[t][jj][k] = D[t + 1u][jj][k];
            const bool got = planes_unit<QV, EPR, false, true>(a, b0 + 8u * i, D[NT - 1u]);
            ok = (ok >> 1) | (got ? 1u << (NT - 1u) : 0u);
            __syncthreads();      // (the next tile's LDS)
        }
        if (!ok) return;
        const uint32_t tid = threadIdx.x;
        if (tid >= kTileGroups * kBlocks) return;           // (the transposing threads)
        const uint32_t G = tid % kTileGroups, blk = tid / kTileGroups, u = blk & (NT - 1u);
        // ---- the tiles' words change places: round s swaps, between blocks 2^s apart, the words of the tiles whose bit s differs
#pragma unroll
        for (uint32_t sft = 0; (1u << sft) < NT; ++sft) {
#pragma unroll
            for (uint32_t t0 = 0; t0 < NT; ++t0) {
                if (t0 & (1u << sft)) continue;
                const uint32_t t1 = t0 | (1u << sft);
#pragma unroll
                for (uint32_t jj = 0; jj < 8u; ++jj)
#pragma unroll
                    for (uint32_t k = 0; k < 3u; ++k) {
                        const int x0 = (int)D[t0][jj][k], x1 = (int)D[t1][jj][k];
                        // lanes whose block has bit s clear take the partner's (4 << s lanes up) tile-t0 words into slot t1,
                        // the others the partner's tile-t1 words into slot t0 (bank = block within the row)
                        if (sft == 0u) {
                            D[t1][jj][k] = (uint32_t)__builtin_amdgcn_update_dpp(x1, x0, 0x104, 0xF, 0x5, false);   // row_shl:4, banks 0 2
                            D[t0][jj][k] = (uint32_t)__builtin_amdgcn_update_dpp(x0, x1, 0x114, 0xF, 0xA, false);   // row_shr:4, banks 1 3
                        } else {
                            D[t1][jj][k] = (uint32_t)__builtin_amdgcn_update_dpp(x1, x0, 0x108, 0xF, 0x3, false);   // row_shl:8, banks 0 1
                            D[t0][jj][k] = (uint32_t)__builtin_amdgcn_update_dpp(x0, x1, 0x118, 0xF, 0xC, false);   // row_shr:8, banks 2 3
                        }
                    }
            }
        }
        // ---- D[p] = this lane's tile's (tile u of the NT) words of block (blk & ~(NT - 1)) + p: NT x 4 lanes a row
        if (!((ok >> u) & 1u)) return;                      // (this lane's tile was handed on)
        const uint32_t sub0 = NT * h, group = xcd + 8u * (qq / a.n_sweeps), sweep = qq % a.n_sweeps;
        const uint32_t X = sweep * kSweep, width = min(a.n_cols, X + kSweep) - X;
        const uint64_t byte = (uint64_t)(kSubTiles * group + sub0 + u) * (kTileReads / 8u) + (uint64_t)G * 4u;
#pragma unroll
        for (uint32_t p = 0; p < NT; ++p) {
            const uint32_t bp = (blk & ~(NT - 1u)) + p;
            if (8u * bp >= width) continue;
            uint8_t *row = a.msa + (uint64_t)((X + 8u * bp) * 3u) * a.plane_stride + byte;
            if (8u * bp + 8u <= width) {
#pragma unroll
                for (uint32_t jj = 0; jj < 8u; ++jj)
#pragma unroll
                    for (uint32_t k = 0; k < 3u; ++k) {
                        *reinterpret_cast<uint32_t *>(row) = D[p][jj][k];
                        row += a.plane_stride;
                    }
            } else {
#pragma unroll
                for (uint32_t jj = 0; jj < 8u; ++jj)
#pragma unroll
                    for (uint32_t k = 0; k < 3u; ++k) {
                        if (8u * bp + jj < width) *reinterpret_cast<uint32_t *>(row) = D[p][jj][k];
                        row += a.plane_stride;
                    }
            }
        }
    }
}

// the counters of a build: [0] pairs listed, [1] units handed on = 0; [2..3] the verdict word = all ones unless an earlier build's is still unread
__global__ void ingest_init_kernel(uint32_t *count, uint32_t fresh_verdict)
{
    const uint32_t t = threadIdx.x;
    if (t < 2u) count[t] = 0u;
    else if (t < 4u && fresh_verdict) count[t] = 0xFFFFFFFFu;
}

}  // namespace

uint32_t jl_ingest_sweeps(uint32_t n_cols) { return (n_cols + kSweep - 1u) / kSweep; }

// the planes kernel's workgroups: (groups of 1024 reads per XCD, rounded up) x sweeps x 8 XCDs x tiles of a group
static uint32_t planes_units(const jl_ctx *ctx)
{
    const uint32_t groups = (uint32_t)(ctx->plane_stride * 8u / 1024u);    // (a multiple of 1024 reads: whole line groups of tiles)
    return (groups + 7u) / 8u * jl_ingest_sweeps(ctx->n_cols) * 8u * kSubTiles;
}

// pairs (8 bytes) of d_slow: the tuning build's stamps, and behind them the units handed on (a word each)
constexpr uint32_t kStampPairs = 160u * 4u * 12u;
size_t jl_ingest_slow_room(const jl_ctx *ctx) { return (size_t)kStampPairs + planes_units(ctx) / 2u + 8u; }

// The host's copy of cigar_walk_kernel's rule — which reads it leaves to the second launch (more than kWalkOps ops, or more entries
// than a read has room for in its LDS) — for the upload (jl_records_append), which looks at the few reads of a CCS sample with more
// ops than a read has entries: when it finds none, jl_launch_ingest does not make that launch.
uint32_t jl_ingest_short_ops() { return kWalkEnt - 3u; }      // a read of so many ops at most cannot be a long one
bool jl_ingest_read_is_long(const uint32_t *cigar, uint64_t n_ops)
{
    if (n_ops > kWalkOps) return true;
    constexpr uint32_t kKinds = (2u << 4) | (3u << 6) | (1u << 14) | (1u << 16);   // two bits per op: D, N, =, X (cigar_walk_kernel)
    uint32_t n_runs = 0, prev_kind = 0;
    for (uint64_t t = 0; t < n_ops; ++t) {
        const uint32_t op = cigar[t] & 15u, len = cigar[t] >> 4;
        const uint32_t kind = len == 0u ? 0u : (kKinds >> (2u * op)) & 3u;
        if (kind != 0u && !(kind == 1u && prev_kind == 1u)) ++n_runs;
        prev_kind = kind;
    }
    return n_runs + 3u > kWalkEnt;
}

// d_runs: n_cig + 3 n_reads + 8 entries; d_nruns: n_reads; d_desc: n_reads x sweeps descriptors; d_slow: jl_ingest_slow_room() pairs.
// d_slow_count, 64 bytes: [0] pairs listed, [1] units handed on — zeroed by the build's first launch; [2..3] the 64-bit word of the
// first malformed record (all ones: none — so it is allocated, and so jl_ingest_verdict leaves it when it has read one; a build
// whose predecessor's word has not been read yet folds its own into it, atomicMin); [4..15] the tuning build's checks.
// Everything is enqueued on ctx->stream; nothing waits.
void jl_launch_ingest(jl_ctx *ctx, const int32_t *d_pos, const uint32_t *d_cigar, const uint64_t *d_cig_off,
                      const uint8_t *d_seq4, const uint64_t *d_seq_off, const uint8_t *d_qual,
                      const uint64_t *d_qual_off, uint32_t min_qv, uint2 *d_runs, uint32_t *d_nruns, uint4 *d_desc,
                      uint32_t *d_slow_count, uint2 *d_slow, bool maybe_long, uint64_t seq_bytes, uint64_t n_entries)
{
    hipStream_t st = ctx->stream;
    const uint32_t ns = jl_ingest_sweeps(ctx->n_cols);
    // The counters are zeroed by the first launch (cigar_walk_kernel); the verdict word is all ones from its allocation on and again
    // whenever a verdict has been read (jl_ingest_verdict): a build whose predecessor's verdict is still unread folds its own into it.
    if (!ctx->n_reads) hipLaunchKernelGGL(ingest_init_kernel, dim3(1), dim3(64), 0, st, d_slow_count, 0u);
    if (ctx->n_reads) {
        const uint32_t per_wg = 4u * kRunsReadsPerWave;
        unsigned long long *bad = reinterpret_cast<unsigned long long *>(d_slow_count + 2);
        const uint64_t *qo = d_qual ? d_qual_off : nullptr;
        (void)per_wg;
        hipLaunchKernelGGL(cigar_walk_kernel, dim3((uint32_t)((ctx->n_reads + kWalkReads - 1u) / kWalkReads)), dim3(256), 0, st,
                           ctx->n_reads, d_pos, d_cigar, d_cig_off, d_seq_off, qo, ctx->win_begin, ctx->n_cols, ns, d_runs, d_nruns, d_desc, bad, d_slow_count, maybe_long ? 1u : 0u);
        // (the launch for the long reads: not when the upload has looked and found none — every CCS sample: 6 us of a build)
        if (maybe_long)
            hipLaunchKernelGGL((cigar_runs_kernel<kRunsLdsLarge, true>), dim3((uint32_t)std::min<uint64_t>(kRunsLongGrid, (ctx->n_reads + 255u) / 256u)), dim3(256), 0, st,
                               ctx->n_reads, d_pos, d_cigar, d_cig_off, d_seq_off, qo, ctx->win_begin, ctx->n_cols, ns, d_runs, d_nruns, d_desc, bad);
    }
#ifdef JL_TUNING
    if (getenv("JL_ING_ONLY_RUNS")) return;     // (probe builds of cigar_runs leave descriptors nobody may follow)
#endif
    ingest_args a;
    a.n_reads = ctx->n_reads;
    a.n_cols = ctx->n_cols;
    a.n_sweeps = ns;
    const uint64_t reads_pad = ctx->plane_stride * 8u;                 // a multiple of 1024: whole line groups of tiles
    a.n_groups = (uint32_t)(reads_pad / 1024u);
    a.min_qv = std::min<uint32_t>(min_qv, 127u);   // (the byte-parallel compare of the QV path; BAM qualities end at 93)
    a.cig_off = d_cig_off;
    a.seq4 = d_seq4;
    a.seq_off = d_seq_off;
    const bool qv = d_qual != nullptr && min_qv != 0u;
    a.qual = qv ? d_qual : nullptr;
    a.qual_off = qv ? d_qual_off : nullptr;
    a.runs = d_runs;
    a.nruns = d_nruns;
    a.desc = d_desc;
    a.slow_count = d_slow_count;
    a.big_list = reinterpret_cast<uint32_t *>(d_slow + kStampPairs);
    a.msa = ctx->d_msa;
    a.plane_stride = ctx->plane_stride;
    a.seq_bytes = seq_bytes;
    a.n_entries = n_entries;
    a.dbg = d_slow_count + 4;    // (twelve spare words of the 64-byte block)
    a.skip = 0;
    a.stamps = nullptr;
#ifdef JL_TUNING
    if (getenv("JL_ING_STAMPS")) {
        a.stamps = reinterpret_cast<unsigned long long *>(d_slow);
        hipMemsetAsync(d_slow, 0, kStampPairs * 8u, st);
    }
    hipMemsetAsync(d_slow_count + 4, 0, 48, st);
    if (const char *e = getenv("JL_ING_SKIP")) a.skip = (uint32_t)atoi(e);
#endif
    const uint32_t grid = planes_units(ctx);
    constexpr uint32_t E = JL_INGEST_ENT_PER_READ, EB = JL_INGEST_ENT_PER_READ_BIG;
    // (sibling tiles a workgroup, tools_tuning/nt_probe.sh: with qualities two — 170 us against 179 with one, 213-219 with four (the
    // registers leave two workgroups a CU); without qualities one: 111-113 with one or two, 132 with four)
    if (qv) {
        hipLaunchKernelGGL((ingest_planes_kernel<true, E, false, JL_INGEST_NT_QV>), dim3(grid / JL_INGEST_NT_QV), dim3(kThreads), 0, st, a);
        hipLaunchKernelGGL((ingest_planes_kernel<true, EB, true>), dim3(kBigGrid), dim3(kThreads), 0, st, a);
    } else {
        hipLaunchKernelGGL((ingest_planes_kernel<false, E, false, JL_INGEST_NT>), dim3(grid / JL_INGEST_NT), dim3(kThreads), 0, st, a);
        hipLaunchKernelGGL((ingest_planes_kernel<false, EB, true>), dim3(kBigGrid), dim3(kThreads), 0, st, a);
    }
}
