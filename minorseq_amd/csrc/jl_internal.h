// jl_internal.h — private to libjuliet_hip.so: context layout, launch helpers, kernel entry points.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/juliet_hip.h"
#include "jl_synth.h"

// Tuning builds (make EXTRA=-DJL_TUNING; -DJL_PILEUP_TUNING implies it) compile in the probes of tools_tuning/:
// environment switches for launch shapes, device-clock stamps between the stages, wrong-by-design kernel variants.
// The shipped library has none of them and reads no environment variable on its launch paths.
#if defined(JL_PILEUP_TUNING) && !defined(JL_TUNING)
#define JL_TUNING 1
#endif

#define JL_VARIANT_CAP 4096u      // rows of the resident variant table (all-gather stride)
#define JL_CAND_CAP 4096u         // haplotype candidates (groups with >= min_reads) the selector can rank
#define JL_POS_PER_WORD 10u       // variant positions per 64-bit key word (6 bits each)
// Largest phase launch whose workgroups may wait for each other inside the launch (the per-read ids are then written by
// the same launch): all of them must be resident at once, next to as many more such launches as queues run at a time.
// 1536 places on the chip (75 VGPRs, 20.5 KB LDS per block); 128 leaves room for eight concurrent launches and more.
#ifndef JL_FOLD_MAX_BLOCKS
#define JL_FOLD_MAX_BLOCKS 128u
#endif
#define JL_TIMELINE_ROWS 4096u
#define JL_TIMELINE_SLOTS 8u
#define JL_INS_LEN_BINS 32u        // insertion lengths 0..30 by value, 31 = longer
#define JL_INS_MAX_BASES 30u       // inserted bases tracked per insertion
#define JL_GUESS_PAD 32u           // zero bytes after the last column's seed base
#ifndef JL_INGEST_SWEEP
#define JL_INGEST_SWEEP 256u        // columns a workgroup of the record ingest expands at a time (kernels_ingest.hip)
#endif


// resolved reference codon per position
#define JL_REF_MAJORITY 0xFFu
#define JL_REF_SKIP 0xFEu

struct jl_phase_meta {  // device-resident scalars of one phasing run
    uint32_t n_var;     // rows used
    uint32_t vp;        // distinct variant columns
    uint32_t kwords;    // 64-bit words per read key
    uint32_t n_occupied;
    uint32_t overflow;  // bit0: more than JL_CAND_CAP candidates; bit1: more than JL_MAX_HAPLOTYPES qualified;
                        // bit2: the key buffer was too small for vp_true positions (phasing skipped, host re-runs)
    uint32_t vp_true;   // distinct variant columns before the capacity check
    uint32_t id_bits;   // width of the per-read ids the run wrote: 4, 8 or 16 (see JL_ID_* below)
    uint32_t pad_;
    jl_phase_summary summary;
};

// Per-read haplotype ids travel in the narrowest code that holds the run's haplotype count H (they cross PCIe into
// pinned host memory: 2 bytes per read were the largest transfer of a step):
//   4 bits (H <= 14): 0..13 haplotype, 14 insufficient coverage, 15 damaged; read i in nibble i & 7 of dword i >> 3
//   8 bits (H <= 254): 0..253, 254 insufficient, 255 damaged
//   16 bits: the id itself (JL_HAP_INSUFFICIENT / JL_HAP_DAMAGED)
// jl_phase_fetch expands to 16 bits; jl_run_view hands out the packed form with its width.
#define JL_ID4_MAX_H 14u
#define JL_ID8_MAX_H 254u

// Small fixed-size result block: everything a typical run returns except the per-read ids, gathered by one
// tiny kernel so that ONE device-to-host copy into pinned memory ends the step (results that do not fit set
// fits_* = 0 and the fetch calls fall back to their piecewise copies).
#define JL_PACK_MAX_VAR 128u
#define JL_PACK_MAX_VP 128u
#define JL_PACK_MAX_HAP 128u
#define JL_PACK_PATTERN_BYTES 8192u   // e.g. 128 haplotypes x 64 positions
#define JL_PACK_HIT_BYTES 16384u      // e.g. 128 variants x 128 haplotypes
#define JL_SEL_HIT_BYTES 8192u        // what the selection out of LDS holds of it (larger results take the general path)
#define JL_PACK_COOC_N 64u
#define JL_PACK_MAGIC 0x4A4C504Bu
struct jl_pack {
    uint32_t magic, nvar_total, fits_call, fits_phase;
    uint32_t phase_ran, overflow, vp, H;
    jl_phase_summary summary;
    uint32_t nv_phase, cooc_fits, id_bits, pad_[5];
    jl_variant variants[JL_PACK_MAX_VAR];
    uint32_t pos_cols[JL_PACK_MAX_VP];
    uint32_t hap_count[JL_PACK_MAX_HAP];
    uint8_t hap_pattern[JL_PACK_PATTERN_BYTES];  // [H][vp]
    uint8_t hit[JL_PACK_HIT_BYTES];              // [nv][H]
    uint32_t cooc[JL_PACK_COOC_N * JL_PACK_COOC_N];  // [nv][nv]
};

// bytes of jl_pack up to and including variants[]: what one rank contributes to the compact all-gather
#define JL_PACK_HEAD_BYTES (offsetof(jl_pack, pos_cols))

// ---- per-window argument blocks of the three stage kernels.  A single run passes one by value; a group run
// (jl_group_run_async) keeps an array of them in device memory and launches each stage ONCE for all windows
// (blockIdx.z = window).  `n_blocks` = workgroups of the window's own grid (the arrival counters count to it).
// Pointers that a kernel loads from an argument block in memory are "generic" to the compiler, which then emits flat
// loads (they count on two wait counters at once, so no wait on them can be an exact count and a register prefetch
// overlaps nothing).  The streaming code therefore takes pointers typed as global memory (address space 1).
#if defined(__HIP_DEVICE_COMPILE__)
#define JL_AS1 __attribute__((address_space(1)))
#else
#define JL_AS1
#endif

struct jl_call_args {
    double alpha, n_tests, match, substitution, min_perc, max_perc;
    int32_t expected_round;
    uint32_t P;
    int32_t tail;   // 0 one-sided greater, 1 two-sided
    uint32_t pad_;
};

struct jl_win_pileup {
    const uint8_t *msa;          // the window's bit planes
    uint64_t plane_stride;       // bytes per plane
    uint32_t n_cols, n_tiles, n_chunks, pad_;
    const uint2 *chunks;
    const uint32_t *guess32;
    uint32_t *counts, *hist;
};

struct jl_win_call {   // call_kernel: the Fisher stage from histograms in HBM (stage API, windows too deep for one block per chunk)
    jl_call_args A;
    const uint32_t *pos_gene, *pos_codon, *pos_col;
    const uint8_t *pos_refcfg;
    const uint32_t *hist;
    const uint64_t *drm;
    uint64_t *called;
    jl_variant *staged;
    jl_phase_meta *meta;   // run counters to zero (null: none)
    uint32_t n_blocks, pad_;
};

// compact_kernel: ordered compaction of the staged rows into the variant table, then optionally the phasing plan
// (multi-word pipeline, stage API) and / or the result block of a run without phasing
struct jl_win_compact {
    uint32_t P, cap, n_cols, kwords_cap;
    const uint64_t *called;
    const jl_variant *staged;
    jl_variant *rows;
    uint32_t *n_rows;
    uint8_t *varcol;
    uint32_t *vpcols, *col2pos;
    jl_phase_meta *meta;
    uint32_t plan, fast_only;        // plan != 0: distinct variant columns (phase_plan.h)
    uint32_t pack, pad_;             // pack != 0: write the result block (run without phasing)
    jl_pack *pk, *mirror;
    uint32_t *seq_dev;
    volatile uint32_t *seq_host;     // non-null: this launch ends a run
    uint8_t *xhead;                  // bound exchange: see jl_select_args
};

// what the last block of the fused phase launch needs to run the selection (and to end the run)
struct jl_select_args {
    uint32_t run;  // 0: the generic (multi-word) pipeline follows with its own select launch
    uint32_t min_reads, n_cols, cooc_cap;
    uint32_t *slot_hap;
    const jl_variant *variants;
    const uint32_t *col2pos;
    uint32_t *hap_count;
    uint8_t *hap_pattern;
    uint8_t *hit;
    const uint32_t *n_rows;
    uint32_t *cooc;
    jl_pack *pk, *mirror;
    uint32_t *arrive;
    uint32_t *seq_dev;
    volatile uint32_t *seq_host;
    // fold != 0: the per-read ids are written by this launch too (all its workgroups are resident together): the
    // other workgroups wait for the selection on `flag`, then map their own reads
    uint32_t fold, pad_;
    uint32_t *flag, *arrive2;
    uint16_t *read_hap;
    // plan from the call masks (whole-path runs): every workgroup derives the variant columns itself, one extra
    // workgroup compacts the rows into the table meanwhile
    const uint64_t *called;          // null: the plan is in meta / vpcols already (stage API, multi-word pipeline)
    const jl_variant *staged;
    const uint32_t *pos_col;
    jl_variant *rows;
    uint32_t *n_rows_out, *vpcols_out, *col2pos_out;
    uint32_t P, cap, kwords_cap, pad2_;
    // export (phasing sharded by reads, SURVEY §8e option A): instead of ranking the groups of THIS matrix the
    // selection writes them out — count and pattern of every occupied slot, in the order of the occupied list — for the
    // host to merge with the other ranks' (jl_phase_groups_fetch); the per-read ids follow in jl_phase_regroup
    uint32_t *exp_count;             // null: normal selection
    uint8_t *exp_pattern;            // [exp_cap][exp_stride]
    uint32_t exp_cap, exp_stride;
    uint32_t *exp_head;              // optional: [JL_EXP_HEAD_WORDS] what a merge needs of the run's scalars (jl_exp_head)
    // bound exchange (jl_group_exchange_bind): where the head of the result block goes a third time — this rank's part of the
    // region the all-gather of the launch works in (pinned host memory, or its device stage).  Null: none.
    uint8_t *xhead;
};

// head of an exported group table: the block's first words when it travels (pinned host memory, or the all-gather)
struct jl_exp_head {
    uint32_t n_groups, vp, overflow;   // overflow != 0: more groups than the block holds (n_groups = the number needed)
    uint32_t damaged, gap, heteroduplex, partial, clean;   // the slice's read categories (clean = in some group)
};
#define JL_EXP_HEAD_WORDS 8u

// The fused phase launch reading its variant columns where they lie (a session whose positions are all in windows of this
// device: no compact matrix, no pack launch, no plan kernel in front).  Position p = nine plane rows (three columns x three
// planes) of `stride` bytes starting at col[p] (already offset to the byte of the slice's first read); vp travels by value.
struct jl_direct_cols {
    const uint8_t *col[JL_POS_PER_WORD];
    uint64_t stride;
    uint32_t on, vp;
};

// The fused phase launch for 11..20 variant positions (two key words): every read's pattern is grouped in three rounds of
// the one-word machinery — its first word numbered in table A, its second word in table B, the PAIR of those numbers
// (one 64-bit word again) in the main table.  Exact by construction: equal pairs <=> equal words <=> equal patterns.
// The half-key tables only number their keys: no counts, no representatives.
struct jl_two_word {
    unsigned long long *key_a, *key_b;   // [table slots] each, emptied after every run
    uint32_t *occ_a, *occ_b;             // the slots a run touched
    uint32_t *n_occ;                     // [2] how many
};

struct jl_done_ent {   // completion word of one window (see done_kernel)
    uint32_t *seq_dev;
    volatile uint32_t *seq_host;
};

struct jl_win_phase {
    const uint8_t *msa;
    uint64_t col_stride, n_reads, reads_pad;
    const uint32_t *vpcols;
    jl_phase_meta *meta;
    uint64_t *keys;
    uint32_t *flagw;
    uint64_t slots_mask;
    unsigned long long *slot_key;
    uint32_t *slot_rep, *slot_count, *occupied, *read_slot;
    uint32_t *blockcat;
    jl_select_args S;
    uint32_t n_blocks, pad_;
};

// Group launches take the argument blocks of their (at most JL_GROUP_MAX) windows BY VALUE, i.e. in the kernel-argument
// segment: pointers loaded from there are known to be global ones, while pointers loaded from a table in device memory
// are generic to the compiler and every access through them becomes a flat access (slower, and never waited for with
// an exact count).
#define JL_GROUP_MAX 8           // windows per call / phase / id launch (their argument blocks are 250-350 bytes each)
#define JL_GROUP_WINDOWS_MAX 32  // windows per group = per pileup launch (56-byte argument blocks)
struct jl_call_group_args { jl_win_call w[JL_GROUP_MAX]; };
struct jl_compact_group_args { jl_win_compact w[JL_GROUP_MAX]; };
struct jl_phase_group_args { jl_win_phase w[JL_GROUP_MAX]; };
struct jl_pileup_group_args { jl_win_pileup w[JL_GROUP_WINDOWS_MAX]; };
// the Fisher stage of a window, evaluated by the pileup workgroup that counted the codon (one workgroup per chunk): the call
// stage's own argument block + the positions by column
struct jl_win_fold {
    jl_win_call c;
    const uint32_t *col_head, *pos_next;
};
struct jl_pileup_fold_group_args { jl_win_pileup w[JL_GROUP_MAX]; jl_win_fold f[JL_GROUP_MAX]; };

// exchange: the heads of the result blocks of up to JL_GATHER_MAX windows copied next to each other (one send buffer,
// one all-gather for the launch)
#define JL_GATHER_MAX 32
struct jl_gather_args { const uint8_t *src[JL_GATHER_MAX]; };
void jl_launch_gather_heads(const uint8_t *const *srcs, uint32_t n, uint8_t *dst, hipStream_t st);
void jl_launch_heads_to_host(const uint8_t *d_region, uint8_t *h_region, uint32_t n_heads, hipStream_t st);

struct jl_comm;

// ---- cross-window phasing with the reads sharded (kernels_xwin.hip)
#define JL_XW_POS_MAX 48u    // owned variant positions per pack launch
#define JL_XW_DST_MAX 8u     // destination ranks per pack launch
#define JL_XW_TAB_MAX 1024u  // exported groups whose haplotypes travel in the kernel arguments
struct jl_xw_pack_args {
    const uint8_t *src[JL_XW_POS_MAX];   // plane 0 of the first column of each owned position in its window, at read 0
    uint64_t src_stride;                 // plane stride of the windows
    uint32_t n_pos, n_dst;
    // dst: where the first plane row of this launch goes; byte_begin / bytes: the slice within a plane row (8 reads a byte);
    // tail_mask: the bits of the slice's last byte that are its reads (0xFF: all eight)
    struct { uint8_t *dst; uint64_t dst_stride, byte_begin, bytes; uint32_t tail_mask, pad_; } d[JL_XW_DST_MAX];
    // the compact matrix's phasing plan, written by the first launch of a step (meta == null: not by this one)
    jl_phase_meta *meta;
    uint32_t *vpcols, *col2pos;
    uint32_t vp_total, kwords, n_var, pad_;
};
struct jl_xw_assign_args {
    uint64_t n_dwords;
    const uint32_t *flagw, *read_slot, *slot_hap;
    uint16_t *read_hap;
    uint32_t n_groups, bits, phased, pad_;
    uint32_t *arrive, *seq_dev;
    volatile uint32_t *seq_host;     // null: no completion word
};
struct jl_xw_hap_table { uint16_t h[JL_XW_TAB_MAX]; };
void jl_launch_xw_pack(const jl_xw_pack_args *a, hipStream_t st);
void jl_launch_xw_assign(const jl_xw_assign_args *a, const uint16_t *host_tab, const uint16_t *d_tab, hipStream_t st);
void jl_launch_xw_fetch(const void *d_src, void *h_dst, uint64_t bytes, uint32_t *arrive, uint32_t *seq_dev, volatile uint32_t *seq_host,
                        hipStream_t st);

// Records uploaded so far by jl_records_append: one run of device arrays, offsets rebased to it.
struct jl_records {
    bool open = false, have_qual = false;
    uint8_t *d_seq = nullptr, *d_qual = nullptr;
    uint32_t *d_cig = nullptr;
    uint64_t *d_co = nullptr, *d_so = nullptr, *d_qo = nullptr;
    int32_t *d_pos = nullptr;
    size_t cap_seq = 0, cap_qual = 0, cap_cig = 0, cap_co = 0, cap_so = 0, cap_qo = 0, cap_pos = 0;
    uint64_t n_reads = 0, n_cig = 0, n_seq = 0, n_qual = 0;
    // does any read need the ingest's launch for long reads (kernels_ingest.hip jl_ingest_read_is_long)?  Found out at the upload for
    // the few reads a CCS sample has with more ops than entries fit; `true` as soon as looking would cost more than the launch
    bool maybe_long = false;
};

struct jl_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;

    // ---- resident MSA: THE format, written directly by every producer (upload, by-row pack, record ingest, synthetic
    // fill) and read by every consumer (pileup, phasing, cross-window exchange, an adopted matrix): per column three BIT
    // PLANES — plane k holds bit k of every read's 3-bit symbol code (A C G T - N ' ' = 0..6), read i in bit i & 7 of byte
    // i >> 3 — i.e. 3 bits per cell.  Plane k of column c at d_msa + (3 c + k) * plane_stride; plane_stride =
    // jl_plane_stride(n_reads) (reads padded to a multiple of 1024 with code 6: whole 128-byte lines per plane), or the
    // caller's for an adopted matrix.  `col_stride` = 4 * plane_stride is the same stride counted in "8 reads per dword"
    // units: the phasing kernels give a lane 8 reads (dword t of a column <=> byte t of each plane), reads_pad = 2 col_stride.
    uint8_t *d_msa = nullptr;
    bool own_msa = false;
    size_t msa_capacity = 0;
    uint64_t plane_stride = 0;
    uint64_t n_reads = 0;
    uint32_t n_cols = 0;
    uint64_t col_stride = 0;
    uint32_t win_begin = 0;

    // ---- insertions per column (fuse-style consensus; off unless jl_msa_track_insertions)
    bool track_insertions = false;
    bool ins_valid = false;
    uint32_t *d_ins_len = nullptr;    // [n_cols][JL_INS_LEN_BINS]
    uint32_t *d_ins_base = nullptr;   // [n_cols][JL_INS_MAX_BASES][4]
    size_t ins_capacity = 0;          // columns

    // ---- aligned records on their way in (jl_records_begin / _append / _finish)
    jl_records rec;
    // scratch of the record ingest INTO this context (kernels_ingest.hip), kept between builds: the reads' runs, the run at
    // every sweep's first column, the units handed on to the planes kernel's second size
    uint2 *d_ing_runs = nullptr;
    uint32_t *d_ing_nruns = nullptr, *d_ing_count = nullptr;
    uint4 *d_ing_desc = nullptr;      // one descriptor per (sweep, read): kernels_ingest.hip
    uint2 *d_ing_slow = nullptr;
    size_t ing_cap_runs = 0, ing_cap_reads = 0, ing_cap_desc = 0, ing_cap_slow = 0;
    bool ing_check_pending = false;   // an ingest ran (or is enqueued) whose verdict on the records has not been read yet

    // ---- phasing sharded by reads: the groups of this matrix exported for the merge (jl_phase_groups_async / _fetch)
    bool phase_export = false;        // the phase launch in flight / last run exported instead of selecting
    uint32_t *d_exp_count = nullptr;  // [exp_cap]
    uint8_t *d_exp_pattern = nullptr; // [exp_cap][exp_stride]
    uint16_t *d_exp_hap = nullptr;    // [exp_cap] the merge's answer on its way to the slots
    uint32_t exp_cap = 0, exp_stride = 0;
    uint32_t exp_n_groups = 0;        // groups the last export produced, once the host has read the count (regroup checks it)
    uint32_t exp_vp = 0;              // ... and its variant positions (0: nothing was phased, no read has flags or a slot)
    bool exp_known = false;
    // a session (capi_xwin.hip) has the groups written into its own block instead (pinned host memory, or the send
    // buffer of the all-gather); null: the context's arrays above
    uint32_t *exp_ext_count = nullptr;
    uint8_t *exp_ext_pattern = nullptr;
    uint32_t *exp_ext_head = nullptr;
    uint32_t exp_ext_cap = 0, exp_ext_stride = 0;
    jl_direct_cols direct = {};       // direct.on: the next fused phase launch reads these columns (see jl_direct_cols)

    // ---- pileup plan (host copies + device arrays)
    std::vector<jl_gene> genes;
    std::vector<uint8_t> refseq;
    bool have_ref = false;
    bool plan_valid = false;
    uint32_t P = 0;
    double default_n_tests = 0.0;
    std::vector<uint32_t> h_pos_gene, h_pos_codon, h_pos_col;
    uint32_t *d_pos_gene = nullptr, *d_pos_codon = nullptr, *d_pos_col = nullptr;
    uint8_t *d_pos_refcfg = nullptr;
    size_t pos_capacity = 0;
    // the positions by column, for the Fisher stage folded into the pileup launch (kernels_pileup.hip): col_head[c] = the first
    // position whose codon begins at column c (all ones: none), pos_next[p] = the next one at the same column (genes that overlap
    // in one frame)
    uint32_t *d_col_head = nullptr, *d_pos_next = nullptr;
    uint8_t *d_guess = nullptr;    // [n_cols + JL_GUESS_PAD] base the codon compare is seeded with (never affects
                                   // results); the zeroed pad lets the kernel fetch it as aligned dwords
    size_t col_capacity = 0;

    // ---- pileup outputs: one zeroed region = col counts [n_cols][6] then hist [n_cols][64]
    uint32_t *d_counts = nullptr;
    uint32_t *d_hist = nullptr;
    size_t counts_words = 0;
    bool pileup_done = false;
    // pileup chunk table (host-built, see build_chunks): chunk k owns columns [c0, c0 + n), n <= pileup_w
    uint32_t pileup_w = 3;
    uint32_t n_chunks = 0;
    uint64_t *d_chunks = nullptr;  // [n_chunks] records {first column, JL_CHUNK_META}: see kernels_pileup.hip
    size_t chunk_capacity = 0;

    // ---- call
    uint64_t *d_called = nullptr;  // [P] mask of called codons
    jl_variant *d_staged = nullptr;  // [P][64] finished rows of the called codons, before the ordered compaction
    uint64_t *d_drm = nullptr;     // [P] optional
    jl_variant *d_variants = nullptr;  // [JL_VARIANT_CAP]
    uint32_t *d_nvar = nullptr;        // [0] rows needed, [1] spare
    bool call_done = false;

    // ---- phase
    jl_phase_meta *d_meta = nullptr;
    uint32_t *d_vpcols = nullptr;   // [JL_VARIANT_CAP]
    uint32_t *d_col2pos = nullptr;  // [n_cols]
    uint8_t *d_varcol = nullptr;    // [n_cols] scratch flags
    uint64_t *d_keys = nullptr;     // [kwords_cap][reads_pad]
    size_t keys_capacity = 0;       // in uint64
    uint32_t keys_words = 0;        // 64-bit words per read the key buffer holds
    uint32_t last_min_reads = 10;
    uint32_t *d_flagw = nullptr;    // [reads_pad/8] nibble flags
    uint32_t *d_blockcat = nullptr; // [phase workgroups][4] read categories of each workgroup's reads (summed by the selection)
    uint32_t *d_read_slot = nullptr;  // [reads_pad]
    uint16_t *d_read_hap = nullptr;   // [reads_pad]
    uint32_t *d_slot_rep = nullptr, *d_slot_count = nullptr;  // [M]
    uint64_t *d_slot_key = nullptr;                           // [M] single-word keys (fused path)
    uint32_t *d_slot_hap = nullptr;                           // [M] haplotype id of a table slot (32-bit: write-through stores)
    uint32_t *d_occupied = nullptr;                           // [reads_pad]
    uint64_t table_slots = 0;
    size_t reads_capacity = 0;
    uint32_t *d_hap_count = nullptr;   // [JL_MAX_HAPLOTYPES]
    uint8_t *d_hap_pattern = nullptr;  // [JL_MAX_HAPLOTYPES][JL_VARIANT_CAP]
    uint8_t *d_hit = nullptr;          // [JL_VARIANT_CAP][JL_MAX_HAPLOTYPES]
    uint32_t *d_cooc = nullptr;        // [cooc_cap][cooc_cap]
    uint32_t cooc_cap = 256;
    bool phase_done = false;
    bool no_fold = false;        // a folded phase launch of this context timed out once (its workgroups were not resident together):
                                 // from then on the per-read ids come from a launch of their own (jl_phase_rerun_unfolded)
    uint32_t fold_reruns = 0;    // how often that happened
    bool phase_generic = false;  // multi-word pipeline selected (more than 20 positions, or results beyond the fused selection)
    bool phase_two = false;      // 11..20 positions: the two-word fused launch (jl_two_word)
    uint64_t *d_slot_key_a = nullptr, *d_slot_key_b = nullptr;   // its half-key tables [table slots]
    uint32_t *d_occ_a = nullptr, *d_occ_b = nullptr;             // [reads_pad]
    uint64_t two_slots = 0;                                      // table size the half-key tables were made for

    // ---- whole-path run: result pack, pinned mirrors, captured graph
    jl_pack *d_pack = nullptr;        // [2]: run n writes block n & 1 (an exchange may still read the other one)
    jl_pack *h_pack = nullptr;        // pinned
    uint16_t *h_read_hap = nullptr;   // pinned, [reads_pad]
    size_t h_read_hap_cap = 0;
    jl_pack *pack_mirror = nullptr;     // where kernels mirror the result block (h_pack during jl_run_async)
    uint16_t *read_hap_out = nullptr;   // where phase_assign_kernel writes (h_read_hap when the host wants the ids)
    bool pack_valid = false;          // the last stage calls were one jl_run_async
    bool run_read_hap = false;
    // completion without a HIP sync: the last kernel of a run bumps d_sync[0] and stores it to *h_seq (pinned)
    uint32_t *d_sync = nullptr;       // [16] zeroed once: [0] runs completed, [1..] arrival counters of fused kernels
    volatile uint32_t *h_seq = nullptr;  // pinned [16]: [0] the run word; [4] the word of jl_fetch_to_host
    uint8_t *h_scratch = nullptr;     // pinned: small device arrays on their way to the host (a first pageable copy of a
    size_t h_scratch_cap = 0;         // process costs the runtime milliseconds: staging buffers, pinning the target)
    uint32_t fetches = 0;             // completed jl_fetch_to_host calls (device word d_sync[8], host word h_seq[4])
    uint32_t runs_launched = 0;
    uint32_t exch_pending = 0;        // exchanges requested and not yet collected: each still reads one of the two result blocks
    std::vector<uint32_t> exch_runs;  // ... and the runs (values of runs_launched) whose blocks they read
    hipStream_t run_stream = nullptr;  // where the last run was enqueued (the ctx stream, or a group's)
    bool pileup_clock = false;        // jl_run_pileup_clock: clock nodes around the pileup of a run -> h_seq[8..11] (two 64-bit stamps)
    uint32_t clock_run = 0;           // value of runs_launched of the run those stamps belong to (0: none)
    uint64_t *d_timeline = nullptr;   // JL_TIMELINE=1 only: [JL_TIMELINE_ROWS][JL_TIMELINE_SLOTS] device clock stamps
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    std::vector<uint8_t> graph_sig;
    std::vector<uint8_t> graph_seen;   // signature of the last eager run (a configuration is captured on its second run)
    uint64_t alloc_version = 0;       // bumped by every (re)allocation: captured pointers go stale
    uint64_t plan_version = 0;
    int pileup_blocks_per_cu[16] = {0};  // occupancy per kernel variant, queried once

    // ---- timing
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

// status helpers -----------------------------------------------------------------------------
int jl_fail(jl_ctx *ctx, int status, const char *fmt, ...);
#define JL_HIP(ctx, expr)                                                                            \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return jl_fail(ctx, e_ == hipErrorOutOfMemory ? JL_ERR_MEMORY : JL_ERR_DEVICE, "%s: %s", \
                           #expr, hipGetErrorString(e_));                                            \
    } while (0)

// kernel launchers (defined in the .hip files) -------------------------------------------------
void jl_launch_guess(jl_ctx *ctx, hipStream_t st);
void jl_launch_pileup(jl_ctx *ctx, hipStream_t st);
// The pileup launch with the Fisher stage in its epilogue (runs only; every chunk counted by ONE workgroup): false = not
// possible for this shape (the reads of a column are split over several workgroups), nothing was launched.
bool jl_pileup_can_fold(jl_ctx *ctx);
bool jl_fold_enabled(void);
void jl_launch_pileup_fold(jl_ctx *ctx, hipStream_t st, const jl_win_call *call);
void jl_fill_win_fold(jl_ctx *ctx, const jl_win_call *call, jl_win_fold *f);
int jl_launch_pileup_fold_group(jl_ctx *const *ctxs, uint32_t n_win, const jl_win_pileup *h_wins, const jl_win_fold *h_fold, uint32_t max_chunks, hipStream_t st);
uint32_t jl_pileup_rsplit(jl_ctx *ctx);
bool jl_pileup_needs_zero(jl_ctx *ctx);
void jl_prepare_pileup(jl_ctx *ctx);
void jl_launch_call(jl_ctx *ctx, hipStream_t st, const jl_params *prm, double n_tests, bool use_drm, bool with_meta);
void jl_launch_compact(jl_ctx *ctx, hipStream_t st, bool plan, bool pack, bool signal);
bool jl_launch_phase(jl_ctx *ctx, hipStream_t st, uint32_t min_reads, bool planned, bool from_called, bool signal);
void jl_fill_call_args(jl_ctx *ctx, const jl_params *prm, double n_tests, jl_call_args *A);
// group runs: fill one window's argument block / launch a stage once for `n_win` <= JL_GROUP_MAX windows (the blocks
// travel by value in the kernel arguments)
void jl_fill_win_pileup(jl_ctx *ctx, jl_win_pileup *w);
void jl_fill_win_call(jl_ctx *ctx, const jl_params *prm, double n_tests, bool use_drm, bool with_meta, jl_win_call *w);
void jl_fill_win_compact(jl_ctx *ctx, bool plan, bool pack, bool signal, jl_win_compact *w);
bool jl_fill_win_phase(jl_ctx *ctx, uint32_t min_reads, bool signal, uint32_t fold_budget, bool from_called, jl_win_phase *w);
int jl_launch_pileup_group(jl_ctx *const *ctxs, uint32_t n_win, const jl_win_pileup *h_wins, uint32_t max_chunks, hipStream_t st);
void jl_launch_call_group(const jl_win_call *h_wins, uint32_t n_win, uint32_t max_blocks, hipStream_t st);
void jl_launch_compact_group(const jl_win_compact *h_wins, uint32_t n_win, hipStream_t st);
void jl_launch_phase_group(const jl_win_phase *h_wins, uint32_t n_win, uint32_t max_blocks, hipStream_t st);
void jl_launch_assign_group(const jl_win_phase *h_wins, uint32_t n_win, uint32_t max_read_blocks, bool to_host, hipStream_t st);
void jl_launch_synth(jl_ctx *ctx, const jl_synth_plan *plan, const uint8_t *d_ref, uint32_t col0);
void jl_launch_pack_rows(jl_ctx *ctx, const uint8_t *d_rows);
// interchange format (column-packed nibbles, include/juliet_hip.h) <-> the resident planes, `n` columns from column c0 on
void jl_launch_nibbles_to_planes(jl_ctx *ctx, const uint8_t *d_nib, uint64_t nib_stride, uint32_t c0, uint32_t n, uint32_t *d_bad);
void jl_launch_planes_to_nibbles(jl_ctx *ctx, uint8_t *d_nib, uint64_t nib_stride, uint32_t c0, uint32_t n);
void jl_launch_done(jl_ctx *ctx);
void jl_launch_done_on(jl_ctx *ctx, hipStream_t st);
// a run whose folded phase launch gave up waiting (meta.overflow & 32): the phasing stage again, unfolded, behind everything
// on the run's stream; blocks until it is done.  The call stage's results are still resident.
extern "C" int jl_phase_rerun_unfolded(jl_ctx *ctx);
void jl_launch_done_group(const jl_done_ent *d_ents, uint32_t n, hipStream_t st);
void jl_launch_stamp(jl_ctx *ctx, uint32_t slot);
extern "C" int jl_run_prepare(jl_ctx *ctx, const jl_gene *genes, uint32_t n_genes, const uint8_t *refseq, uint32_t ref_len,
                              const jl_params *prm, const uint64_t *drm_masks, int phasing, uint32_t min_reads,
                              int want_read_hap, double *n_tests_out);
extern "C" void jl_run_finish(jl_ctx *ctx, int phasing, int want_read_hap);
extern "C" int jl_run_wait_impl(jl_ctx *ctx);
extern "C" int jl_run_wait_seq(jl_ctx *ctx, uint32_t want);
void jl_launch_consensus(jl_ctx *ctx, uint8_t *d_out);
void jl_launch_clock(jl_ctx *ctx, hipStream_t st, uint32_t which);   // h_seq[8 + 2 which ..] = the device's 100 MHz clock
uint32_t jl_ingest_short_ops();
bool jl_ingest_read_is_long(const uint32_t *cigar, uint64_t n_ops);
void jl_launch_ingest(jl_ctx *ctx, const int32_t *d_pos, const uint32_t *d_cigar, const uint64_t *d_cig_off,
                      const uint8_t *d_seq4, const uint64_t *d_seq_off, const uint8_t *d_qual,
                      const uint64_t *d_qual_off, uint32_t min_qv, uint2 *d_runs, uint32_t *d_nruns, uint4 *d_desc,
                      uint32_t *d_slow_count, uint2 *d_slow, bool maybe_long, uint64_t seq_bytes, uint64_t n_entries);
uint32_t jl_ingest_sweeps(uint32_t n_cols);
size_t jl_ingest_slow_room(const jl_ctx *ctx);
extern "C" int jl_ingest_verdict(jl_ctx *ctx);
void jl_launch_regroup(jl_ctx *ctx, const uint16_t *d_hap_of_group, uint32_t n_groups, uint32_t n_haplotypes, bool phased);
void jl_launch_insertions(jl_ctx *ctx, const int32_t *d_pos, const uint32_t *d_cigar, const uint64_t *d_cig_off,
                          const uint8_t *d_seq4, const uint64_t *d_seq_off);
void jl_launch_fisher_eval(jl_ctx *ctx, uint32_t n, const uint32_t *a, const uint32_t *c, const uint32_t *cov, int tail,
                           double *p, double *lp);
// per-read ids in their packed form (4 / 8 / 16 bits, see JL_ID4_MAX_H) expanded to 16-bit ids on the host
extern "C" void jl_expand_ids(const void *packed, uint32_t bits, uint64_t n_reads, uint16_t *out);
int jl_msa_alloc_strided(jl_ctx *ctx, uint64_t n_reads, uint32_t n_cols, uint64_t plane_stride, uint32_t win_begin);
// capi_xwin.hip: buffers of an exporting phase run whose plan (vp positions at columns 3k) a kernel of the caller writes
int jl_phase_groups_prepare(jl_ctx *ctx, uint32_t vp);
// the variant table of a context's last call stage on the host: a pointer into the pinned result block when the run left
// it there, else copied into `scratch`
int jl_ctx_table_host(jl_ctx *ctx, std::vector<jl_variant> *scratch, const jl_variant **rows, uint32_t *n);
// jl_run_wait_seq without touching ctx->err (threads other than the context's owner)
int jl_run_wait_seq_quiet(jl_ctx *ctx, uint32_t want, hipStream_t stream);
// `bytes` of device memory to `dst` (pageable is fine) behind everything enqueued on the context's run stream: a copy kernel into
// the context's pinned scratch + a completion word, no runtime copy path
int jl_fetch_to_host(jl_ctx *ctx, const void *d_src, size_t bytes, void *dst, size_t readable);
