/*
 * juliet_hip.h — C ABI of libjuliet_hip.so: juliet's call+phase hot path on MI355X (gfx950).
 *
 * What this replaces.  The reference exposes NO library, plugin or FFI surface for this path — only the
 * `juliet` process boundary (doc/JULIET.md:62-66) — and ships no source, so there is no
 * reference interface file:line for an entry point to mirror.  Each entry point below therefore cites the
 * reference TEXT whose behaviour it implements (doc/JULIET.md as J:line) and the docs/SPEC.md section that
 * fixes the details the text leaves open.  A host (the C++ `juliet` front end in minorseq_amd/host/, or any
 * FFI: ctypes in minorseq_amd/capi.py, cgo/JNI stubs in INTEGRATION.md) drives the path through these calls.
 *
 * Conventions.  Plain C, fixed-width PODs, no exceptions cross the boundary.  Every function returns
 * JL_OK (0) or a negative jl_status; jl_last_error(ctx) has the detail.  The caller owns every host
 * buffer; the ctx owns device memory unless a buffer is adopted.  One ctx per (thread, device); all work
 * of a ctx is ordered on one HIP stream (its own, or the caller's — e.g. torch's current stream).
 * `*_async` calls only enqueue; results are valid after jl_sync() or a `*_fetch`/blocking call.
 * There is no CPU fallback: without a usable gfx950 device every call fails with JL_ERR_DEVICE.
 */
#ifndef JULIET_HIP_H
#define JULIET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: the functions declared between this push and the pop at the end of
 * the file ARE its export list (tests/test_capi_exports.py checks the equality both ways). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define JL_ABI_VERSION 5

/* symbol codes of the MSA (SPEC §1; J:99-100, 256-259, 372-381) */
enum { JL_SYM_A = 0, JL_SYM_C = 1, JL_SYM_G = 2, JL_SYM_T = 3, JL_SYM_GAP = 4, JL_SYM_MASK = 5, JL_SYM_NONE = 6 };

typedef enum {
    JL_OK = 0,
    JL_ERR_ARG = -1,      /* bad argument / shape */
    JL_ERR_DEVICE = -2,   /* no gfx950 device, HIP error */
    JL_ERR_MEMORY = -3,   /* allocation failed */
    JL_ERR_STATE = -4,    /* call order: e.g. jl_call before jl_pileup */
    JL_ERR_OVERFLOW = -5, /* an output capacity was exceeded (n_out holds the needed size) */
    JL_ERR_COMM = -6      /* RCCL failure */
} jl_status;

enum { JL_MAX_HAPLOTYPES = 702 /* [A-Z][a-z]?  J:198 */, JL_HAP_INSUFFICIENT = 0xFFFE, JL_HAP_DAMAGED = 0xFFFF };

typedef struct jl_ctx jl_ctx;
typedef struct jl_comm jl_comm;

/* gene / ORF: 1-based [begin, end) in alignment space (J:134-136) */
typedef struct {
    uint32_t begin;
    uint32_t end;
} jl_gene;

/* ErrorEstimates (SPEC §5; J:40-41, 221-225).  Values are UNPINNED, hence parameters. */
typedef struct {
    double match;
    double substitution; /* per target base */
    double deletion;     /* reserved: indels are ignored (J:26-27) */
} jl_error_model;

typedef struct {
    double alpha;           /* significance level after Bonferroni; default 0.01 */
    double n_tests;         /* Bonferroni factor; <= 0 selects sum of codons over all genes passed to jl_pileup */
    jl_error_model err;
    int32_t expected_round; /* 0 ceil (default), 1 floor, 2 nearest */
    int32_t tail;           /* 0 one-sided greater (default), 1 two-sided */
    double min_perc;        /* --min-perc: keep 100*count/coverage >  min_perc; < 0 disables (J:342-344) */
    double max_perc;        /* --max-perc: keep 100*count/coverage <  max_perc; < 0 disables (J:352-354) */
} jl_params;

/* one called variant codon (SPEC §6; J:94-98).  48 bytes, the all-gather payload row. */
typedef struct {
    uint32_t gene;      /* index into the genes passed to jl_pileup */
    uint32_t codon_pos; /* 1-based amino-acid position in the gene */
    uint32_t col;       /* window column of the codon's first base */
    uint8_t ref_codon;  /* 0..63 = 16*b0+4*b1+b2 */
    uint8_t codon;
    uint16_t flags;
    uint32_t count;
    uint32_t coverage;
    uint32_t expected;
    uint32_t pad_;
    double p_value; /* Bonferroni-adjusted, <= 1 */
    double log_p;   /* ln of the unadjusted p */
} jl_variant;

/* read categories (J:372-381) and sizes of the phasing result */
typedef struct {
    uint32_t reported_reads;
    uint32_t insufficient_reads;
    uint32_t damaged_reads;
    uint32_t marginal_gap;
    uint32_t marginal_heteroduplex;
    uint32_t marginal_partial;
    uint32_t n_positions;  /* Vp: distinct variant columns */
    uint32_t n_haplotypes; /* H */
} jl_phase_summary;

/* synthetic aligned-CCS generator (bench + tests) */
typedef struct {
    uint64_t seed;
    double sub_rate;      /* per base, uniform over the other three bases */
    double del_rate;      /* '-' */
    double mask_rate;     /* 'N' */
    double partial_rate;  /* fraction of reads that start late / end early */
    uint32_t minor_permille[4]; /* frequency of the four minor haplotypes, in 1/1000 of the reads */
    uint32_t reserved;
} jl_synth_params;

/* ---------------------------------------------------------------- context */

int jl_abi_version(void);
const char *jl_strerror(int status);
/* Number of usable gfx950 devices (0 if none). Does not create a context. */
int jl_device_count(void);
/* `stream` = NULL: the ctx creates its own non-blocking stream; else a hipStream_t owned by the caller. */
int jl_ctx_create(int device, void *stream, jl_ctx **out);
/* The hipStream_t this context orders its work on (its own or the caller's).  Contexts of one device that a thread drives
 * one after the other may share one: a stream of its own costs the runtime about 8 ms to create. */
void *jl_ctx_stream(const jl_ctx *ctx);
void jl_ctx_destroy(jl_ctx *ctx);
const char *jl_last_error(const jl_ctx *ctx);
/* Waits for everything enqueued for this context: its own stream and, after a group run, the group's. */
int jl_sync(jl_ctx *ctx);

/* ---------------------------------------------------------------- MSA residency */

/* THE resident format (ABI 5): per column three BIT PLANES — plane k holds bit k of every read's 3-bit symbol code, read i
 * in bit i & 7 of byte i >> 3; plane k of column c at base + (3 c + k) * plane_stride; reads past n_reads are padding with
 * code 6 (plane 0 clear, planes 1 and 2 set).  3 bits per cell: 112.5 MB for 100k reads x 3 kb.  Every producer (upload,
 * pack_rows, record ingest, synthetic fill) writes the planes directly and every stage (pileup, phasing, the exchange of
 * cross-window phasing) reads them; there is no second copy.  The column-packed NIBBLE layout of earlier ABI versions
 * (read i in byte i / 2, low nibble even; column stride jl_col_stride) remains the host-side INTERCHANGE format of
 * jl_msa_upload and jl_msa_download only: it is converted on the device, a bounded run of columns at a time. */

/* Bytes per column of the interchange (nibble) format for n_reads reads (ceil(n/2) rounded up to 128). */
uint64_t jl_col_stride(uint64_t n_reads);
/* Bytes per plane of the resident format the library allocates: ceil(n/1024) * 128 (whole 128-byte lines). */
uint64_t jl_plane_stride(uint64_t n_reads);
/* A host matrix in the interchange format, [n_cols][col_stride] nibbles, into resident planes owned by the ctx; symbol
 * codes outside 0..6 are rejected (JL_ERR_ARG); whatever the buffer holds past read n_reads - 1 of a column is ignored. */
int jl_msa_upload(jl_ctx *ctx, const uint8_t *colpacked, uint64_t n_reads, uint32_t n_cols, uint64_t col_stride,
                  uint32_t win_begin);
/* Allocate an uninitialised resident matrix (for jl_synth_fill / jl_msa_pack_rows). */
int jl_msa_alloc(jl_ctx *ctx, uint64_t n_reads, uint32_t n_cols, uint32_t win_begin);
/* Use caller-owned device memory (e.g. a torch tensor) holding the resident format — [n_cols][3][plane_stride] bytes,
 * 16-byte aligned, plane_stride a multiple of 16 and >= ceil(n_reads / 8), padding reads = code 6 — as the resident
 * matrix; never copied, not freed by the ctx, and the caller may rewrite it between runs. */
int jl_msa_adopt(jl_ctx *ctx, void *d_planes, uint64_t n_reads, uint32_t n_cols, uint64_t plane_stride,
                 uint32_t win_begin);
/* Device-side transpose of a host by-row matrix uint8[n_reads][n_cols] (codes 0..6) into the resident layout. */
int jl_msa_pack_rows(jl_ctx *ctx, const uint8_t *rows, uint64_t n_reads, uint32_t n_cols, uint32_t win_begin);
/*
 * Aligned records -> resident matrix entirely on the device: cigar expansion (= X I D S H N; M is rejected,
 * J:53), insertions dropped and deletions '-' (J:26-27), optional QV masking to N (J:256-259), transpose.
 * pos[r]: 0-based leftmost reference position; cigar words are BAM's (len << 4 | op), read r owns
 * cigar[cig_off[r] .. cig_off[r+1]); seq4 holds BAM's 4-bit bases, read r starting at byte seq_off[r];
 * qual (optional, with qual_off) one byte per base, 0xFF = absent; bases with qual < min_qv become N (min_qv is
 * taken as at most 127: BAM qualities end at 93).
 */
int jl_msa_ingest_records(jl_ctx *ctx, uint64_t n_reads, uint32_t n_cols, uint32_t win_begin, const int32_t *pos,
                          const uint32_t *cigar, const uint64_t *cig_off, const uint8_t *seq4, const uint64_t *seq_off,
                          const uint8_t *qual, const uint64_t *qual_off, uint32_t min_qv);
/*
 * The same in pieces, so a decoder can hand over each batch of records while it inflates the next one (the upload
 * then hides under the decode; the kernels run in jl_records_finish, once the window is known).  The hints size the
 * device arrays up front (they grow when exceeded; qual_bytes_hint 0 = no qualities expected).  A chunk's offsets
 * index ITS arrays (read r of the chunk owns cigar[cig_off[r] .. cig_off[r+1]) etc., cig_off[0] need not be 0); the
 * host arrays may be reused as soon as jl_records_append returns.  Either every chunk carries qualities or none.
 * Validation and errors are those of jl_msa_ingest_records, which is begin + one append + finish.  One thread at
 * a time per context.
 */
int jl_records_begin(jl_ctx *ctx, uint64_t reads_hint, uint64_t cigar_words_hint, uint64_t seq_bytes_hint,
                     uint64_t qual_bytes_hint);
int jl_records_append(jl_ctx *ctx, uint64_t n_reads, const int32_t *pos, const uint32_t *cigar, const uint64_t *cig_off,
                      const uint8_t *seq4, const uint64_t *seq_off, const uint8_t *qual, const uint64_t *qual_off);
int jl_records_finish(jl_ctx *ctx, uint32_t n_cols, uint32_t win_begin, uint32_t min_qv);
/* One upload, several column windows (`juliet --windows K`: the BAM is decoded and uploaded once): the resident matrix
 * of `window` — another context of the same device — from the records appended to `records`, which stay until
 * jl_records_drop (or jl_records_finish on `records` itself). */
int jl_records_window(jl_ctx *records, jl_ctx *window, uint32_t n_cols, uint32_t win_begin, uint32_t min_qv);
/* The same, enqueued only (three launches on the window's stream, no allocation once a window of this shape was built on
 * `window`): the matrix is complete when the window's stream gets there, so a run enqueued behind it (jl_run_async on
 * `window`) reads it — records -> planes -> call + phase without a host round trip in between. */
int jl_records_window_async(jl_ctx *records, jl_ctx *window, uint32_t n_cols, uint32_t win_begin, uint32_t min_qv);
int jl_records_drop(jl_ctx *records);
/*
 * Insertions are not part of the matrix (J:26-27) but `fuse` "includes in-frame insertions with a certain distance to
 * each other" (doc/FUSE.md:19): with tracking on, jl_msa_ingest_records also counts them per window column — an
 * insertion sits BEFORE the column of the next reference base.  len_hist[n_cols][32]: insertions by length (31 = longer
 * than 30); base_counts[n_cols][30][4]: inserted bases A C G T by offset.  Either pointer may be NULL.
 */
int jl_msa_track_insertions(jl_ctx *ctx, int on);
int jl_insertions_fetch(jl_ctx *ctx, uint32_t *len_hist, uint32_t *base_counts);
/* The resident matrix back on the host in the interchange format, [n_cols][jl_col_stride(n_reads)] nibbles (tests). */
int jl_msa_download(jl_ctx *ctx, uint8_t *colpacked, uint64_t bytes);
/* Fill the resident matrix with synthetic reads, on the device. `ref` = n_cols base codes (host). */
int jl_synth_fill(jl_ctx *ctx, const jl_synth_params *sp, const uint8_t *ref);
/* The same reads seen through a WINDOW of a longer reference (BASELINE.json configs[3]/[4]: one reference whose
 * reads span every window): `ref` = ref_len base codes of the whole reference, the resident matrix is its columns
 * [win_begin, win_begin + n_cols).  Windows filled with the same parameters hold the same reads. */
int jl_synth_fill_window(jl_ctx *ctx, const jl_synth_params *sp, const uint8_t *ref, uint32_t ref_len);

/* ---------------------------------------------------------------- call */

/*
 * Column pileup + per-codon histograms for every evaluated position of `genes` (SPEC §2-3; J:94-100).
 * `refseq`: base codes of the whole reference (0..3, other = non-ACGT), 0-based, or NULL for
 * majority-codon mode (J:133-134).  Enqueues; returns without waiting.
 */
int jl_pileup_async(jl_ctx *ctx, const jl_gene *genes, uint32_t n_genes, const uint8_t *refseq, uint32_t ref_len);
/* Number of evaluated positions P of the last jl_pileup_async. */
uint32_t jl_n_positions(const jl_ctx *ctx);
/*
 * Wait, then copy out: col_counts[n_cols][6] (A C G T - N), pos_gene/pos_codon/pos_col[P] (codon_pos 1-based),
 * hist[P][64], coverage[P].  Any pointer may be NULL.
 */
int jl_pileup_fetch(jl_ctx *ctx, uint32_t *col_counts, uint32_t *pos_gene, uint32_t *pos_codon, uint32_t *pos_col,
                    uint32_t *hist, uint32_t *coverage);
/*
 * By-product of the pileup: per-column consensus (majority of A C G T -), the part of `fuse`
 * (doc/FUSE.md:17-20) that needs no insertion tracking.  out[n_cols]: 0..3 base, 4 = majority deletion
 * ("major deletions are being removed"), 5 = column without coverage.  Blocking.
 */
int jl_consensus_fetch(jl_ctx *ctx, uint8_t *out);
/*
 * Reference/majority codon, error model, Fisher's exact x Bonferroni, filters, variant table
 * (SPEC §4-7; J:38-42).  `drm_masks`: optional [P] 64-bit codon masks; with --drm-only a codon is kept
 * only if its bit is set (J:370); NULL disables.  Table stays on the device; enqueues only.
 */
int jl_call_async(jl_ctx *ctx, const jl_params *prm, const uint64_t *drm_masks);
/* Wait and copy the table out.  *n_out = rows produced; JL_ERR_OVERFLOW if > cap (first cap rows are valid). */
int jl_call_fetch(jl_ctx *ctx, jl_variant *out, uint32_t cap, uint32_t *n_out);
/* Device address and capacity (rows) of the resident variant table and of its row counter (uint32). */
int jl_variant_table_device(jl_ctx *ctx, void **d_rows, void **d_count, uint32_t *cap_rows);

/* ---------------------------------------------------------------- phase */

/*
 * Read x variant phasing (SPEC §8; J:192-211, 253-254, 372-381).  `variants` = host table to phase
 * against (e.g. all-gathered and filtered), or NULL to use the table jl_call_async left on the device.
 * Enqueues only.
 */
int jl_phase_async(jl_ctx *ctx, const jl_variant *variants, uint32_t n_var, uint32_t min_reads);
/*
 * Wait and copy out.  pos_cols[Vp]; hap_count[H]; hap_pattern[H][Vp] (codon indices);
 * hit[V][H] (J:207-209); read_hap[n_reads]; cooc[V][V].  Any pointer may be NULL; capacities are the
 * caller's: pos_cols/hap_pattern rows are sized with cap_var, hap_* with JL_MAX_HAPLOTYPES.
 */
int jl_phase_fetch(jl_ctx *ctx, jl_phase_summary *summary, uint32_t *pos_cols, uint32_t *hap_count,
                   uint8_t *hap_pattern, uint8_t *hit, uint16_t *read_hap, uint32_t *cooc, uint32_t cap_var);

/* ---------------------------------------------------------------- the whole path in one enqueue */

/*
 * pileup -> call (-> phase) -> one small result copy, enqueued as a captured HIP graph that is replayed
 * while genes / reference / parameters / buffers are unchanged (what `juliet in.bam out.json` does per
 * window, J:62-66, 195).  Results are then read with jl_call_fetch / jl_phase_fetch, which decode the
 * pinned result block without further device traffic.  `want_read_hap` also copies the per-read haplotype
 * ids (the haplotype block's read lists, J:209-211).  Set the environment variable JL_NO_GRAPH to run eagerly.
 */
int jl_run_async(jl_ctx *ctx, const jl_gene *genes, uint32_t n_genes, const uint8_t *refseq, uint32_t ref_len,
                 const jl_params *prm, const uint64_t *drm_masks, int phasing, uint32_t min_reads, int want_read_hap);

/*
 * Completion of the last jl_run_async without a HIP synchronisation call: the run's last kernel stores a
 * sequence word into pinned host memory behind a system-scope fence and these calls read it (a
 * hipStreamSynchronize after a graph launch costs 10-18 us of host time on ROCm 7.2 even when the work has
 * long finished).  jl_run_wait spins until the results are on the host; jl_run_done never blocks (1 done, 0 not).
 * jl_call_fetch / jl_phase_fetch wait the same way, so calling these first is optional.
 */
int jl_run_wait(jl_ctx *ctx);
int jl_run_done(jl_ctx *ctx);

/*
 * Zero-copy view of the results of the last jl_run_async: pointers into the context's pinned result block
 * (host memory the kernels stored into directly), valid until the next jl_run_async / stage call on this
 * context.  `complete` = 0 means some part did not fit the fixed-size block (more than 128 variants /
 * positions / haplotypes): use jl_call_fetch / jl_phase_fetch then.  Waits like jl_run_wait.
 */
typedef struct {
    uint32_t complete;      /* 1: every field below is valid */
    uint32_t n_variants;    /* rows of `variants` (total called; J:94-98) */
    uint32_t phased;        /* 1 if the run included phasing */
    uint32_t n_positions;   /* Vp */
    uint32_t n_haplotypes;  /* H */
    uint32_t n_var_phase;   /* rows the hit / co-occurrence matrices cover */
    uint64_t n_reads;
    jl_phase_summary summary;
    const jl_variant *variants;
    const uint32_t *pos_cols;     /* [Vp] */
    const uint32_t *hap_count;    /* [H] */
    const uint8_t *hap_pattern;   /* [H][Vp] */
    const uint8_t *hit;           /* [n_var_phase][H] (haplotype_hit, J:207-209) */
    const uint32_t *cooc;         /* [n_var_phase][n_var_phase], NULL if it did not fit */
    /* Per-read haplotype ids (the haplotype block's read lists, J:209-211), NULL / 0 unless the run asked for them.
     * They cross PCIe in the narrowest code that holds H: read_hap_bits = 4 (H <= 14: haplotype 0..13, 14 = insufficient
     * coverage, 15 = damaged; read i in bits 4*(i&1).. of byte i/2), 8 (H <= 254: 254 insufficient, 255 damaged) or
     * 16 (the ids themselves, JL_HAP_INSUFFICIENT / JL_HAP_DAMAGED).  `read_hap` is set only for 16-bit ids;
     * jl_phase_fetch and jl_expand_read_hap give 16-bit ids for every width. */
    const uint16_t *read_hap;
    const void *read_hap_packed;
    uint32_t read_hap_bits;
    uint32_t reserved;
} jl_run_view;
int jl_run_view_get(jl_ctx *ctx, jl_run_view *out);
/* Packed per-read ids of a view (read_hap_packed, read_hap_bits) -> out[n_reads] 16-bit ids.  Host only. */
int jl_expand_read_hap(const void *packed, uint32_t bits, uint64_t n_reads, uint16_t *out);

/*
 * Group runs: the whole path for SEVERAL (at most 32) resident windows (one context each, same device) in a few launches —
 * per stage ONE launch for up to eight windows (blockIdx.z = window): counting, the Fisher stage, phasing, per-read ids.  A 150 MB window is too short a stream to hide a launch's ramp and drain, and its phasing stage
 * is a latency chain that occupies a hardware queue while doing little; grouped, the pileup runs at the rate of one long
 * stream and the latency chains of all windows overlap.  Groups of more than eight windows are pipelined inside the
 * launch: the counting of the next eight runs beside the phasing of the previous eight.
 * Results are per window and identical to jl_run_async on each context: every context keeps its own result block,
 * per-read ids and completion word, so jl_run_wait, jl_run_done, jl_run_view_get, jl_call_fetch, jl_phase_fetch and
 * jl_allgather_variants_async apply unchanged.  All windows get the same genes / reference / parameters (they are windows
 * of one reference, or samples of one amplicon); phasing may be off (call only).  A window with more than 10 variant
 * positions is flagged as in jl_run_async (its fetch calls then re-run the multi-word pipeline).  Every window is counted
 * by one block per column chunk, so windows of millions of reads are better run one by one.
 * The contexts' own streams must be idle (uploads finished).
 */
typedef struct jl_group jl_group;
int jl_group_create(jl_ctx *const *ctxs, uint32_t n_ctx, jl_group **out);
void jl_group_destroy(jl_group *group);
const char *jl_group_last_error(const jl_group *group);
int jl_group_run_async(jl_group *group, const jl_gene *genes, uint32_t n_genes, const uint8_t *refseq, uint32_t ref_len,
                       const jl_params *prm, int phasing, uint32_t min_reads, int want_read_hap);
/* The same with --drm-only masks per window (J:370): drm_masks[k] = [P] 64-bit codon masks of window k as in
 * jl_call_async, or NULL for a window without; drm_masks itself may be NULL. */
int jl_group_run_masked_async(jl_group *group, const jl_gene *genes, uint32_t n_genes, const uint8_t *refseq,
                              uint32_t ref_len, const jl_params *prm, const uint64_t *const *drm_masks, int phasing,
                              uint32_t min_reads, int want_read_hap);
/* Timing hook (bench): average device time in ms of the grouped pileup launch alone, `reps` back-to-back launches
 * rotating over `groups` (each must have run once); `bytes_per_launch` = algorithmic bytes of one launch of groups[0]: 3 bits per cell,
 * every cell read once. */
/* The results of the last group run, one view per window (the group's context order): jl_run_view_get on every context in
 * ONE call — waits for each window's completion word in turn.  out[cap]; *n = the group's windows.  A window whose view
 * fails ends the call with its status (jl_group_last_error names it). */
int jl_group_views(jl_group *group, jl_run_view *out, uint32_t cap, uint32_t *n);
int jl_group_time_pileup(jl_group *const *groups, uint32_t n_groups, uint32_t reps, float *ms_avg, uint64_t *bytes_per_launch);

/* ---------------------------------------------------------------- numerics self-check */

/*
 * Evaluate the device's Fisher routine on `n` tables [[a, cov-a], [c, cov-c]] (host arrays in and out):
 * p[i] = P(X >= a[i]) unadjusted, log_p[i] = ln p.  Lets a host pin the device numerics against its own
 * golden vectors (tests/golden/fisher_golden.json).
 */
int jl_fisher_eval(jl_ctx *ctx, const uint32_t *a, const uint32_t *c, const uint32_t *cov, uint32_t n, double *p,
                   double *log_p);
/* The same with the tail of jl_params: 0 = P(X >= a), 1 = two-sided. */
int jl_fisher_eval_tail(jl_ctx *ctx, const uint32_t *a, const uint32_t *c, const uint32_t *cov, uint32_t n, int tail,
                        double *p, double *log_p);

/* ---------------------------------------------------------------- timing hooks (bench) */

/* Latency of the whole path at this boundary, host clock: `reps` runs one after the other, each jl_run_async ->
 * jl_run_view_get (which waits for the completion word) before the next is launched; average ms per run. */
int jl_time_run(jl_ctx *ctx, const jl_gene *genes, uint32_t n_genes, const uint8_t *refseq, uint32_t ref_len,
                const jl_params *prm, const uint64_t *drm_masks, int phasing, uint32_t min_reads, int want_read_hap,
                uint32_t reps, float *ms_avg);
/* Average device time in ms of `reps` back-to-back launches of the pileup kernel alone, by HIP events on the ctx stream. */
int jl_time_pileup(jl_ctx *ctx, uint32_t reps, float *ms_avg);
/* The same over several contexts' resident windows in rotation, all on the first context's stream: no launch finds
 * its input in the Infinity Cache (4 x 150 MB > 256 MiB), as in a pipelined loop over several batches. */
int jl_time_pileup_set(jl_ctx *const *ctxs, uint32_t n_ctx, uint32_t reps, float *ms_avg);
/* The pileup of a RUN, timed where it runs: `on` puts two one-thread nodes around the pileup launch of this context's runs
 * (jl_run_async), each storing the device's constant-rate clock into pinned host memory; jl_run_pileup_ms (after jl_run_wait or
 * a fetch of the run) gives the distance between the two of the LAST run in ms — the pileup as it ran beside whatever else the
 * device was doing (another sample's latency-bound stages, another window's pileup), not alone (jl_time_pileup).  Off by
 * default: two more nodes are 2-3 us of a run. */
int jl_run_pileup_clock(jl_ctx *ctx, int on);
/* begin_ticks (may be null): the first of the two stamps, in ticks of that 100 MHz clock — one clock for all contexts of a
 * device, so the pileups of several contexts can be laid on one time line (do they overlap?  how long is the device without one?). */
int jl_run_pileup_ms(jl_ctx *ctx, float *ms, uint64_t *begin_ticks);
/* Name of the dominant kernel as rocprofv3 reports it. */
const char *jl_pileup_kernel_name(void);

/* ---------------------------------------------------------------- multi-GPU */

/* Rank 0 makes a 128-byte RCCL unique id and hands it to the other ranks out of band. */
int jl_comm_unique_id(uint8_t id[128]);
/* One communicator per (rank, device); any context on that device may use it.  Collectives run on the
 * communicator's own stream, ordered behind the producing context by an event. */
int jl_comm_create(jl_ctx *ctx, const uint8_t id[128], int rank, int world, jl_comm **out);
/* The same communicator for ranks that are THREADS OF ONE PROCESS (a host that drives several devices from one process,
 * or — RCCL refuses that — several ranks on one device): every exchange is a set of device copies between the ranks'
 * buffers (same device, or peer devices over xGMI) between two barriers of the rank threads; no RCCL communicator is
 * made.  `id`: 128 bytes that name the world (jl_comm_unique_id, or any bytes unique in the process); every rank of the
 * world calls this once.  Everything that takes a jl_comm works on it unchanged. */
int jl_comm_create_inproc(jl_ctx *ctx, const uint8_t id[128], int rank, int world, jl_comm **out);
void jl_comm_destroy(jl_comm *comm);
/* What the communicator is, for a caller's records: rccl_ranks = what ncclCommCount reports (0: the in-process form, which has
 * no RCCL communicator), rank / world as given at creation, exchange_form = how a group's bound exchange travels on it
 * (-1 not decided yet: nothing was bound; 1 the all-gather works in place in pinned host memory; 0 staged: device region +
 * heads_to_host kernel).  Any pointer may be null.  Match: SURVEY 8e (one all-gather of the variant table). */
int jl_comm_info(jl_comm *comm, int *rccl_ranks, int *rank, int *world, int *exchange_form);
/*
 * The one collective of the path: all-gather of the fixed-stride variant table over RCCL/xGMI.
 * all_rows[world*cap_rows], all_counts[world] are HOST outputs; rows of rank r start at r*cap_rows.
 */
int jl_allgather_variants(jl_ctx *ctx, jl_comm *comm, jl_variant *all_rows, uint32_t *all_counts, uint32_t cap_rows);
/*
 * Enqueue-only half for hosts that keep several batches in flight: call right after jl_run_async; the
 * exchange (6.2 KB per rank: result header + up to 128 rows) then overlaps other work and the following
 * jl_allgather_variants on the same ctx/comm only waits and unpacks.  The call itself makes no HIP call: a worker
 * thread of the communicator waits for the run's completion word and then issues the collective on the
 * communicator's stream.  The device result block is double-buffered by run parity, so a context may launch its
 * next run — and request that run's exchange — before collecting this one: at most TWO exchanges pending per
 * context (a third, and a run started while two are pending, are refused with JL_ERR_STATE);
 * jl_allgather_variants returns the OLDEST pending one.  A run that called more than 128 variants on some rank falls back
 * to the full fixed-stride table, which is not double-buffered: such an exchange must be collected before the
 * context's next run (JL_ERR_STATE otherwise, on every rank alike).
 * At most 128 exchanges in flight per communicator; every collective of a communicator is issued by its worker thread.
 */
int jl_allgather_variants_async(jl_ctx *ctx, jl_comm *comm);
/* The same for several contexts at once (the windows of a group run): their all-gathers are issued as ONE RCCL group,
 * i.e. one collective launch.  Every rank must pass the same number of contexts in the same call order.  Collect
 * each context's exchange with jl_allgather_variants as usual. */
int jl_allgather_variants_async_many(jl_ctx *const *ctxs, uint32_t n, jl_comm *comm);
/* The collecting half for several contexts in one call (e.g. the windows of one launch, one cycle later):
 * all_rows [n_ctx][world][cap_rows], all_counts [n_ctx][world]; jl_last_error of the failing context tells why. */
int jl_allgather_variants_many(jl_ctx *const *ctxs, uint32_t n_ctx, jl_comm *comm, jl_variant *all_rows,
                               uint32_t *all_counts, uint32_t cap_rows);
/*
 * The exchange of a group run as ONE device operation, carried by the run itself.  After jl_group_exchange_bind(group, comm)
 * every jl_group_run(_masked)_async of the group (at most 32 windows) also exchanges the heads of its windows' tables: the
 * run's last kernels write them — besides the result blocks — into this rank's part of a pinned host region
 * [rank][window][head], and the same call issues ONE all-gather IN PLACE in that region on the group's stream, behind
 * the run, and one event.  No gather kernel, no copy back, no worker thread, no second stream; with one rank the
 * all-gather is nothing at all.  The launching thread makes the RCCL call itself, behind whatever the communicator's
 * worker still had to issue, so every rank keeps one order of collectives as long as every rank runs the same program.
 * jl_group_exchange_collect returns the OLDEST pending exchange of the group — all_rows [n_windows][world][cap_rows],
 * all_counts [n_windows][world], as jl_allgather_variants_many — spinning on its event (JL_ERR_COMM after 60 s, and when a
 * rank's run did not reach its result block: every rank sees that rank's empty head).  Two regions by run parity: a
 * group may launch its next run before collecting this one; a third pending exchange is refused (JL_ERR_STATE).  A run
 * that fails on a rank before or at its launch still issues that rank's collective, with empty heads: the launching call
 * returns the run's own error, the collecting calls of ALL ranks report that rank (JL_ERR_COMM), nobody waits for ever.  A run
 * in which some rank called more than 128 variants falls back — on every rank alike — to the full fixed-stride table of
 * every window, which must then be collected before the group's next run.
 * Binding is a collective the first time a communicator is bound: the ranks try a 64-byte all-gather in pinned host memory
 * and agree on the outcome; where that does not work (or with JL_EXCHANGE_STAGED=1) the all-gather works in device memory
 * and a copy to the pinned region follows it (two operations).  comm = NULL unbinds; a group must be unbound or destroyed
 * before its communicator is.  In-process communicators block
 * in the launching call, as all their exchanges do.  Match: SURVEY 8e "exchange the variant table", doc/JULIET.md:94-100.
 */
int jl_group_exchange_bind(jl_group *group, jl_comm *comm);
int jl_group_exchange_collect(jl_group *group, jl_variant *all_rows, uint32_t *all_counts, uint32_t cap_rows);

/* ---------------------------------------------------------------- cross-window phasing */

/*
 * Reads that span several windows are phased on a compact matrix made of only the variant columns of ALL
 * windows: position k (ascending global column) occupies columns 3k..3k+2 of `pc`'s resident matrix.
 * `merged`: every window's called rows with GLOBAL (reference) columns, e.g. the all-gathered table.
 * Outputs: `remapped` (same rows, col = 3k) to pass to jl_phase_async(pc, remapped, ...), `pos_global[vp]`.
 * All windows must hold the same reads in the same order.  Phasing itself then runs replicated on the
 * compact matrix (3*Vp columns: a few MB even at 1e7 reads).
 */
/* The plan alone, on the host (no device, no communicator — what both functions below compute first, exposed for hosts that
 * move the columns themselves): vp_total distinct positions, pos_global[k] ascending, remapped rows (col = 3k), and
 * owner[k] = index of the window [win_begin[w], win_begin[w] + win_ncols[w]) that holds columns pos..pos+2 (-1: none). */
int jl_xwin_plan(const uint32_t *win_begin, const uint32_t *win_ncols, uint32_t n_windows, const jl_variant *merged,
                 uint32_t n_var, jl_variant *remapped, uint32_t *pos_global, int32_t *owner, uint32_t *vp_total);
/* every window resident on this device: device-to-device copies */
int jl_xwin_assemble_local(jl_ctx *pc, jl_ctx *const *windows, uint32_t n_windows, const jl_variant *merged,
                           uint32_t n_var, jl_variant *remapped, uint32_t *pos_global, uint32_t *vp_total);
/* one window per rank: each position's owner broadcasts its 3 columns over RCCL (second exchange of the run) */
int jl_xwin_assemble_rccl(jl_ctx *pc, jl_ctx *window, jl_comm *comm, const uint32_t *win_begin,
                          const uint32_t *win_ncols, const jl_variant *merged, uint32_t n_var, jl_variant *remapped,
                          uint32_t *pos_global, uint32_t *vp_total);

/*
 * Cross-window phasing with the READS sharded (SURVEY §8e option A; docs/SPEC.md §8).  Rank s phases reads
 * [slice_begin[s], slice_begin[s+1]) only: its compact matrix holds that slice of every variant position's three columns,
 * so the second exchange moves 1/world of the bytes and every rank groups 1/world of the reads.  Sequence per rank:
 *   jl_xwin_assemble_slice_*  ->  jl_phase_groups_async  ->  jl_phase_groups_fetch  (this slice's groups: pattern + count)
 *   all-gather of the group tables (KB) and merge on the host: counts of equal patterns add up; the merged groups with
 *     >= min_reads reads are the haplotypes, ordered as SPEC §8 says; hit and co-occurrence follow from patterns and counts
 *   jl_phase_regroup          (each exported group's haplotype id back to the device: per-read ids of the slice)
 * Slices start on multiples of 256 reads.  The host side of the merge is minorseq_amd/sharding.py (merge_groups,
 * select_haplotypes); tests/test_gpu_sharded_phase.py checks the whole sequence against the unsharded run.
 */
int jl_xwin_assemble_slice_local(jl_ctx *pc, jl_ctx *const *windows, uint32_t n_windows, const jl_variant *merged,
                                 uint32_t n_var, uint64_t read_begin, uint64_t n_slice, jl_variant *remapped,
                                 uint32_t *pos_global, uint32_t *vp_total);
/* one window per rank: the owner of a position sends rank s its slice (ncclSend / ncclRecv in one group);
 * slice_begin has world + 1 entries, the same on every rank */
int jl_xwin_assemble_slice_rccl(jl_ctx *pc, jl_ctx *window, jl_comm *comm, const uint32_t *win_begin,
                                const uint32_t *win_ncols, const jl_variant *merged, uint32_t n_var,
                                const uint64_t *slice_begin, jl_variant *remapped, uint32_t *pos_global, uint32_t *vp_total);
/* keys + grouping of the resident matrix, the groups written out instead of ranked (variants as in jl_phase_async) */
int jl_phase_groups_async(jl_ctx *ctx, const jl_variant *variants, uint32_t n_var);
/* group q: counts[q] reads with codon code patterns[q * pattern_stride + p] at variant position p (pos_cols[p]);
 * *partial: this matrix's damaged reads and marginals, its clean reads under insufficient_reads.  Pointers except
 * n_groups / n_positions may be NULL. */
int jl_phase_groups_fetch(jl_ctx *ctx, uint8_t *patterns, uint32_t pattern_stride, uint32_t *counts, uint32_t cap_groups,
                          uint32_t *n_groups, uint32_t *n_positions, uint32_t *pos_cols, uint32_t cap_var,
                          jl_phase_summary *partial);
/* hap_of_group[q] = haplotype id of exported group q after the merge (JL_HAP_INSUFFICIENT: not reported);
 * read_hap (optional, [n_reads of this matrix]) receives the per-read ids */
int jl_phase_regroup(jl_ctx *ctx, const uint16_t *hap_of_group, uint32_t n_groups, uint32_t n_haplotypes, uint16_t *read_hap);

/* ---------------------------------------------------------------- the merge of the sharded path: host only
 *
 * No device and no communicator: these run anywhere the library loads (the gloo tests call them on CPU ranks).
 * Behaviour: the >= 10-read rule applies to the MERGED count of a pattern (J:253-254); order, ids, haplotype_hit and the
 * read categories as docs/SPEC.md §8 (J:192-211, 372-381). */

/* Per-window tables (window-relative `col`) -> one table with GLOBAL columns in (gene, codon_pos, codon) order.
 * tables[t] has counts[t] rows and belongs to the window starting at reference column win_begin[t].
 * JL_ERR_OVERFLOW if the rows do not fit `cap` (*n_merged then holds the number needed). */
int jl_merge_tables(const jl_variant *const *tables, const uint32_t *counts, const uint32_t *win_begin, uint32_t n_tables,
                    jl_variant *merged, uint32_t cap, uint32_t *n_merged);
/* Exported group tables (jl_phase_groups_fetch: patterns[t][q * pattern_stride[t] + p], counts[t][q], n_groups[t] groups
 * of vp positions) -> the distinct patterns in ascending order (codon codes compared position by position), the summed
 * counts, and index[t][q] = merged row of group q of table t (index or index[t] may be NULL). */
int jl_merge_groups(const uint8_t *const *patterns, const uint32_t *pattern_stride, const uint32_t *const *counts,
                    const uint32_t *n_groups, uint32_t n_tables, uint32_t vp, uint8_t *merged_patterns /* [cap][vp] */,
                    uint64_t *merged_counts, uint32_t cap, uint32_t *n_merged, uint32_t *const *index);
/* docs/SPEC.md §8 on merged groups: reported = count >= min_reads, ordered by (count desc, pattern asc), at most
 * JL_MAX_HAPLOTYPES.  `variants`: the table with col = 3 * position index (jl_xwin_plan's `remapped`) or any table whose
 * `col` values appear in pos_cols[vp].  partials[n_partials]: each slice's damaged reads and marginals.
 * Outputs (any may be NULL): summary; hap_count[H]; hap_pattern[H][vp]; hit[n_var][hit_stride] (J:207-209);
 * cooc[n_var][n_var]; hap_of_group[n_groups] (JL_HAP_INSUFFICIENT where not reported). */
int jl_select_haplotypes(const uint8_t *patterns, const uint64_t *counts, uint32_t n_groups, uint32_t vp,
                         const jl_variant *variants, uint32_t n_var, const uint32_t *pos_cols, uint32_t min_reads,
                         const jl_phase_summary *partials, uint32_t n_partials, jl_phase_summary *summary,
                         uint32_t *hap_count, uint8_t *hap_pattern, uint8_t *hit, uint32_t hit_stride, uint32_t *cooc,
                         uint16_t *hap_of_group);

/* The schedule of the column-slice exchange as data (what jl_xwin_phase_sharded / jl_xwin_assemble_slice_rccl issue):
 * positions are ascending global columns and windows ascending column ranges, so the positions a rank owns are ONE run
 * k_begin .. k_begin + k_count, and a rank sends every peer ONE packed message: slice s of the 9 * k_count plane
 * rows of the owned columns (three columns x three planes per position), each row padded with 'not covered' (code 6: 0x00
 * in plane 0, 0xFF in planes 1 and 2) to dst_stride = jl_plane_stride(reads of slice s) — exactly the bytes of columns
 * 3 * k_begin .. of the receiver's compact matrix, where the matching receive lands them.
 * win_rank[w] = rank that holds window w (non-decreasing).  ops of rank `rank`, in issue order: its own slice (a device
 * copy), then per peer in ascending order the send and the receive.  Every send has exactly one matching receive
 * (same byte count) in the peer's list.  JL_ERR_ARG if a variant column lies in no window. */
enum { JL_XWIN_OP_LOCAL = 0, JL_XWIN_OP_SEND = 1, JL_XWIN_OP_RECV = 2 };
typedef struct {
    int32_t op, peer;            /* peer: the other rank (own rank for JL_XWIN_OP_LOCAL) */
    uint32_t k_begin, k_count;   /* positions whose columns travel */
    uint64_t read_begin, n_reads;/* the slice of the reads (the RECEIVER's slice) */
    uint64_t dst_stride;         /* bytes per plane row in the message = the receiver's plane stride */
    uint64_t bytes;              /* 9 * k_count * dst_stride */
    uint64_t dst_offset;         /* where the message lands in the receiver's compact matrix: 9 * k_begin * dst_stride */
} jl_xwin_op;
int jl_xwin_slice_plan(const uint32_t *win_begin, const uint32_t *win_ncols, const int32_t *win_rank, uint32_t n_windows,
                       const jl_variant *merged, uint32_t n_var, const uint64_t *slice_begin, int32_t world, int32_t rank,
                       jl_xwin_op *ops, uint32_t cap_ops, uint32_t *n_ops);

/* ---------------------------------------------------------------- cross-window phasing, the whole sequence
 *
 * A session = this rank's windows of ONE reference whose reads span every window (BASELINE.json configs[3]/[4]), the
 * communicator (NULL when world = 1: every window is on this device), the column layout of ALL windows and the read
 * slices.  windows[0 .. n_local) are this rank's windows in ascending column order, i.e. the windows w with
 * win_rank[w] == rank.  Slices start on multiples of 256 reads; slice_begin has world + 1 entries.
 *
 * jl_xwin_phase_sharded, called by every rank after its windows' call stage (jl_run_async / jl_group_run_async with
 * phasing off, or jl_call_async), runs the rest of the path with the reads sharded (option A):
 *   all-gather of the variant tables (RCCL, one collective; none at world = 1)  ->  merge + plan on the host  ->
 *   one packed send per peer of the variant columns' slices (RCCL; a device copy for the own slice)  ->
 *   keys + grouping of the slice on the device, the groups exported  ->  all-gather of the group tables (RCCL) ->
 *   merge + selection on the host (jl_merge_groups, jl_select_haplotypes)  ->  per-read ids of the slice on the device.
 * The result's pointers are into the session and stay valid until the next call; the per-read ids stay in HBM — their
 * launch is enqueued when the call returns, not awaited — until jl_xwin_read_hap_fetch (which waits for it).  While the call runs the communicator must have no asynchronous exchange pending
 * (JL_ERR_STATE otherwise): the session issues its collectives from the calling thread. */
typedef struct jl_xwin jl_xwin;
typedef struct {
    uint32_t n_variants;          /* rows of `merged` */
    uint32_t n_positions;         /* Vp */
    uint32_t n_haplotypes;        /* H */
    uint32_t n_groups;            /* distinct patterns over all slices */
    jl_phase_summary summary;     /* read categories over ALL reads (J:372-381) */
    const jl_variant *merged;     /* every window's rows, global columns, (gene, codon_pos, codon) order */
    const uint32_t *pos_global;   /* [Vp] ascending global columns */
    const uint32_t *hap_count;    /* [H] */
    const uint8_t *hap_pattern;   /* [H][Vp] */
    const uint8_t *hit;           /* [n_variants][H] (haplotype_hit, J:207-209) */
    const uint32_t *cooc;         /* [n_variants][n_variants] */
    uint64_t slice_begin, slice_reads;  /* this rank's reads */
    uint32_t read_hap_bits;       /* width of the per-read ids in HBM: 4, 8 or 16 */
    uint32_t reserved;
} jl_xwin_result;
int jl_xwin_create(jl_ctx *const *windows, uint32_t n_local, jl_comm *comm, const uint32_t *win_begin,
                   const uint32_t *win_ncols, const int32_t *win_rank, uint32_t n_windows, const uint64_t *slice_begin,
                   jl_xwin **out);
void jl_xwin_destroy(jl_xwin *x);
const char *jl_xwin_last_error(const jl_xwin *x);
int jl_xwin_phase_sharded(jl_xwin *x, uint32_t min_reads, jl_xwin_result *out);
/* Host time of the last jl_xwin_phase_sharded by stage, in microseconds (out[JL_XWIN_STAGES]): 0 the windows' tables on
 * the host (waits for their call stage), 1 all-gather + merge of the tables, 2 plan, 3 pack + exchange + grouping
 * enqueued, 4 waiting for the groups (and their all-gather), 5 merge + selection, 6 per-read ids. */
enum { JL_XWIN_STAGES = 8 };
int jl_xwin_stage_us(const jl_xwin *x, float *out);
/* 16-bit ids of this rank's slice (read_hap[slice_reads]) of the last jl_xwin_phase_sharded. */
int jl_xwin_read_hap_fetch(jl_xwin *x, uint16_t *read_hap);
/* The collective on its own, for hosts that drive the stages themselves: fixed-stride all-gather over RCCL of the
 * groups jl_phase_groups_async exported on `ctx` (cap_groups rows of pattern_stride bytes per rank, plus counts, the
 * partial summary and the number of positions).  Host outputs: patterns[world][cap_groups][pattern_stride],
 * counts[world][cap_groups], n_groups[world], partials[world].  JL_ERR_OVERFLOW, on every rank alike, when a rank
 * exported more than cap_groups groups. */
int jl_allgather_groups(jl_ctx *ctx, jl_comm *comm, uint32_t cap_groups, uint32_t pattern_stride, uint8_t *patterns,
                        uint32_t *counts, uint32_t *n_groups, jl_phase_summary *partials, uint32_t *n_positions);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif
#endif /* JULIET_HIP_H */
