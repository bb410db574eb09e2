// capi.hip — the C ABI of include/juliet_hip.h: context, residency, planning, launches, copies.
// No compute happens on the host here and there is no CPU fallback: every entry point needs a gfx950 device.
#include <stdarg.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <new>

#include "jl_internal.h"

// 1: a small window's last launch stores the completion word itself (every workgroup that wrote host memory releases
// it at system scope first); 0: always a one-thread node of its own behind the last stage.  Measured equal on one
// window alone (74.1 vs 74.4 us) and no faster in the pipelined loop: the node of its own is the default.
#ifndef JL_SIGNAL_IN_KERNEL
#define JL_SIGNAL_IN_KERNEL 0
#endif

static thread_local std::string g_create_error;

// (The library does not touch the process environment.  A host that keeps several launches in flight beside an exchange
// wants GPU_MAX_HW_QUEUES=8 — the HIP runtime multiplexes a process's streams onto 4 hardware queues otherwise and reads the
// setting at the process's first HIP call: the juliet front end and bench.py set it before that call, INTEGRATION.md says so.)

int jl_fail(jl_ctx *ctx, int status, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    else g_create_error = buf;
    return status;
}

template <typename T>
static int regrow(jl_ctx *ctx, T **p, size_t n)
{
    if (*p) hipFree(*p);
    *p = nullptr;
    ctx->alloc_version++;  // pointers captured in a graph are stale now
    JL_HIP(ctx, hipMalloc(p, n * sizeof(T)));
    return JL_OK;
}

extern "C" {

int jl_abi_version(void) { return JL_ABI_VERSION; }

const char *jl_strerror(int status)
{
    switch (status) {
    case JL_OK: return "ok";
    case JL_ERR_ARG: return "bad argument";
    case JL_ERR_DEVICE: return "no usable gfx950 device / HIP error";
    case JL_ERR_MEMORY: return "device allocation failed";
    case JL_ERR_STATE: return "call order violated";
    case JL_ERR_OVERFLOW: return "output capacity exceeded";
    case JL_ERR_COMM: return "RCCL failure";
    default: return "unknown status";
    }
}

int jl_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int d = 0; d < n; ++d) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, d) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}

const char *jl_last_error(const jl_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

#ifdef JL_TUNING
// JL_CTX_TIMES=1: where jl_ctx_create spends its time (tools_tuning/ctx_create_cost.py)
#define JL_CTX_MARK(what)                                                                                              \
    do {                                                                                                               \
        if (ctx_times) {                                                                                               \
            const auto now_ = std::chrono::steady_clock::now();                                                        \
            fprintf(stderr, "  jl_ctx_create: %-28s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(now_ - mark_).count()); \
            mark_ = now_;                                                                                              \
        }                                                                                                              \
    } while (0)
#else
#define JL_CTX_MARK(what) do { } while (0)
#endif

int jl_ctx_create(int device, void *stream, jl_ctx **out)
{
    if (!out) return JL_ERR_ARG;
    *out = nullptr;
#ifdef JL_TUNING
    const bool ctx_times = getenv("JL_CTX_TIMES") != nullptr;
    auto mark_ = std::chrono::steady_clock::now();
#endif
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return jl_fail(nullptr, JL_ERR_DEVICE, "no HIP device visible; this library has no CPU fallback");
    if (device < 0 || device >= n) return jl_fail(nullptr, JL_ERR_ARG, "device %d out of range (%d visible)", device, n);
    JL_CTX_MARK("hipGetDeviceCount");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return jl_fail(nullptr, JL_ERR_DEVICE, "hipGetDeviceProperties failed");
    JL_CTX_MARK("hipGetDeviceProperties");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return jl_fail(nullptr, JL_ERR_DEVICE, "device %d is %s; kernels are built for gfx950 only", device, prop.gcnArchName);
    jl_ctx *ctx = new (std::nothrow) jl_ctx();
    if (!ctx) return JL_ERR_MEMORY;
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess) { delete ctx; return jl_fail(nullptr, JL_ERR_DEVICE, "hipSetDevice failed"); }
    JL_CTX_MARK("hipSetDevice");
    if (stream) {
        ctx->stream = (hipStream_t)stream;
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
            delete ctx;
            return jl_fail(nullptr, JL_ERR_DEVICE, "hipStreamCreate failed");
        }
        ctx->own_stream = true;
    }
    JL_CTX_MARK("stream");
    hipEventCreate(&ctx->ev0);
    hipEventCreate(&ctx->ev1);
    JL_CTX_MARK("two events");

    bool ok = hipMalloc(&ctx->d_variants, sizeof(jl_variant) * JL_VARIANT_CAP) == hipSuccess &&
              hipMalloc(&ctx->d_nvar, 2 * sizeof(uint32_t)) == hipSuccess &&
              hipMalloc(&ctx->d_meta, sizeof(jl_phase_meta)) == hipSuccess &&
              hipMalloc(&ctx->d_vpcols, sizeof(uint32_t) * JL_VARIANT_CAP) == hipSuccess &&
              hipMalloc(&ctx->d_hap_count, sizeof(uint32_t) * JL_MAX_HAPLOTYPES) == hipSuccess &&
              hipMalloc(&ctx->d_hap_pattern, (size_t)JL_MAX_HAPLOTYPES * JL_VARIANT_CAP) == hipSuccess &&
              hipMalloc(&ctx->d_hit, (size_t)JL_VARIANT_CAP * JL_MAX_HAPLOTYPES) == hipSuccess &&
              hipMalloc(&ctx->d_cooc, sizeof(uint32_t) * ctx->cooc_cap * ctx->cooc_cap) == hipSuccess &&
              hipMalloc(&ctx->d_pack, 2 * sizeof(jl_pack)) == hipSuccess &&
              hipMalloc(&ctx->d_sync, 16 * sizeof(uint32_t)) == hipSuccess;
    JL_CTX_MARK("ten hipMalloc");
    ok = ok && hipHostMalloc(&ctx->h_pack, sizeof(jl_pack), hipHostMallocDefault) == hipSuccess &&
              hipHostMalloc((void **)&ctx->h_seq, 64, hipHostMallocDefault) == hipSuccess &&
              hipHostMalloc(&ctx->h_scratch, (size_t)1 << 20, hipHostMallocDefault) == hipSuccess;
    JL_CTX_MARK("three hipHostMalloc");
    if (ok) ctx->h_scratch_cap = (size_t)1 << 20;
    if (!ok) { jl_ctx_destroy(ctx); return jl_fail(nullptr, JL_ERR_MEMORY, "context allocation failed"); }
    hipMemsetAsync(ctx->d_nvar, 0, 2 * sizeof(uint32_t), ctx->stream);
    hipMemsetAsync(ctx->d_meta, 0, sizeof(jl_phase_meta), ctx->stream);
    hipMemsetAsync(ctx->d_sync, 0, 16 * sizeof(uint32_t), ctx->stream);
    for (int k = 0; k < 16; ++k) ctx->h_seq[k] = 0;
    JL_CTX_MARK("three hipMemsetAsync");
    *out = ctx;
    return JL_OK;
}

static void free_msa(jl_ctx *ctx)
{
    ctx->alloc_version++;
    if (ctx->own_msa && ctx->d_msa) hipFree(ctx->d_msa);
    ctx->d_msa = nullptr;
    ctx->own_msa = false;
    ctx->msa_capacity = 0;
}

static void records_drop(jl_ctx *ctx);

void *jl_ctx_stream(const jl_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

void jl_ctx_destroy(jl_ctx *ctx)
{
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    if (ctx->graph_exec) hipGraphExecDestroy(ctx->graph_exec);
    if (ctx->graph) hipGraphDestroy(ctx->graph);
    if (ctx->h_pack) hipHostFree(ctx->h_pack);
    if (ctx->h_seq) hipHostFree((void *)ctx->h_seq);
    if (ctx->h_read_hap) hipHostFree(ctx->h_read_hap);
    if (ctx->h_scratch) hipHostFree(ctx->h_scratch);
    free_msa(ctx);
    records_drop(ctx);
    void *ptrs[] = {ctx->d_pos_gene, ctx->d_pos_codon, ctx->d_pos_col, ctx->d_pos_refcfg, ctx->d_col_head, ctx->d_pos_next, ctx->d_guess, ctx->d_chunks,
                    ctx->d_counts, ctx->d_called, ctx->d_staged, ctx->d_drm, ctx->d_variants, ctx->d_nvar, ctx->d_meta, ctx->d_vpcols,
                    ctx->d_col2pos, ctx->d_varcol, ctx->d_keys, ctx->d_flagw, ctx->d_read_slot, ctx->d_read_hap,
                    ctx->d_slot_rep, ctx->d_slot_count, ctx->d_slot_key, ctx->d_slot_hap, ctx->d_occupied, ctx->d_hap_count,
                    ctx->d_hap_pattern, ctx->d_hit, ctx->d_cooc, ctx->d_pack, ctx->d_sync, ctx->d_timeline,
                    ctx->d_ins_len, ctx->d_ins_base, ctx->d_ing_runs, ctx->d_ing_nruns, ctx->d_ing_desc, ctx->d_ing_count, ctx->d_ing_slow,
                    ctx->d_exp_count,
                    ctx->d_exp_pattern, ctx->d_exp_hap, ctx->d_blockcat, ctx->d_slot_key_a, ctx->d_slot_key_b, ctx->d_occ_a, ctx->d_occ_b};
    for (void *p : ptrs)
        if (p) hipFree(p);
    if (ctx->ev0) hipEventDestroy(ctx->ev0);
    if (ctx->ev1) hipEventDestroy(ctx->ev1);

    if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

int jl_sync(jl_ctx *ctx)
{
    if (!ctx) return JL_ERR_ARG;
    JL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // the last run may have been enqueued on a group's stream
    if (ctx->run_stream && ctx->run_stream != ctx->stream) JL_HIP(ctx, hipStreamSynchronize(ctx->run_stream));
    return jl_ingest_verdict(ctx);   // (an ingest that was only enqueued: what it found wrong with the records)
}

/* ---------------------------------------------------------------- MSA residency */

uint64_t jl_col_stride(uint64_t n_reads) { return ((n_reads + 1) / 2 + 127) / 128 * 128; }
uint64_t jl_plane_stride(uint64_t n_reads) { return (n_reads + 1023) / 1024 * 128; }

static int set_shape(jl_ctx *ctx, uint64_t n_reads, uint32_t n_cols, uint64_t plane_stride, uint32_t win_begin)
{
    if (n_reads == 0 || n_cols == 0) return jl_fail(ctx, JL_ERR_ARG, "empty matrix (%llu reads x %u columns)", (unsigned long long)n_reads, n_cols);
    if (n_reads > 0x7FFFFFFFull) return jl_fail(ctx, JL_ERR_ARG, "more than 2^31-1 reads per context");
    if (plane_stride % 16 != 0 || plane_stride < (n_reads + 7) / 8)
        return jl_fail(ctx, JL_ERR_ARG, "plane_stride %llu must be a multiple of 16 and >= ceil(n_reads/8)", (unsigned long long)plane_stride);
    if (n_reads != ctx->n_reads || n_cols != ctx->n_cols || plane_stride != ctx->plane_stride || win_begin != ctx->win_begin)
        ctx->plan_valid = false;
    ctx->n_reads = n_reads;
    ctx->n_cols = n_cols;
    ctx->plane_stride = plane_stride;
    ctx->col_stride = plane_stride * 4u;   // the same stride in 8-reads-per-dword units (jl_internal.h)
    ctx->win_begin = win_begin;
    ctx->alloc_version++;
    ctx->pack_valid = false;
    ctx->pileup_done = ctx->call_done = ctx->phase_done = false;
    ctx->ins_valid = false;
    return JL_OK;
}

static int reserve_msa(jl_ctx *ctx, size_t bytes)
{
    if (ctx->own_msa && ctx->msa_capacity >= bytes) return JL_OK;
    free_msa(ctx);
    JL_HIP(ctx, hipMalloc(&ctx->d_msa, bytes));
    ctx->own_msa = true;
    ctx->msa_capacity = bytes;
    return JL_OK;
}

int jl_msa_alloc(jl_ctx *ctx, uint64_t n_reads, uint32_t n_cols, uint32_t win_begin)
{
    return jl_msa_alloc_strided(ctx, n_reads, n_cols, jl_plane_stride(n_reads), win_begin);
}
}  // extern "C"

// internal: a resident matrix with the caller's plane stride (cross-window phasing keeps the stride of the windows
// whose columns it copies)
int jl_msa_alloc_strided(jl_ctx *ctx, uint64_t n_reads, uint32_t n_cols, uint64_t plane_stride, uint32_t win_begin)
{
    if (!ctx) return JL_ERR_ARG;
    JL_HIP(ctx, hipSetDevice(ctx->device));
    int rc = set_shape(ctx, n_reads, n_cols, plane_stride, win_begin);
    if (rc) return rc;
    return reserve_msa(ctx, (size_t)ctx->plane_stride * 3u * n_cols);
}

extern "C" {

// The interchange format (column-packed nibbles) goes through a bounded staging buffer: a run of columns is copied to the
// device and ONE kernel validates its codes (0..6, SPEC §1) and writes its planes — no resident copy of the nibbles.
int jl_msa_upload(jl_ctx *ctx, const uint8_t *colpacked, uint64_t n_reads, uint32_t n_cols, uint64_t col_stride,
                  uint32_t win_begin)
{
    if (!ctx || !colpacked) return JL_ERR_ARG;
    if (col_stride % 128 != 0 || col_stride < (n_reads + 1) / 2)
        return jl_fail(ctx, JL_ERR_ARG, "col_stride %llu must be a multiple of 128 and >= ceil(n_reads/2)", (unsigned long long)col_stride);
    int rc = jl_msa_alloc(ctx, n_reads, n_cols, win_begin);
    if (rc) return rc;
    const uint32_t per = (uint32_t)std::min<uint64_t>(n_cols, std::max<uint64_t>(1, ((uint64_t)64 << 20) / col_stride));
    uint8_t *d_stage = nullptr;
    JL_HIP(ctx, hipMalloc(&d_stage, (size_t)per * col_stride));
    uint32_t bad = 0;
    hipError_t e = hipMemsetAsync(ctx->d_nvar + 1, 0, 4, ctx->stream);
    for (uint32_t c0 = 0; c0 < n_cols && e == hipSuccess; c0 += per) {
        const uint32_t n = std::min(per, n_cols - c0);
        e = hipMemcpyAsync(d_stage, colpacked + (size_t)c0 * col_stride, (size_t)n * col_stride, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) break;
        jl_launch_nibbles_to_planes(ctx, d_stage, col_stride, c0, n, ctx->d_nvar + 1);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, ctx->d_nvar + 1, 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hipFree(d_stage);
    if (e != hipSuccess) { free_msa(ctx); return jl_fail(ctx, JL_ERR_DEVICE, "upload: %s", hipGetErrorString(e)); }
    if (bad) {
        free_msa(ctx);
        return jl_fail(ctx, JL_ERR_ARG, "matrix holds a symbol code outside 0..6");
    }
    return JL_OK;
}

int jl_msa_adopt(jl_ctx *ctx, void *d_planes, uint64_t n_reads, uint32_t n_cols, uint64_t plane_stride, uint32_t win_begin)
{
    if (!ctx || !d_planes) return JL_ERR_ARG;
    if (((uintptr_t)d_planes & 15u) != 0) return jl_fail(ctx, JL_ERR_ARG, "adopted matrix must be 16-byte aligned");
    int rc = set_shape(ctx, n_reads, n_cols, plane_stride, win_begin);
    if (rc) return rc;
    free_msa(ctx);
    ctx->d_msa = (uint8_t *)d_planes;
    return JL_OK;
}

int jl_msa_pack_rows(jl_ctx *ctx, const uint8_t *rows, uint64_t n_reads, uint32_t n_cols, uint32_t win_begin)
{
    if (!ctx || !rows) return JL_ERR_ARG;
    int rc = jl_msa_alloc(ctx, n_reads, n_cols, win_begin);
    if (rc) return rc;
    uint8_t *d_rows = nullptr;
    JL_HIP(ctx, hipMalloc(&d_rows, (size_t)n_reads * n_cols));
    hipError_t e = hipMemcpyAsync(d_rows, rows, (size_t)n_reads * n_cols, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        jl_launch_pack_rows(ctx, d_rows);
        e = hipStreamSynchronize(ctx->stream);
    }
    hipFree(d_rows);
    if (e != hipSuccess) return jl_fail(ctx, JL_ERR_DEVICE, "pack_rows: %s", hipGetErrorString(e));
    return JL_OK;
}

// Aligned records straight to the resident layout: cigar expansion, QV masking and the transpose all run
// on the device.  Arrays are what a BAM decoder holds: per read its leftmost position, its
// cigar words (len << 4 | op), its 4-bit packed bases exactly as stored in BAM, optionally its qualities.
// Streamed form: jl_records_begin / jl_records_append (any number of chunks, e.g. one per inflated BGZF batch,
// so the upload hides under the decode of the next chunk) / jl_records_finish (alloc + kernels).
static void records_drop(jl_ctx *ctx)
{
    jl_records &r = ctx->rec;
    void *tmp[] = {r.d_seq, r.d_cig, r.d_co, r.d_so, r.d_pos, r.d_qual, r.d_qo};
    for (void *p : tmp)
        if (p) hipFree(p);
    r = jl_records();
}

// room for `need` elements of `elem` bytes (+pad bytes behind them); what is already there moves along
static hipError_t records_room_bytes(jl_ctx *ctx, void **d, size_t *cap, size_t elem, size_t used, size_t need, size_t pad_bytes)
{
    if (*d && need <= *cap) return hipSuccess;
    const size_t ncap = std::max<size_t>({need, *cap + *cap / 2, (size_t)1024});
    void *nd = nullptr;
    hipError_t e = hipMalloc(&nd, ncap * elem + pad_bytes);
    if (e != hipSuccess) return e;
    if (*d && used) e = hipMemcpyAsync(nd, *d, used * elem, hipMemcpyDeviceToDevice, ctx->stream);
    if (*d) {
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        hipFree(*d);
    }
    *d = nd;
    *cap = ncap;
    return e;
}
#define records_room(ctx, d, cap, used, need, pad) records_room_bytes(ctx, (void **)&(d), &(cap), sizeof(*(d)), used, need, pad)

int jl_records_begin(jl_ctx *ctx, uint64_t reads_hint, uint64_t cigar_words_hint, uint64_t seq_bytes_hint, uint64_t qual_bytes_hint)
{
    if (!ctx) return JL_ERR_ARG;
    JL_HIP(ctx, hipSetDevice(ctx->device));
    records_drop(ctx);
    jl_records &r = ctx->rec;
    r.open = true;
    hipError_t e = records_room(ctx, r.d_pos, r.cap_pos, 0, (size_t)reads_hint, 0);
    if (e == hipSuccess) e = records_room(ctx, r.d_co, r.cap_co, 0, (size_t)reads_hint + 1, 0);
    if (e == hipSuccess) e = records_room(ctx, r.d_so, r.cap_so, 0, (size_t)reads_hint + 1, 0);
    if (e == hipSuccess) e = records_room(ctx, r.d_cig, r.cap_cig, 0, (size_t)cigar_words_hint, 64);
    if (e == hipSuccess) e = records_room(ctx, r.d_seq, r.cap_seq, 0, (size_t)seq_bytes_hint, 64);
    // (the qualities begin 16 bytes into their array: the ingest reads a piece's 32 qualities from up to six bytes before a read's first)
    r.n_qual = 16;
    if (e == hipSuccess && qual_bytes_hint) e = records_room(ctx, r.d_qual, r.cap_qual, 0, (size_t)qual_bytes_hint + 16, 64);
    if (e == hipSuccess && qual_bytes_hint) e = records_room(ctx, r.d_qo, r.cap_qo, 0, (size_t)reads_hint + 1, 0);
    if (e != hipSuccess) {
        records_drop(ctx);
        return jl_fail(ctx, e == hipErrorOutOfMemory ? JL_ERR_MEMORY : JL_ERR_DEVICE, "records: %s", hipGetErrorString(e));
    }
    return JL_OK;
}

int jl_records_append(jl_ctx *ctx, uint64_t n_reads, const int32_t *pos, const uint32_t *cigar, const uint64_t *cig_off,
                      const uint8_t *seq4, const uint64_t *seq_off, const uint8_t *qual, const uint64_t *qual_off)
{
    if (!ctx || !pos || !cigar || !cig_off || !seq4 || !seq_off || (qual && !qual_off)) return JL_ERR_ARG;
    jl_records &R = ctx->rec;
    if (!R.open) return jl_fail(ctx, JL_ERR_STATE, "jl_records_append before jl_records_begin");
    if (R.n_reads && (qual != nullptr) != R.have_qual) {
        records_drop(ctx);
        return jl_fail(ctx, JL_ERR_ARG, "records: either every chunk carries qualities or none does");
    }
    if (!n_reads) return JL_OK;
    const uint64_t first = R.n_reads;
    // a chunk that fails validation ends the stream (jl_records_begin starts over).  Here: the offsets, which the uploads
    // below follow; what the cigars say — an 'M', more bases than the record holds — is checked where they are walked, on
    // the device (cigar_runs_kernel), and reported by the build (jl_records_finish / jl_records_window): the loop over
    // twelve million cigar words was most of an append on the host.
    auto bad = [&](uint64_t r, const char *what, uint64_t v) {
        records_drop(ctx);
        return jl_fail(ctx, JL_ERR_ARG, what, (unsigned long long)(first + r), (unsigned long long)v);
    };
    // ... and whether a read needs the ingest's launch for long reads: only a read with more ops than entries fit can, and a CCS
    // sample has few of those — their cigars are looked at here, a word per read of the chunk at most (then: "maybe")
    const uint64_t short_ops = jl_ingest_short_ops();
    uint64_t looked = 0;
    for (uint64_t r = 0; r < n_reads; ++r) {
        if (cig_off[r + 1] < cig_off[r] || seq_off[r + 1] < seq_off[r] || (qual && qual_off[r + 1] < qual_off[r]))
            return bad(r, "record %llu: offsets must not decrease", 0);
        const uint64_t n_ops = cig_off[r + 1] - cig_off[r];
        if (n_ops > short_ops && !R.maybe_long) {
            looked += n_ops;
            R.maybe_long = looked > n_reads + 4096u || jl_ingest_read_is_long(cigar + cig_off[r], n_ops);
        }
    }
    JL_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    // the chunk's arrays start wherever its offsets say; on the device everything is one run of arrays
    const uint64_t c0 = cig_off[0], s0 = seq_off[0], q0 = qual ? qual_off[0] : 0;
    const size_t n_cig = (size_t)(cig_off[n_reads] - c0), n_seq = (size_t)(seq_off[n_reads] - s0),
                 n_q = qual ? (size_t)(qual_off[n_reads] - q0) : 0;
    const size_t nr = (size_t)R.n_reads;
    hipError_t e = records_room(ctx, R.d_pos, R.cap_pos, nr, nr + n_reads, 0);
    if (e == hipSuccess) e = records_room(ctx, R.d_co, R.cap_co, nr + 1, nr + n_reads + 1, 0);
    if (e == hipSuccess) e = records_room(ctx, R.d_so, R.cap_so, nr + 1, nr + n_reads + 1, 0);
    if (e == hipSuccess) e = records_room(ctx, R.d_cig, R.cap_cig, (size_t)R.n_cig, (size_t)R.n_cig + n_cig, 64);
    // the kernel reads the bases in aligned 32-byte pieces: padding behind them
    if (e == hipSuccess) e = records_room(ctx, R.d_seq, R.cap_seq, (size_t)R.n_seq, (size_t)R.n_seq + n_seq, 64);
    if (e == hipSuccess && qual) e = records_room(ctx, R.d_qual, R.cap_qual, (size_t)R.n_qual, (size_t)R.n_qual + n_q, 64);
    if (e == hipSuccess && qual) e = records_room(ctx, R.d_qo, R.cap_qo, nr + 1, nr + n_reads + 1, 0);
    std::vector<uint64_t> off((size_t)(n_reads + 1) * (qual ? 3 : 2));
    uint64_t *co = off.data(), *so = co + n_reads + 1, *qo = so + n_reads + 1;
    for (uint64_t r = 0; r <= n_reads; ++r) {
        co[r] = cig_off[r] - c0 + R.n_cig;
        so[r] = seq_off[r] - s0 + R.n_seq;
        if (qual) qo[r] = qual_off[r] - q0 + R.n_qual;
    }
    const size_t off_bytes = (size_t)(n_reads + 1) * 8;
    if (e == hipSuccess && n_seq) e = hipMemcpyAsync(R.d_seq + R.n_seq, seq4 + s0, n_seq, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && n_cig) e = hipMemcpyAsync(R.d_cig + R.n_cig, cigar + c0, n_cig * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(R.d_co + nr, co, off_bytes, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(R.d_so + nr, so, off_bytes, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(R.d_pos + nr, pos, (size_t)n_reads * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && qual && n_q) e = hipMemcpyAsync(R.d_qual + R.n_qual, qual + q0, n_q, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && qual) e = hipMemcpyAsync(R.d_qo + nr, qo, off_bytes, hipMemcpyHostToDevice, st);
    // the caller may reuse its chunk buffers (and `off` goes away) as soon as this returns
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        records_drop(ctx);
        return jl_fail(ctx, e == hipErrorOutOfMemory ? JL_ERR_MEMORY : JL_ERR_DEVICE, "records: %s", hipGetErrorString(e));
    }
    R.have_qual = qual != nullptr;
    R.n_reads += n_reads;
    R.n_cig += n_cig;
    R.n_seq += n_seq;
    R.n_qual += n_q;
    return JL_OK;
}

// room for `n` elements of `elem` bytes in one of the ingest's scratch arrays of `ctx` (grow-only; the old contents are not kept)
static hipError_t ingest_room_bytes(jl_ctx *ctx, void **d, size_t *cap, size_t elem, size_t n)
{
    if (*d && *cap >= n) return hipSuccess;
    if (*d) {
        hipError_t e = hipStreamSynchronize(ctx->stream);   // an earlier build may still read it
        if (e != hipSuccess) return e;
        hipFree(*d);
    }
    *d = nullptr;
    *cap = 0;
    const size_t want = n + n / 8 + 64;
    hipError_t e = hipMalloc(d, want * elem);
    if (e == hipSuccess) *cap = want;
    return e;
}
#define ingest_room(ctx, d, cap, n) ingest_room_bytes(ctx, (void **)(d), cap, sizeof(**(d)), n)

// What the last record ingest into `ctx` found wrong with the records (its kernels have run): the first malformed read.
// Called by the blocking builds, and by the first blocking call behind an enqueued one (jl_sync, jl_run_wait, the fetches).
int jl_ingest_verdict(jl_ctx *ctx)
{
    if (!ctx->ing_check_pending) return JL_OK;
    ctx->ing_check_pending = false;
    // (through the context's pinned block: a process's first pageable device-to-host copy costs the runtime milliseconds)
    unsigned long long both[2] = {0, ~0ull};     // (the counters, the verdict: kernels_ingest.hip jl_launch_ingest)
#ifdef JL_TUNING
    {   // the address checks of the tuning build's ingest kernels (kernels_ingest.hip JL_ING_CHECK)
        uint32_t w[16] = {0};
        if (int rc = jl_fetch_to_host(ctx, ctx->d_ing_count, 64, w, 64)) return rc;
        for (int c = 1; c < 5; ++c)
            if (w[4 + c]) fprintf(stderr, "ingest check %d failed %u times (a value: %u)\n", c, w[4 + c], w[9 + c]);
        if (getenv("JL_ING_STAMPS") && ctx->d_ing_slow) {   // phase stamps of the sampled workgroups (kernels_ingest.hip JL_ING_STAMP)
            std::vector<unsigned long long> t(160 * 4 * 12);
            hipMemcpy(t.data(), ctx->d_ing_slow, t.size() * 8, hipMemcpyDeviceToHost);
            double sum[4][12] = {{0}};
            int n = 0;
            for (int g = 0; g < 160; ++g) {
                const unsigned long long *q = &t[(size_t)g * 48];
                if (!q[0] || !q[10]) continue;
                ++n;
                for (int wv = 0; wv < 4; ++wv)
                    for (int k = 1; k <= 10; ++k)
                        if (q[wv * 12 + k] && q[wv * 12 + k - 1]) sum[wv][k] += 0.01 * (double)(q[wv * 12 + k] - q[wv * 12 + k - 1]);
                        else if (q[wv * 12 + k] && k >= 2 && q[wv * 12 + k - 2]) sum[wv][k] += 0.01 * (double)(q[wv * 12 + k] - q[wv * 12 + k - 2]);
            }
            if (n) {
                fprintf(stderr, "ingest stamps, %d workgroups, us per step (desc, prologue, barrier, fetch+fill, pieces, stage, barrier, general, barrier, transpose):\n", n);
                for (int wv = 0; wv < 4; ++wv) {
                    fprintf(stderr, "  wave %d:", wv);
                    for (int k = 1; k <= 10; ++k) fprintf(stderr, " %6.2f", sum[wv][k] / n);
                    fprintf(stderr, "\n");
                }
            }
        }
    }
#endif
    if (int rc = jl_fetch_to_host(ctx, ctx->d_ing_count, 16, both, 64)) return rc;
    const unsigned long long w = both[1];
    if (w == ~0ull) return JL_OK;
    // (read: the word is all ones again for the builds to come — behind whatever this context has enqueued)
    JL_HIP(ctx, hipMemsetAsync(ctx->d_ing_count + 2, 0xFF, 8, ctx->stream));
    const unsigned long long r = w >> 8;
    const unsigned code = (unsigned)(w & 0xFFu);
    ctx->pileup_done = ctx->call_done = ctx->phase_done = false;
    if (code == 1u) return jl_fail(ctx, JL_ERR_ARG, "record %llu: cigar M is forbidden in PacBio-compliant BAM (doc/JULIET.md:53)", r);
    if (code == 4u) return jl_fail(ctx, JL_ERR_ARG, "record %llu: its cigar spans 2^30 reference bases or more", r);
    if (code == 5u) return jl_fail(ctx, JL_ERR_STATE, "record %llu: a long cigar the upload had not seen (jl_ingest_read_is_long and cigar_walk_kernel disagree)", r);
    return jl_fail(ctx, JL_ERR_ARG, "record %llu: its cigar consumes more %s than the record holds", r, code == 2u ? "bases" : "qualities");
}

// The resident matrix of `dst` from the records uploaded to `src` (the same context for jl_records_finish; another one of
// the same device when one upload feeds several column windows).  The records stay.  Everything is ENQUEUED on dst's
// stream (three launches + the insertion counters when asked for); `wait`: return when it has run.
static int records_build(jl_ctx *src, jl_ctx *dst, uint32_t n_cols, uint32_t win_begin, uint32_t min_qv, bool wait)
{
    jl_records &R = src->rec;
    int rc = jl_msa_alloc(dst, R.n_reads, n_cols, win_begin);
    if (rc) return rc;
    hipStream_t st = dst->stream;
    hipError_t e = hipSuccess;
    if (!R.n_reads) {   // nothing was appended: the offset arrays still need their first entry
        e = records_room(src, R.d_co, R.cap_co, 0, 1, 0);
        if (e == hipSuccess) e = records_room(src, R.d_so, R.cap_so, 0, 1, 0);
        if (e == hipSuccess) e = records_room(src, R.d_pos, R.cap_pos, 0, 1, 0);
        if (e == hipSuccess) e = records_room(src, R.d_cig, R.cap_cig, 0, 1, 64);
        if (e == hipSuccess) e = records_room(src, R.d_seq, R.cap_seq, 0, 1, 64);
        if (e == hipSuccess) e = hipMemsetAsync(R.d_co, 0, 8, st);
        if (e == hipSuccess) e = hipMemsetAsync(R.d_so, 0, 8, st);
    }
    const uint32_t ns = jl_ingest_sweeps(n_cols);
    const size_t nr = (size_t)R.n_reads;
    if (e == hipSuccess) e = ingest_room(dst, &dst->d_ing_runs, &dst->ing_cap_runs, (size_t)R.n_cig + 3 * nr + 8);   // (three entries around a read's runs; + 8: the planes kernel reads entries four and eight at a time)
    if (e == hipSuccess) e = ingest_room(dst, &dst->d_ing_nruns, &dst->ing_cap_reads, nr + 1);
    if (e == hipSuccess) e = ingest_room(dst, &dst->d_ing_desc, &dst->ing_cap_desc, (nr + 1) * ns);
    if (e == hipSuccess) e = ingest_room(dst, &dst->d_ing_slow, &dst->ing_cap_slow, jl_ingest_slow_room(dst));
    if (e == hipSuccess && !dst->d_ing_count) {
        e = hipMalloc(&dst->d_ing_count, 64);
        // (counters zero, the verdict word — [2..3] — all ones: no malformed record seen; kernels_ingest.hip jl_launch_ingest)
        if (e == hipSuccess) e = hipMemsetAsync(dst->d_ing_count, 0, 64, st);
        if (e == hipSuccess) e = hipMemsetAsync(dst->d_ing_count + 2, 0xFF, 8, st);
    }
    dst->ins_valid = false;
    if (e == hipSuccess && dst->track_insertions) {
        if (dst->ins_capacity < n_cols) {
            if (dst->d_ins_len) hipFree(dst->d_ins_len);
            if (dst->d_ins_base) hipFree(dst->d_ins_base);
            dst->d_ins_len = dst->d_ins_base = nullptr;
            dst->ins_capacity = 0;
            e = hipMalloc(&dst->d_ins_len, (size_t)n_cols * JL_INS_LEN_BINS * 4);
            if (e == hipSuccess) e = hipMalloc(&dst->d_ins_base, (size_t)n_cols * JL_INS_MAX_BASES * 16);
            if (e == hipSuccess) dst->ins_capacity = n_cols;
        }
        if (e == hipSuccess) e = hipMemsetAsync(dst->d_ins_len, 0, (size_t)n_cols * JL_INS_LEN_BINS * 4, st);
        if (e == hipSuccess) e = hipMemsetAsync(dst->d_ins_base, 0, (size_t)n_cols * JL_INS_MAX_BASES * 16, st);
        if (e == hipSuccess) {
            jl_launch_insertions(dst, R.d_pos, R.d_cig, R.d_co, R.d_seq, R.d_so);
            e = hipGetLastError();
            dst->ins_valid = e == hipSuccess;
        }
    }
    if (e == hipSuccess) {
        jl_launch_ingest(dst, R.d_pos, R.d_cig, R.d_co, R.d_seq, R.d_so, R.have_qual ? R.d_qual : nullptr,
                         R.have_qual ? R.d_qo : nullptr, min_qv, dst->d_ing_runs, dst->d_ing_nruns, dst->d_ing_desc, dst->d_ing_count,
                         dst->d_ing_slow, R.maybe_long, R.n_seq, R.n_cig + 3 * nr + 8);
        e = hipGetLastError();
        dst->ing_check_pending = e == hipSuccess;
        if (e == hipSuccess && wait) e = hipStreamSynchronize(st);
    }
    if (e != hipSuccess) return jl_fail(dst, e == hipErrorOutOfMemory ? JL_ERR_MEMORY : JL_ERR_DEVICE, "ingest: %s", hipGetErrorString(e));
    return wait ? jl_ingest_verdict(dst) : JL_OK;
}

int jl_records_finish(jl_ctx *ctx, uint32_t n_cols, uint32_t win_begin, uint32_t min_qv)
{
    if (!ctx) return JL_ERR_ARG;
    if (!ctx->rec.open) return jl_fail(ctx, JL_ERR_STATE, "jl_records_finish before jl_records_begin");
    JL_HIP(ctx, hipSetDevice(ctx->device));
    const int rc = records_build(ctx, ctx, n_cols, win_begin, min_qv, true);
    records_drop(ctx);
    return rc;
}

int jl_records_window(jl_ctx *records, jl_ctx *window, uint32_t n_cols, uint32_t win_begin, uint32_t min_qv)
{
    if (!records || !window) return JL_ERR_ARG;
    if (!records->rec.open) return jl_fail(window, JL_ERR_STATE, "jl_records_window: no records uploaded (jl_records_begin / _append)");
    if (records->device != window->device) return jl_fail(window, JL_ERR_ARG, "records and window are on different devices");
    JL_HIP(window, hipSetDevice(window->device));
    JL_HIP(window, hipStreamSynchronize(records->stream));   // the uploads are complete
    return records_build(records, window, n_cols, win_begin, min_qv, true);
}

// The same, enqueued only: the window's matrix is complete when the window's stream reaches this point — a run enqueued
// behind it on that stream (jl_run_async) reads it.  No allocation once a window of this shape has been built on `window`.
int jl_records_window_async(jl_ctx *records, jl_ctx *window, uint32_t n_cols, uint32_t win_begin, uint32_t min_qv)
{
    if (!records || !window) return JL_ERR_ARG;
    if (!records->rec.open) return jl_fail(window, JL_ERR_STATE, "jl_records_window_async: no records uploaded (jl_records_begin / _append)");
    if (records->device != window->device) return jl_fail(window, JL_ERR_ARG, "records and window are on different devices");
    JL_HIP(window, hipSetDevice(window->device));
    return records_build(records, window, n_cols, win_begin, min_qv, false);
}

int jl_records_drop(jl_ctx *ctx)
{
    if (!ctx) return JL_ERR_ARG;
    JL_HIP(ctx, hipSetDevice(ctx->device));
    records_drop(ctx);
    return JL_OK;
}

int jl_msa_ingest_records(jl_ctx *ctx, uint64_t n_reads, uint32_t n_cols, uint32_t win_begin, const int32_t *pos,
                          const uint32_t *cigar, const uint64_t *cig_off, const uint8_t *seq4, const uint64_t *seq_off,
                          const uint8_t *qual, const uint64_t *qual_off, uint32_t min_qv)
{
    if (!ctx || !pos || !cigar || !cig_off || !seq4 || !seq_off || (qual && !qual_off)) return JL_ERR_ARG;
    int rc = jl_records_begin(ctx, n_reads, cig_off[n_reads] - cig_off[0], seq_off[n_reads] - seq_off[0],
                              qual ? std::max<uint64_t>(qual_off[n_reads] - qual_off[0], 1) : 0);
    if (rc == JL_OK) rc = jl_records_append(ctx, n_reads, pos, cigar, cig_off, seq4, seq_off, qual, qual_off);
    if (rc == JL_OK) rc = jl_records_finish(ctx, n_cols, win_begin, min_qv);
    else if (ctx->rec.open) records_drop(ctx);
    return rc;
}

int jl_msa_track_insertions(jl_ctx *ctx, int on)
{
    if (!ctx) return JL_ERR_ARG;
    ctx->track_insertions = on != 0;
    return JL_OK;
}

int jl_insertions_fetch(jl_ctx *ctx, uint32_t *len_hist, uint32_t *base_counts)
{
    if (!ctx) return JL_ERR_ARG;
    if (!ctx->ins_valid) return jl_fail(ctx, JL_ERR_STATE, "no insertion counts: jl_msa_track_insertions(ctx, 1) before jl_msa_ingest_records");
    JL_HIP(ctx, hipSetDevice(ctx->device));
    if (len_hist) JL_HIP(ctx, hipMemcpyAsync(len_hist, ctx->d_ins_len, (size_t)ctx->n_cols * JL_INS_LEN_BINS * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (base_counts) JL_HIP(ctx, hipMemcpyAsync(base_counts, ctx->d_ins_base, (size_t)ctx->n_cols * JL_INS_MAX_BASES * 16, hipMemcpyDeviceToHost, ctx->stream));
    JL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JL_OK;
}

// The resident planes back in the interchange format (column stride jl_col_stride(n_reads)), through the same bounded staging.
int jl_msa_download(jl_ctx *ctx, uint8_t *colpacked, uint64_t bytes)
{
    if (!ctx || !colpacked || !ctx->d_msa) return JL_ERR_ARG;
    const uint64_t stride = jl_col_stride(ctx->n_reads);
    if (bytes > stride * ctx->n_cols) return jl_fail(ctx, JL_ERR_ARG, "download larger than the matrix");
    JL_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t n_cols = (uint32_t)((bytes + stride - 1) / stride);
    if (!n_cols) return JL_OK;
    const uint32_t per = (uint32_t)std::min<uint64_t>(n_cols, std::max<uint64_t>(1, ((uint64_t)64 << 20) / stride));
    uint8_t *d_stage = nullptr;
    JL_HIP(ctx, hipMalloc(&d_stage, (size_t)per * stride));
    hipError_t e = hipSuccess;
    for (uint32_t c0 = 0; c0 < n_cols && e == hipSuccess; c0 += per) {
        const uint32_t n = std::min(per, n_cols - c0);
        jl_launch_planes_to_nibbles(ctx, d_stage, stride, c0, n);
        e = hipGetLastError();
        const uint64_t lo = (uint64_t)c0 * stride, hi = std::min<uint64_t>(bytes, lo + (uint64_t)n * stride);
        if (e == hipSuccess) e = hipMemcpyAsync(colpacked + lo, d_stage, (size_t)(hi - lo), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
    hipFree(d_stage);
    if (e != hipSuccess) return jl_fail(ctx, JL_ERR_DEVICE, "download: %s", hipGetErrorString(e));
    return JL_OK;
}

// `ref` holds ref_len base codes of the WHOLE reference; the resident matrix is its window [win_begin, win_begin + n_cols)
static int synth_fill(jl_ctx *ctx, const jl_synth_params *sp, const uint8_t *ref, uint32_t ref_len, uint32_t col0)
{
    jl_synth_plan plan;
    jl_synth_make_plan(&plan, sp->seed, ref_len, sp->sub_rate, sp->del_rate, sp->mask_rate, sp->partial_rate,
                       sp->minor_permille, ref);
    uint8_t *d_ref = nullptr;
    JL_HIP(ctx, hipMalloc(&d_ref, ctx->n_cols));
    hipError_t e = hipMemcpyAsync(d_ref, ref + col0, ctx->n_cols, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        jl_launch_synth(ctx, &plan, d_ref, col0);
        e = hipStreamSynchronize(ctx->stream);
    }
    hipFree(d_ref);
    if (e != hipSuccess) return jl_fail(ctx, JL_ERR_DEVICE, "synth_fill: %s", hipGetErrorString(e));
    ctx->pileup_done = ctx->call_done = ctx->phase_done = false;
    ctx->pack_valid = false;
    return JL_OK;
}

int jl_synth_fill(jl_ctx *ctx, const jl_synth_params *sp, const uint8_t *ref)
{
    if (!ctx || !sp || !ref || !ctx->d_msa) return JL_ERR_ARG;
    return synth_fill(ctx, sp, ref, ctx->n_cols, 0);
}

int jl_synth_fill_window(jl_ctx *ctx, const jl_synth_params *sp, const uint8_t *ref, uint32_t ref_len)
{
    if (!ctx || !sp || !ref || !ctx->d_msa) return JL_ERR_ARG;
    if ((uint64_t)ctx->win_begin + ctx->n_cols > ref_len)
        return jl_fail(ctx, JL_ERR_ARG, "window [%u, %u) is not inside the %u-column reference", ctx->win_begin,
                       ctx->win_begin + ctx->n_cols, ref_len);
    return synth_fill(ctx, sp, ref, ref_len, ctx->win_begin);
}

/* ---------------------------------------------------------------- call */

// per-column device arrays (pileup outputs, plan flags, phasing maps)
static int reserve_columns(jl_ctx *ctx)
{
    int rc;
    if (ctx->col_capacity < ctx->n_cols) {
        if ((rc = regrow(ctx, &ctx->d_guess, (size_t)ctx->n_cols + JL_GUESS_PAD))) return rc;
        if ((rc = regrow(ctx, &ctx->d_col2pos, ctx->n_cols))) return rc;
        if ((rc = regrow(ctx, &ctx->d_varcol, ctx->n_cols))) return rc;
        if ((rc = regrow(ctx, &ctx->d_col_head, ctx->n_cols))) return rc;
        ctx->counts_words = (size_t)ctx->n_cols * (6 + 64);
        if ((rc = regrow(ctx, &ctx->d_counts, ctx->counts_words))) return rc;
        ctx->col_capacity = ctx->n_cols;
    }
    ctx->counts_words = (size_t)ctx->n_cols * (6 + 64);
    ctx->d_hist = ctx->d_counts + (size_t)ctx->n_cols * 6;
    return JL_OK;
}

// Column chunks of the pileup kernel.  Measured on MI355X: 3-column chunks that START ON A CODON
// need no halo columns and run fastest; where reading frames mix densely, 6-column chunks amortise the
// two halo columns.  The walk below starts a chunk at every codon start it meets, so stretches that are
// locally single-frame (different genes in different frames, as in HIV) stay halo-free.
static void build_chunks(jl_ctx *ctx, const std::vector<uint8_t> &colflag, std::vector<uint32_t> &c0s,
                         std::vector<uint8_t> &ns)
{
    const uint32_t L = ctx->n_cols;
    size_t per_frame[3] = {0, 0, 0}, total = 0;
    for (uint32_t c = 0; c < L; ++c)
        if (colflag[c] & 1) { per_frame[c % 3]++; ++total; }
    const size_t major = std::max(per_frame[0], std::max(per_frame[1], per_frame[2]));
    // dense mixing = more than a quarter of the codon starts fall outside their neighbourhood's frame; estimated
    // globally by counting starts that have another start within the two following columns
    size_t crowded = 0;
    for (uint32_t c = 0; c + 2 < L; ++c)
        if ((colflag[c] & 1) && ((colflag[c + 1] & 1) || (colflag[c + 2] & 1))) ++crowded;
    uint32_t W = (total && crowded * 4 > total) ? 6u : 3u;
#ifdef JL_TUNING
    if (const char *env_w = getenv("JL_PILEUP_W"))
        if (*env_w) W = (uint32_t)atoi(env_w) % 100u;
#endif
    if (W != 3 && W != 6) W = 3;   // the kernels exist for these two widths
    (void)major;
    ctx->pileup_w = W;
    c0s.clear();
    ns.clear();
    if (W != 3) {  // uniform grid
        for (uint32_t c = 0; c < L; c += W) { c0s.push_back(c); ns.push_back((uint8_t)std::min(W, L - c)); }
        return;
    }
    uint32_t c = 0;
    while (c < L) {
        if (colflag[c] & 1) {  // chunk = this codon's three columns
            const uint32_t n = std::min(3u, L - c);
            c0s.push_back(c);
            ns.push_back((uint8_t)n);
            c += n;
        } else {  // filler up to the next codon start (or three columns)
            uint32_t n = 1;
            while (n < 3 && c + n < L && !(colflag[c + n] & 1)) ++n;
            c0s.push_back(c);
            ns.push_back((uint8_t)n);
            c += n;
        }
    }
}

// SPEC §3: evaluated positions of every gene, in (gene, k) order
static int build_plan(jl_ctx *ctx, const jl_gene *genes, uint32_t n_genes, const uint8_t *refseq, uint32_t ref_len)
{
    ctx->genes.assign(genes, genes + n_genes);
    ctx->have_ref = refseq != nullptr;
    if (refseq) ctx->refseq.assign(refseq, refseq + ref_len);
    else ctx->refseq.clear();
    ctx->h_pos_gene.clear();
    ctx->h_pos_codon.clear();
    ctx->h_pos_col.clear();
    std::vector<uint8_t> refcfg, colflag(ctx->n_cols, 0), guess((size_t)ctx->n_cols + JL_GUESS_PAD, 0);
    double n_tests = 0.0;
    for (uint32_t g = 0; g < n_genes; ++g) {
        if (genes[g].begin == 0 || genes[g].end <= genes[g].begin) continue;
        const uint32_t ncod = (genes[g].end - genes[g].begin) / 3;
        n_tests += (double)ncod;
        for (uint32_t k = 0; k < ncod; ++k) {
            const int64_t r = (int64_t)genes[g].begin - 1 + 3 * (int64_t)k;
            const int64_t c = r - (int64_t)ctx->win_begin;
            if (c < 0 || c + 2 >= (int64_t)ctx->n_cols) continue;
            uint8_t cfg = JL_REF_MAJORITY;
            if (refseq) {
                if ((uint64_t)r + 2 >= ref_len || refseq[r] > 3 || refseq[r + 1] > 3 || refseq[r + 2] > 3) cfg = JL_REF_SKIP;
                else cfg = (uint8_t)(16 * refseq[r] + 4 * refseq[r + 1] + refseq[r + 2]);
            }
            ctx->h_pos_gene.push_back(g);
            ctx->h_pos_codon.push_back(k + 1);
            ctx->h_pos_col.push_back((uint32_t)c);
            refcfg.push_back(cfg);
            colflag[c] |= 1;
        }
    }
    ctx->default_n_tests = n_tests;
    ctx->P = (uint32_t)ctx->h_pos_col.size();
    std::vector<uint32_t> chunk_c0;
    std::vector<uint8_t> chunk_n;
    build_chunks(ctx, colflag, chunk_c0, chunk_n);
    if (refseq)
        for (uint32_t c = 0; c < ctx->n_cols; ++c) {
            const uint64_t r = (uint64_t)c + ctx->win_begin;
            guess[c] = (r < ref_len && refseq[r] < 4) ? refseq[r] : 0;
        }

    int rc;
    if ((rc = reserve_columns(ctx))) return rc;
    const size_t P = ctx->P ? ctx->P : 1;
    if (ctx->pos_capacity < P) {
        if ((rc = regrow(ctx, &ctx->d_pos_gene, P))) return rc;
        if ((rc = regrow(ctx, &ctx->d_pos_codon, P))) return rc;
        if ((rc = regrow(ctx, &ctx->d_pos_col, P))) return rc;
        if ((rc = regrow(ctx, &ctx->d_pos_refcfg, P))) return rc;
        if ((rc = regrow(ctx, &ctx->d_pos_next, P))) return rc;
        if ((rc = regrow(ctx, &ctx->d_called, P))) return rc;
        if ((rc = regrow(ctx, &ctx->d_staged, P * 64))) return rc;
        if ((rc = regrow(ctx, &ctx->d_drm, P))) return rc;
        ctx->pos_capacity = P;
    }
    ctx->n_chunks = (uint32_t)chunk_c0.size();
    if (ctx->chunk_capacity < chunk_c0.size()) {
        if ((rc = regrow(ctx, &ctx->d_chunks, chunk_c0.size()))) return rc;
        ctx->chunk_capacity = chunk_c0.size();
    }
    // chunk records: first column | own columns, codon-start flags of the own columns, "needs the two halo columns"
    std::vector<uint64_t> recs(chunk_c0.size());
    for (size_t k = 0; k < chunk_c0.size(); ++k) {
        const uint32_t c0 = chunk_c0[k], n = chunk_n[k];
        uint32_t startf = 0;
        for (uint32_t j = 0; j < n; ++j)
            if (colflag[c0 + j] & 1) startf |= 1u << j;
        const uint32_t halo = n >= 2u ? (startf >> (n - 2u)) != 0 : startf != 0;
        const uint32_t meta = n | (startf << 4) | (halo << 16);
        recs[k] = (uint64_t)c0 | ((uint64_t)meta << 32);
    }
    hipStream_t st = ctx->stream;
    JL_HIP(ctx, hipMemcpyAsync(ctx->d_chunks, recs.data(), recs.size() * 8, hipMemcpyHostToDevice, st));
    // the pad behind the last column is zero; in majority mode guess_kernel overwrites [0, n_cols) only
    JL_HIP(ctx, hipMemcpyAsync(ctx->d_guess, guess.data(), guess.size(), hipMemcpyHostToDevice, st));
    // the positions by column (the folded Fisher stage looks its positions up by the column it has just counted)
    std::vector<uint32_t> col_head(ctx->n_cols, 0xFFFFFFFFu), pos_next(P, 0xFFFFFFFFu);
    for (uint32_t q = ctx->P; q-- > 0;) {      // (from the last: the lists come out in position order)
        pos_next[q] = col_head[ctx->h_pos_col[q]];
        col_head[ctx->h_pos_col[q]] = q;
    }
    if (ctx->n_cols) JL_HIP(ctx, hipMemcpyAsync(ctx->d_col_head, col_head.data(), (size_t)ctx->n_cols * 4, hipMemcpyHostToDevice, st));
    if (ctx->P) {
        JL_HIP(ctx, hipMemcpyAsync(ctx->d_pos_next, pos_next.data(), P * 4, hipMemcpyHostToDevice, st));
        JL_HIP(ctx, hipMemcpyAsync(ctx->d_pos_gene, ctx->h_pos_gene.data(), P * 4, hipMemcpyHostToDevice, st));
        JL_HIP(ctx, hipMemcpyAsync(ctx->d_pos_codon, ctx->h_pos_codon.data(), P * 4, hipMemcpyHostToDevice, st));
        JL_HIP(ctx, hipMemcpyAsync(ctx->d_pos_col, ctx->h_pos_col.data(), P * 4, hipMemcpyHostToDevice, st));
        JL_HIP(ctx, hipMemcpyAsync(ctx->d_pos_refcfg, refcfg.data(), P, hipMemcpyHostToDevice, st));
    }
    // the staging vectors above are pageable: wait before they go out of scope
    JL_HIP(ctx, hipStreamSynchronize(st));
    ctx->plan_valid = true;
    ctx->plan_version++;
    return JL_OK;
}

static bool same_plan(const jl_ctx *ctx, const jl_gene *genes, uint32_t n_genes, const uint8_t *refseq, uint32_t ref_len)
{
    if (!ctx->plan_valid || ctx->genes.size() != n_genes) return false;
    if (n_genes && memcmp(ctx->genes.data(), genes, n_genes * sizeof(jl_gene)) != 0) return false;
    if (ctx->have_ref != (refseq != nullptr)) return false;
    if (refseq && (ctx->refseq.size() != ref_len || memcmp(ctx->refseq.data(), refseq, ref_len) != 0)) return false;
    return true;
}

int jl_pileup_async(jl_ctx *ctx, const jl_gene *genes, uint32_t n_genes, const uint8_t *refseq, uint32_t ref_len)
{
    if (!ctx || (!genes && n_genes)) return JL_ERR_ARG;
    if (!ctx->d_msa) return jl_fail(ctx, JL_ERR_STATE, "no resident matrix: call jl_msa_upload/alloc/adopt first");
    JL_HIP(ctx, hipSetDevice(ctx->device));
    if (!same_plan(ctx, genes, n_genes, refseq, ref_len)) {
        int rc = build_plan(ctx, genes, n_genes, refseq, ref_len);
        if (rc) return rc;
    }
    if (jl_pileup_needs_zero(ctx))
        JL_HIP(ctx, hipMemsetAsync(ctx->d_counts, 0, ctx->counts_words * sizeof(uint32_t), ctx->stream));
    if (!ctx->have_ref) jl_launch_guess(ctx, ctx->stream);
    jl_launch_pileup(ctx, ctx->stream);
    JL_HIP(ctx, hipGetLastError());
    ctx->pileup_done = true;
    ctx->call_done = ctx->phase_done = false;
    ctx->pack_valid = false;
    return JL_OK;
}

uint32_t jl_n_positions(const jl_ctx *ctx) { return ctx ? ctx->P : 0; }

}  // extern "C"

int jl_fetch_to_host(jl_ctx *ctx, const void *d_src, size_t bytes, void *dst, size_t readable)
{
    // `readable` >= bytes: how much of the source may be read (the copy moves whole 16-byte pieces)
    const size_t moved = (bytes + 15u) & ~(size_t)15u;
    hipStream_t st = ctx->run_stream ? ctx->run_stream : ctx->stream;
    if (ctx->run_stream && ctx->run_stream != ctx->stream) JL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (moved > ctx->h_scratch_cap || moved > readable || ((uintptr_t)d_src & 15u)) {   // large or odd: the runtime's copy
        JL_HIP(ctx, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, st));
        JL_HIP(ctx, hipStreamSynchronize(st));
        return JL_OK;
    }
    jl_launch_xw_fetch(d_src, ctx->h_scratch, moved, ctx->d_sync + 9, ctx->d_sync + 8, ctx->h_seq + 4, st);
    JL_HIP(ctx, hipGetLastError());
    const uint32_t want = ++ctx->fetches;
    volatile uint32_t *p = ctx->h_seq + 4;
    uint64_t spins = 0;
    while ((int32_t)(*p - want) < 0) {
        __builtin_ia32_pause();
        if ((++spins & 0x3FFFFFu) == 0) {   // a long wait: ask the stream (a fault would leave the word unset for ever)
            JL_HIP(ctx, hipStreamSynchronize(st));
            if ((int32_t)(*p - want) < 0) return jl_fail(ctx, JL_ERR_DEVICE, "fetch finished without its completion word");
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    memcpy(dst, ctx->h_scratch, bytes);
    return JL_OK;
}

extern "C" {

int jl_pileup_fetch(jl_ctx *ctx, uint32_t *col_counts, uint32_t *pos_gene, uint32_t *pos_codon, uint32_t *pos_col,
                    uint32_t *hist, uint32_t *coverage)
{
    if (!ctx) return JL_ERR_ARG;
    if (!ctx->pileup_done) return jl_fail(ctx, JL_ERR_STATE, "jl_pileup_fetch before jl_pileup_async");
    JL_HIP(ctx, hipSetDevice(ctx->device));
    if (col_counts)
        if (int rc = jl_fetch_to_host(ctx, ctx->d_counts, (size_t)ctx->n_cols * 6 * 4, col_counts, ctx->counts_words * 4)) return rc;
    std::vector<uint32_t> full;
    if ((hist || coverage) && ctx->P) {
        full.resize((size_t)ctx->n_cols * 64);
        if (int rc = jl_fetch_to_host(ctx, ctx->d_hist, full.size() * 4, full.data(), full.size() * 4)) return rc;
    }
    for (uint32_t p = 0; p < ctx->P; ++p) {
        if (pos_gene) pos_gene[p] = ctx->h_pos_gene[p];
        if (pos_codon) pos_codon[p] = ctx->h_pos_codon[p];
        if (pos_col) pos_col[p] = ctx->h_pos_col[p];
        if (hist || coverage) {
            const uint32_t *src = full.data() + (size_t)ctx->h_pos_col[p] * 64;
            uint32_t cov = 0;
            for (int j = 0; j < 64; ++j) cov += src[j];
            if (hist) memcpy(hist + (size_t)p * 64, src, 64 * 4);
            if (coverage) coverage[p] = cov;
        }
    }
    return JL_OK;
}

// Column consensus of the last pileup (the `fuse` by-product): out[c] = 0..3 majority base,
// 4 = majority deletion (column dropped from a consensus sequence), 5 = no covering read.
int jl_consensus_fetch(jl_ctx *ctx, uint8_t *out)
{
    if (!ctx || !out) return JL_ERR_ARG;
    if (!ctx->pileup_done) return jl_fail(ctx, JL_ERR_STATE, "jl_consensus_fetch before a pileup");
    JL_HIP(ctx, hipSetDevice(ctx->device));
    jl_launch_consensus(ctx, ctx->d_varcol);  // [n_cols] byte scratch, rewritten by every phase plan
    JL_HIP(ctx, hipMemcpyAsync(out, ctx->d_varcol, ctx->n_cols, hipMemcpyDeviceToHost, ctx->stream));
    JL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return JL_OK;
}

int jl_call_async(jl_ctx *ctx, const jl_params *prm, const uint64_t *drm_masks)
{
    if (!ctx || !prm) return JL_ERR_ARG;
    if (!ctx->pileup_done) return jl_fail(ctx, JL_ERR_STATE, "jl_call_async before jl_pileup_async");
    if (prm->tail != 0 && prm->tail != 1) return jl_fail(ctx, JL_ERR_ARG, "tail must be 0 (one-sided greater) or 1 (two-sided)");
    if (!(prm->alpha > 0.0) || !(prm->err.match > 0.0) || !(prm->err.substitution >= 0.0))
        return jl_fail(ctx, JL_ERR_ARG, "alpha/match must be > 0 and substitution >= 0");
    JL_HIP(ctx, hipSetDevice(ctx->device));
    if (drm_masks && ctx->P) {
        JL_HIP(ctx, hipMemcpyAsync(ctx->d_drm, drm_masks, (size_t)ctx->P * 8, hipMemcpyHostToDevice, ctx->stream));
        JL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    const double n_tests = prm->n_tests > 0.0 ? prm->n_tests : ctx->default_n_tests;
    jl_launch_call(ctx, ctx->stream, prm, n_tests, drm_masks != nullptr, false);
    jl_launch_compact(ctx, ctx->stream, false, false, false);
    JL_HIP(ctx, hipGetLastError());
    ctx->call_done = true;
    ctx->phase_done = false;
    ctx->pack_valid = false;
    return JL_OK;
}

int jl_call_fetch(jl_ctx *ctx, jl_variant *out, uint32_t cap, uint32_t *n_out)
{
    if (!ctx || !n_out || (!out && cap)) return JL_ERR_ARG;
    if (!ctx->call_done) return jl_fail(ctx, JL_ERR_STATE, "jl_call_fetch before jl_call_async");
    if (ctx->pack_valid) {  // one pinned copy already holds the table
        if (int rc = jl_run_wait_impl(ctx)) return rc;
        const jl_pack *pk = ctx->h_pack;
        if (pk->magic == JL_PACK_MAGIC && pk->fits_call) {
            *n_out = pk->nvar_total;
            const uint32_t take = pk->nvar_total < cap ? pk->nvar_total : cap;
            if (take) memcpy(out, pk->variants, (size_t)take * sizeof(jl_variant));
            if (pk->nvar_total > cap) return jl_fail(ctx, JL_ERR_OVERFLOW, "%u variant rows, capacity %u", pk->nvar_total, cap);
            return JL_OK;
        }
    }
    uint32_t n = 0;
    JL_HIP(ctx, hipMemcpyAsync(&n, ctx->d_nvar, 4, hipMemcpyDeviceToHost, ctx->stream));
    JL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_out = n;
    uint32_t have = n < JL_VARIANT_CAP ? n : JL_VARIANT_CAP;
    uint32_t take = have < cap ? have : cap;
    if (take) {
        JL_HIP(ctx, hipMemcpyAsync(out, ctx->d_variants, (size_t)take * sizeof(jl_variant), hipMemcpyDeviceToHost, ctx->stream));
        JL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (n > take) return jl_fail(ctx, JL_ERR_OVERFLOW, "%u variant rows, capacity %u (device table holds %u)", n, cap, JL_VARIANT_CAP);
    return JL_OK;
}

int jl_variant_table_device(jl_ctx *ctx, void **d_rows, void **d_count, uint32_t *cap_rows)
{
    if (!ctx) return JL_ERR_ARG;
    if (d_rows) *d_rows = ctx->d_variants;
    if (d_count) *d_count = ctx->d_nvar;
    if (cap_rows) *cap_rows = JL_VARIANT_CAP;
    return JL_OK;
}

/* ---------------------------------------------------------------- phase */

static int reserve_phase(jl_ctx *ctx, uint32_t kwords_needed)
{
    const size_t reads_pad = (size_t)ctx->col_stride * 2;
    int rc;
    if (ctx->reads_capacity < reads_pad) {
        uint64_t slots = 1024;
        while (slots < 2 * (uint64_t)ctx->n_reads) slots <<= 1;
        ctx->table_slots = slots;
        if ((rc = regrow(ctx, &ctx->d_flagw, reads_pad / 8))) return rc;
        if ((rc = regrow(ctx, &ctx->d_blockcat, (reads_pad / 2048u + 2u) * 4u))) return rc;
        if ((rc = regrow(ctx, &ctx->d_read_slot, reads_pad))) return rc;
        if ((rc = regrow(ctx, &ctx->d_read_hap, reads_pad))) return rc;
        if ((rc = regrow(ctx, &ctx->d_occupied, reads_pad))) return rc;
        if ((rc = regrow(ctx, &ctx->d_slot_rep, (size_t)slots))) return rc;
        if ((rc = regrow(ctx, &ctx->d_slot_count, (size_t)slots))) return rc;
        if ((rc = regrow(ctx, &ctx->d_slot_key, (size_t)slots))) return rc;
        if ((rc = regrow(ctx, &ctx->d_slot_hap, (size_t)slots))) return rc;
        // the grouping table is initialised once; phase_select_kernel empties the slots a run touched
        JL_HIP(ctx, hipMemsetAsync(ctx->d_slot_rep, 0xFF, (size_t)slots * 4, ctx->stream));
        JL_HIP(ctx, hipMemsetAsync(ctx->d_slot_count, 0, (size_t)slots * 4, ctx->stream));
        JL_HIP(ctx, hipMemsetAsync(ctx->d_slot_key, 0xFF, (size_t)slots * 8, ctx->stream));
        JL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->reads_capacity = reads_pad;
        ctx->keys_capacity = 0;
    }
    if (ctx->phase_two && ctx->two_slots != ctx->table_slots) {   // the two-word fused launch numbers its half keys here
        void *old[] = {ctx->d_slot_key_a, ctx->d_slot_key_b, ctx->d_occ_a, ctx->d_occ_b};
        for (void *p : old)
            if (p) hipFree(p);
        ctx->d_slot_key_a = ctx->d_slot_key_b = nullptr;
        ctx->d_occ_a = ctx->d_occ_b = nullptr;
        ctx->two_slots = 0;
        ctx->alloc_version++;
        const size_t slots = (size_t)ctx->table_slots;
        JL_HIP(ctx, hipMalloc(&ctx->d_slot_key_a, slots * 8));
        JL_HIP(ctx, hipMalloc(&ctx->d_slot_key_b, slots * 8));
        JL_HIP(ctx, hipMalloc(&ctx->d_occ_a, reads_pad * 4));
        JL_HIP(ctx, hipMalloc(&ctx->d_occ_b, reads_pad * 4));
        JL_HIP(ctx, hipMemsetAsync(ctx->d_slot_key_a, 0xFF, slots * 8, ctx->stream));
        JL_HIP(ctx, hipMemsetAsync(ctx->d_slot_key_b, 0xFF, slots * 8, ctx->stream));
        JL_HIP(ctx, hipMemsetAsync(ctx->d_sync + 10, 0, 8, ctx->stream));
        JL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->two_slots = ctx->table_slots;
    }
    const size_t need = (size_t)kwords_needed * reads_pad;
    if (ctx->keys_capacity < need) {
        if ((rc = regrow(ctx, &ctx->d_keys, need))) return rc;
        ctx->keys_capacity = need;
    }
    ctx->keys_words = (uint32_t)(ctx->keys_capacity / reads_pad);
    return JL_OK;
}

// room for the groups an exporting run writes out: one count and kwords * JL_POS_PER_WORD pattern bytes per group, for
// as many groups as the table can hold, within 256 MB
static int reserve_export(jl_ctx *ctx, uint32_t kwords)
{
    const uint32_t stride = (kwords * JL_POS_PER_WORD + 7u) / 8u * 8u;   // rows are written 8 bytes at a time
    uint64_t cap = ctx->table_slots ? ctx->table_slots : 1024;
    const uint64_t budget = ((uint64_t)256 << 20) / stride;
    if (cap > budget) cap = budget;
    if (ctx->exp_cap >= cap && ctx->exp_stride >= stride) return JL_OK;
    void *old[] = {ctx->d_exp_count, ctx->d_exp_pattern, ctx->d_exp_hap};
    for (void *p : old)
        if (p) hipFree(p);
    ctx->d_exp_count = nullptr; ctx->d_exp_pattern = nullptr; ctx->d_exp_hap = nullptr;
    ctx->exp_cap = ctx->exp_stride = 0;
    JL_HIP(ctx, hipMalloc(&ctx->d_exp_count, (size_t)cap * 4));
    JL_HIP(ctx, hipMalloc(&ctx->d_exp_pattern, (size_t)cap * stride));
    JL_HIP(ctx, hipMalloc(&ctx->d_exp_hap, (size_t)cap * 2));
    ctx->exp_cap = (uint32_t)cap;
    ctx->exp_stride = stride;
    return JL_OK;
}

static int phase_async_impl(jl_ctx *ctx, const jl_variant *variants, uint32_t n_var, uint32_t min_reads, bool exporting)
{
    if (!ctx) return JL_ERR_ARG;
    if (!ctx->d_msa) return jl_fail(ctx, JL_ERR_STATE, "no resident matrix");
    JL_HIP(ctx, hipSetDevice(ctx->device));
    if (!variants && !ctx->call_done) return jl_fail(ctx, JL_ERR_STATE, "jl_phase_async(NULL) needs jl_call_async first");
    {
        int rcc = reserve_columns(ctx);  // a context that only phases (cross-window matrix) never ran a pileup plan
        if (rcc) return rcc;
    }
    uint32_t kwords;
    if (variants) {
        if (n_var > JL_VARIANT_CAP) return jl_fail(ctx, JL_ERR_OVERFLOW, "%u variants, table holds %u", n_var, JL_VARIANT_CAP);
        if (n_var) JL_HIP(ctx, hipMemcpyAsync(ctx->d_variants, variants, (size_t)n_var * sizeof(jl_variant), hipMemcpyHostToDevice, ctx->stream));
        uint32_t cnt[2] = {n_var, 0};
        JL_HIP(ctx, hipMemcpyAsync(ctx->d_nvar, cnt, 8, hipMemcpyHostToDevice, ctx->stream));
        JL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        kwords = (n_var + JL_POS_PER_WORD - 1) / JL_POS_PER_WORD;
    } else {
        if (!ctx->call_done) return jl_fail(ctx, JL_ERR_STATE, "jl_phase_async(NULL) needs jl_call_async first");
        // The row count lives on the device.  Keep what is allocated (at least 4 words = 40 variant
        // positions); if a run needs more, the plan kernel flags it and jl_phase_fetch re-runs it exactly.
        kwords = ctx->keys_words > 4 ? ctx->keys_words : 4;
    }
    if (kwords == 0) kwords = 1;
    int rc = reserve_phase(ctx, kwords);
    if (rc) return rc;
    if (exporting && (rc = reserve_export(ctx, kwords))) return rc;
    ctx->phase_export = exporting;
    ctx->direct.on = 0;
    ctx->exp_known = false;
    ctx->exp_ext_count = nullptr; ctx->exp_ext_pattern = nullptr; ctx->exp_ext_head = nullptr;
    ctx->exp_ext_cap = ctx->exp_ext_stride = 0;
    ctx->last_min_reads = min_reads;
    ctx->pack_mirror = nullptr;
    ctx->read_hap_out = nullptr;
    jl_launch_phase(ctx, ctx->stream, min_reads, false, false, false);
    JL_HIP(ctx, hipGetLastError());
    ctx->phase_done = true;
    ctx->pack_valid = false;
    return JL_OK;
}

int jl_phase_async(jl_ctx *ctx, const jl_variant *variants, uint32_t n_var, uint32_t min_reads)
{
    return phase_async_impl(ctx, variants, n_var, min_reads, false);
}

// Phasing sharded by reads (SURVEY §8e option A): keys and grouping of THIS matrix (a slice of the reads), the groups
// written out instead of ranked.  Every read keeps its flags and its slot for jl_phase_regroup.
int jl_phase_groups_async(jl_ctx *ctx, const jl_variant *variants, uint32_t n_var)
{
    return phase_async_impl(ctx, variants, n_var, 0xFFFFFFFFu, true);
}

// per-read ids in their packed form -> 16-bit ids (JL_ID4_MAX_H / JL_ID8_MAX_H in jl_internal.h)
void jl_expand_ids(const void *packed, uint32_t bits, uint64_t n_reads, uint16_t *out)
{
    if (bits == 4) {
        const uint8_t *p = (const uint8_t *)packed;
        for (uint64_t i = 0; i < n_reads; ++i) {
            const uint32_t c = (p[i >> 1] >> (4u * (i & 1u))) & 15u;
            out[i] = c == 15u ? (uint16_t)JL_HAP_DAMAGED : (c == 14u ? (uint16_t)JL_HAP_INSUFFICIENT : (uint16_t)c);
        }
    } else if (bits == 8) {
        const uint8_t *p = (const uint8_t *)packed;
        for (uint64_t i = 0; i < n_reads; ++i)
            out[i] = p[i] == 255u ? (uint16_t)JL_HAP_DAMAGED : (p[i] == 254u ? (uint16_t)JL_HAP_INSUFFICIENT : (uint16_t)p[i]);
    } else {
        memcpy(out, packed, (size_t)n_reads * 2);
    }
}

// the ids of the last phase launch: from the pinned block the kernels stored into, or copied out of HBM
static int fetch_ids(jl_ctx *ctx, uint32_t bits, uint16_t *read_hap)
{
    if (bits != 4 && bits != 8 && bits != 16) return jl_fail(ctx, JL_ERR_STATE, "per-read ids of unknown width %u", bits);
    if (ctx->read_hap_out) {
        jl_expand_ids(ctx->h_read_hap, bits, ctx->n_reads, read_hap);
        return JL_OK;
    }
    if (bits == 16) {
        JL_HIP(ctx, hipMemcpyAsync(read_hap, ctx->d_read_hap, (size_t)ctx->n_reads * 2, hipMemcpyDeviceToHost, ctx->stream));
        JL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return JL_OK;
    }
    const size_t bytes = bits == 4 ? (size_t)(ctx->n_reads + 1) / 2 : (size_t)ctx->n_reads;
    std::vector<uint8_t> tmp(bytes);
    JL_HIP(ctx, hipMemcpyAsync(tmp.data(), ctx->d_read_hap, bytes, hipMemcpyDeviceToHost, ctx->stream));
    JL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    jl_expand_ids(tmp.data(), bits, ctx->n_reads, read_hap);
    return JL_OK;
}

// Waits for the last phase launch and reads its scalars; a launch that found more variant positions than its kernels
// or its key buffer cover is run again with what it needs (the variant table is resident).
static int phase_settle(jl_ctx *ctx, jl_phase_meta *out)
{
    hipStream_t st = ctx->stream;
    if (ctx->run_stream && ctx->run_stream != st) JL_HIP(ctx, hipStreamSynchronize(ctx->run_stream));
    jl_phase_meta meta;
    JL_HIP(ctx, hipMemcpyAsync(&meta, ctx->d_meta, sizeof meta, hipMemcpyDeviceToHost, st));
    JL_HIP(ctx, hipStreamSynchronize(st));
    if (meta.overflow & 12u) {
        // more variant positions than the fused launch in use (bit 3) or the resident key buffer (bit 2) covers: up to 20
        // positions take the two-word fused launch, more — or a result beyond its selection — the multi-word pipeline;
        // grow the buffer if need be and run phasing again (the variant table is resident)
        for (int attempt = 0; attempt < 3 && (meta.overflow & 12u); ++attempt) {
            if (meta.vp_true > JL_POS_PER_WORD) {
                if (meta.vp_true <= 2u * JL_POS_PER_WORD && !ctx->phase_two && !ctx->phase_generic && !ctx->phase_export) ctx->phase_two = true;
                else ctx->phase_generic = true;
            }
            const uint32_t kw = (meta.vp_true + JL_POS_PER_WORD - 1) / JL_POS_PER_WORD;
            int rc = reserve_phase(ctx, kw);
            if (rc == JL_OK && ctx->phase_export) rc = reserve_export(ctx, kw);
            if (rc) return rc;
            jl_launch_phase(ctx, st, ctx->last_min_reads, false, false, false);
            JL_HIP(ctx, hipGetLastError());
            JL_HIP(ctx, hipMemcpyAsync(&meta, ctx->d_meta, sizeof meta, hipMemcpyDeviceToHost, st));
            JL_HIP(ctx, hipStreamSynchronize(st));
        }
    }
    if (meta.overflow & 32u) {
        // A workgroup of the folded launch gave up waiting for the selection (the launch's workgroups were not resident
        // together): some reads have no id.  The stage runs again with the ids in a launch of their own — transparently;
        // this context stays unfolded from now on.
        ctx->no_fold = true;
        ctx->fold_reruns++;
        ctx->alloc_version++;
        jl_launch_phase(ctx, st, ctx->last_min_reads, false, false, false);
        JL_HIP(ctx, hipGetLastError());
        JL_HIP(ctx, hipMemcpyAsync(&meta, ctx->d_meta, sizeof meta, hipMemcpyDeviceToHost, st));
        JL_HIP(ctx, hipStreamSynchronize(st));
        if (meta.overflow & 32u) return jl_fail(ctx, JL_ERR_DEVICE, "the unfolded phase launch reports a time-out");
    }
    *out = meta;
    return JL_OK;
}

int jl_phase_fetch(jl_ctx *ctx, jl_phase_summary *summary, uint32_t *pos_cols, uint32_t *hap_count,
                   uint8_t *hap_pattern, uint8_t *hit, uint16_t *read_hap, uint32_t *cooc, uint32_t cap_var)
{
    if (!ctx) return JL_ERR_ARG;
    if (!ctx->phase_done) return jl_fail(ctx, JL_ERR_STATE, "jl_phase_fetch before jl_phase_async");
    hipStream_t st = ctx->stream;
    if (ctx->pack_valid) {
        if (int rc = jl_run_wait_impl(ctx)) return rc;
        const jl_pack *pk = ctx->h_pack;
        if (pk->magic == JL_PACK_MAGIC && pk->phase_ran && pk->fits_phase && (!cooc || pk->cooc_fits)) {
            const uint32_t vp = pk->vp, H = pk->H, nv = pk->nv_phase;
            if ((pos_cols || hap_pattern) && vp > cap_var) return jl_fail(ctx, JL_ERR_OVERFLOW, "%u variant positions, caller capacity %u", vp, cap_var);
            if ((hit || cooc) && nv > cap_var) return jl_fail(ctx, JL_ERR_OVERFLOW, "%u variants, caller capacity %u", nv, cap_var);
            if (summary) *summary = pk->summary;
            if (pos_cols && vp) memcpy(pos_cols, pk->pos_cols, (size_t)vp * 4);
            if (hap_count && H) memcpy(hap_count, pk->hap_count, (size_t)H * 4);
            if (hap_pattern)
                for (uint32_t h = 0; h < H; ++h) memcpy(hap_pattern + (size_t)h * cap_var, pk->hap_pattern + (size_t)h * vp, vp);
            if (hit)
                for (uint32_t v = 0; v < nv; ++v) memcpy(hit + (size_t)v * JL_MAX_HAPLOTYPES, pk->hit + (size_t)v * H, H);
            if (cooc)
                for (uint32_t v = 0; v < nv; ++v) memcpy(cooc + (size_t)v * cap_var, pk->cooc + (size_t)v * nv, (size_t)nv * 4);
            if (read_hap) {
                // a group run's stream is not the context's: its kernels are complete (the completion word), but a
                // copy on the context stream must not overtake them in the eyes of the runtime
                if (!ctx->read_hap_out && ctx->run_stream && ctx->run_stream != st) JL_HIP(ctx, hipStreamSynchronize(ctx->run_stream));
                if (int rc = fetch_ids(ctx, pk->id_bits, read_hap)) return rc;
            }
            return JL_OK;
        }
    }
    jl_phase_meta meta;
    if (int rc = phase_settle(ctx, &meta)) return rc;
    if (read_hap)
        if (int rc = fetch_ids(ctx, meta.id_bits, read_hap)) return rc;
    if (summary) *summary = meta.summary;
    const uint32_t vp = meta.vp, H = meta.summary.n_haplotypes, nv = meta.n_var;
    if ((pos_cols || hap_pattern) && vp > cap_var) return jl_fail(ctx, JL_ERR_OVERFLOW, "%u variant positions, caller capacity %u", vp, cap_var);
    if ((hit || cooc) && nv > cap_var) return jl_fail(ctx, JL_ERR_OVERFLOW, "%u variants, caller capacity %u", nv, cap_var);
    if (pos_cols && vp) JL_HIP(ctx, hipMemcpyAsync(pos_cols, ctx->d_vpcols, (size_t)vp * 4, hipMemcpyDeviceToHost, st));
    if (hap_count && H) JL_HIP(ctx, hipMemcpyAsync(hap_count, ctx->d_hap_count, (size_t)H * 4, hipMemcpyDeviceToHost, st));
    // pitched copies of narrow rows go row by row in the runtime (8 us each): flat copies + a repack on the host instead
    std::vector<uint8_t> flat_pat;
    if (hap_pattern && H && vp) {
        flat_pat.resize((size_t)(H - 1) * JL_VARIANT_CAP + vp);
        JL_HIP(ctx, hipMemcpyAsync(flat_pat.data(), ctx->d_hap_pattern, flat_pat.size(), hipMemcpyDeviceToHost, st));
    }
    if (hit && nv && H)   // same pitch on both sides: one run of bytes
        JL_HIP(ctx, hipMemcpyAsync(hit, ctx->d_hit, (size_t)(nv - 1) * JL_MAX_HAPLOTYPES + H, hipMemcpyDeviceToHost, st));
    if (cooc && nv) {
        const uint32_t n = nv < ctx->cooc_cap ? nv : ctx->cooc_cap;
        JL_HIP(ctx, hipMemcpy2DAsync(cooc, (size_t)cap_var * 4, ctx->d_cooc, (size_t)ctx->cooc_cap * 4, (size_t)n * 4, n, hipMemcpyDeviceToHost, st));
    }
    JL_HIP(ctx, hipStreamSynchronize(st));
    for (uint32_t h = 0; h < H && !flat_pat.empty(); ++h) memcpy(hap_pattern + (size_t)h * cap_var, flat_pat.data() + (size_t)h * JL_VARIANT_CAP, vp);
    if (meta.overflow & 1u) return jl_fail(ctx, JL_ERR_OVERFLOW, "more than %u haplotype candidates", JL_CAND_CAP);
    return JL_OK;
}

// The groups of the last jl_phase_groups_async: for group q (the order is kept for jl_phase_regroup) its read count and
// its pattern, one codon code per variant position (patterns[q * pattern_stride + p]); the partial summary holds this
// matrix's damaged reads and marginals, its clean reads under insufficient_reads.
int jl_phase_groups_fetch(jl_ctx *ctx, uint8_t *patterns, uint32_t pattern_stride, uint32_t *counts, uint32_t cap_groups,
                          uint32_t *n_groups, uint32_t *n_positions, uint32_t *pos_cols, uint32_t cap_var,
                          jl_phase_summary *partial)
{
    if (!ctx || !n_groups || !n_positions) return JL_ERR_ARG;
    if (!ctx->phase_done || !ctx->phase_export) return jl_fail(ctx, JL_ERR_STATE, "jl_phase_groups_fetch needs jl_phase_groups_async first");
    jl_phase_meta meta;
    if (int rc = phase_settle(ctx, &meta)) return rc;
    if (meta.overflow & 16u)
        return jl_fail(ctx, JL_ERR_OVERFLOW, "%u groups of reads, the export holds %u", meta.n_occupied, ctx->exp_cap);
    const uint32_t vp = meta.vp, ng = vp ? meta.n_occupied : 0u;
    *n_groups = ng;
    *n_positions = vp;
    ctx->exp_n_groups = ng;   // what jl_phase_regroup has to answer for
    ctx->exp_vp = vp;
    ctx->exp_known = true;
    if (partial) *partial = meta.summary;
    if (pos_cols && vp > cap_var) return jl_fail(ctx, JL_ERR_OVERFLOW, "%u variant positions, caller capacity %u", vp, cap_var);
    if ((patterns || counts) && ng > cap_groups) return jl_fail(ctx, JL_ERR_OVERFLOW, "%u groups, caller capacity %u", ng, cap_groups);
    if (patterns && vp > pattern_stride) return jl_fail(ctx, JL_ERR_OVERFLOW, "%u variant positions, pattern stride %u", vp, pattern_stride);
    hipStream_t st = ctx->stream;
    if (pos_cols && vp) JL_HIP(ctx, hipMemcpyAsync(pos_cols, ctx->d_vpcols, (size_t)vp * 4, hipMemcpyDeviceToHost, st));
    if (counts && ng) JL_HIP(ctx, hipMemcpyAsync(counts, ctx->d_exp_count, (size_t)ng * 4, hipMemcpyDeviceToHost, st));
    // one flat copy and a repack on the host: a pitched copy of rows a few bytes wide goes row by row (0.8 ms for 100 groups)
    std::vector<uint8_t> flat;
    if (patterns && ng && vp) {
        flat.resize((size_t)ng * ctx->exp_stride);
        JL_HIP(ctx, hipMemcpyAsync(flat.data(), ctx->d_exp_pattern, flat.size(), hipMemcpyDeviceToHost, st));
    }
    JL_HIP(ctx, hipStreamSynchronize(st));
    for (uint32_t q = 0; q < ng && !flat.empty(); ++q) memcpy(patterns + (size_t)q * pattern_stride, flat.data() + (size_t)q * ctx->exp_stride, vp);
    return JL_OK;
}

// The merge's answer: hap_of_group[q] = haplotype id of exported group q (JL_HAP_INSUFFICIENT for a group that is not
// reported), n_haplotypes = haplotypes reported in all.  Maps this matrix's reads; read_hap (optional, [n_reads]) gets
// their ids (JL_HAP_DAMAGED for flagged reads).
int jl_phase_regroup(jl_ctx *ctx, const uint16_t *hap_of_group, uint32_t n_groups, uint32_t n_haplotypes, uint16_t *read_hap)
{
    if (!ctx || (!hap_of_group && n_groups)) return JL_ERR_ARG;
    if (!ctx->phase_done || !ctx->phase_export) return jl_fail(ctx, JL_ERR_STATE, "jl_phase_regroup needs jl_phase_groups_async first");
    if (!ctx->exp_known) return jl_fail(ctx, JL_ERR_STATE, "jl_phase_regroup needs jl_phase_groups_fetch first (the number of exported groups)");
    // one answer per exported group: a shorter table would leave reads of the other groups without a haplotype
    if (n_groups != ctx->exp_n_groups)
        return jl_fail(ctx, JL_ERR_ARG, "%u groups, the run exported %u", n_groups, ctx->exp_n_groups);
    if (n_haplotypes > JL_MAX_HAPLOTYPES) return jl_fail(ctx, JL_ERR_ARG, "%u haplotypes, at most %u have names", n_haplotypes, JL_MAX_HAPLOTYPES);
    for (uint32_t q = 0; q < n_groups; ++q)
        if (hap_of_group[q] != JL_HAP_INSUFFICIENT && hap_of_group[q] >= n_haplotypes)
            return jl_fail(ctx, JL_ERR_ARG, "group %u: haplotype %u of %u", q, hap_of_group[q], n_haplotypes);
    JL_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (n_groups) JL_HIP(ctx, hipMemcpyAsync(ctx->d_exp_hap, hap_of_group, (size_t)n_groups * 2, hipMemcpyHostToDevice, st));
    jl_launch_regroup(ctx, ctx->d_exp_hap, n_groups, n_haplotypes, ctx->exp_vp != 0);
    JL_HIP(ctx, hipGetLastError());
    JL_HIP(ctx, hipStreamSynchronize(st));   // hap_of_group may be pageable: the copy has read it by now
    if (read_hap) {
        const uint32_t bits = n_haplotypes <= JL_ID4_MAX_H ? 4u : (n_haplotypes <= JL_ID8_MAX_H ? 8u : 16u);
        if (int rc = fetch_ids(ctx, bits, read_hap)) return rc;
    }
    return JL_OK;
}

}  // extern "C"

// A session's exporting phase run (capi_xwin.hip): the buffers jl_phase_groups_async would reserve, for a compact matrix
// whose plan — vp positions at columns 3k — the session's own kernel writes.  The caller sets ctx->exp_ext_* and launches.
int jl_phase_groups_prepare(jl_ctx *ctx, uint32_t vp)
{
    if (!ctx || !ctx->d_msa) return JL_ERR_ARG;
    JL_HIP(ctx, hipSetDevice(ctx->device));
    int rc = reserve_columns(ctx);
    if (rc) return rc;
    uint32_t kwords = (vp + JL_POS_PER_WORD - 1) / JL_POS_PER_WORD;
    if (kwords == 0) kwords = 1;
    // up to 10 positions one key word, up to 20 the two-word fused launch, more the multi-word pipeline
    ctx->phase_two = vp > JL_POS_PER_WORD && vp <= 2u * JL_POS_PER_WORD;
    ctx->phase_generic = vp > 2u * JL_POS_PER_WORD;
    if ((rc = reserve_phase(ctx, kwords))) return rc;
    ctx->direct.on = 0;
    ctx->phase_export = true;
    ctx->exp_known = false;
    ctx->last_min_reads = 0xFFFFFFFFu;
    ctx->pack_mirror = nullptr;
    ctx->read_hap_out = nullptr;
    ctx->phase_done = true;
    ctx->pack_valid = false;
    return JL_OK;
}

// The variant table of the context's last call stage, on the host.
int jl_ctx_table_host(jl_ctx *ctx, std::vector<jl_variant> *scratch, const jl_variant **rows, uint32_t *n)
{
    if (!ctx->call_done) return jl_fail(ctx, JL_ERR_STATE, "no variant table: run the call stage first");
    if (ctx->pack_valid) {
        if (int rc = jl_run_wait_impl(ctx)) return rc;
        const jl_pack *pk = ctx->h_pack;
        if (pk->magic == JL_PACK_MAGIC && pk->fits_call) {
            *rows = pk->variants;
            *n = pk->nvar_total;
            return JL_OK;
        }
    }
    JL_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->run_stream && ctx->run_stream != ctx->stream) JL_HIP(ctx, hipStreamSynchronize(ctx->run_stream));
    uint32_t cnt = 0;
    JL_HIP(ctx, hipMemcpyAsync(&cnt, ctx->d_nvar, 4, hipMemcpyDeviceToHost, ctx->stream));
    JL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (cnt > JL_VARIANT_CAP) return jl_fail(ctx, JL_ERR_OVERFLOW, "%u variant rows, the device table holds %u", cnt, JL_VARIANT_CAP);
    scratch->resize(cnt ? cnt : 1);
    if (cnt) {
        JL_HIP(ctx, hipMemcpyAsync(scratch->data(), ctx->d_variants, (size_t)cnt * sizeof(jl_variant), hipMemcpyDeviceToHost, ctx->stream));
        JL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    *rows = scratch->data();
    *n = cnt;
    return JL_OK;
}

extern "C" {

/* ---------------------------------------------------------------- the whole path as one enqueue */

// One window through the path: the launches of a step.
//   counting + Fisher   ONE launch when one workgroup counts a chunk alone (the Fisher stage rides in the epilogue
//                       of the pileup launch); otherwise the pileup and call_kernel reading the histograms back
//   phasing on          the fused phase launch: plan out of the call masks, keys, grouping, selection, result block,
//                       and — small windows — the per-read ids and the completion word
//   phasing off         compact_kernel: ordered table + result block
//   multi-word phasing  compact_kernel (table + plan), then the generic phase pipeline
static void enqueue_path(jl_ctx *ctx, const jl_params *prm, double n_tests, bool use_drm, bool phasing, uint32_t min_reads,
                         bool want_read_hap)
{
    hipStream_t st = ctx->stream;
    (void)want_read_hap;   // no copy nodes: results are stored straight into pinned host memory (ctx->pack_mirror / read_hap_out)
    if (jl_pileup_needs_zero(ctx)) hipMemsetAsync(ctx->d_counts, 0, ctx->counts_words * sizeof(uint32_t), st);
    if (!ctx->have_ref) jl_launch_guess(ctx, st);
    jl_launch_stamp(ctx, 0);
    if (ctx->pileup_clock) jl_launch_clock(ctx, st, 0);
    if (jl_pileup_can_fold(ctx)) {
        // every chunk is counted by ONE workgroup: it tests its codon from the histogram still in LDS (kernels_pileup.hip) —
        // no call launch, 8 us of a 64 us window
        jl_win_call w;
        jl_fill_win_call(ctx, prm, n_tests, use_drm, phasing, &w);
        jl_launch_pileup_fold(ctx, st, &w);
        if (ctx->pileup_clock) jl_launch_clock(ctx, st, 1);
    } else {
        jl_launch_pileup(ctx, st);
        if (ctx->pileup_clock) jl_launch_clock(ctx, st, 1);
        jl_launch_call(ctx, st, prm, n_tests, use_drm, phasing);
    }
    jl_launch_stamp(ctx, 1);
    // The completion word (jl_run_wait) is stored by a one-thread node of its own behind the last stage: the end of
    // that stage's kernel is what pushes the results every compute die wrote for the host out of the dies' L2s.
    bool signaled = false;
    if (!phasing) {
        // ONE workgroup compacts the table and writes the result block: it stores the completion word itself, behind its own
        // drained stores and a system-scope fence (no other die has written anything the host reads)
        jl_launch_compact(ctx, st, false, true, true);
        signaled = true;
    } else if (ctx->phase_generic) {
        jl_launch_compact(ctx, st, true, false, false);
        jl_launch_stamp(ctx, 2);
        jl_launch_phase(ctx, st, min_reads, true, false, false);
    } else {
        signaled = jl_launch_phase(ctx, st, min_reads, true, true, JL_SIGNAL_IN_KERNEL != 0);
    }
    jl_launch_stamp(ctx, 3);
    if (!signaled) jl_launch_done(ctx);
}

// Everything of a run that allocates, uploads or waits: done before the enqueue (and before any capture).  Shared by
// jl_run_async and jl_group_run_async.
int jl_run_prepare(jl_ctx *ctx, const jl_gene *genes, uint32_t n_genes, const uint8_t *refseq, uint32_t ref_len,
                   const jl_params *prm, const uint64_t *drm_masks, int phasing, uint32_t min_reads, int want_read_hap,
                   double *n_tests_out)
{
    if (!ctx || !prm || (!genes && n_genes)) return JL_ERR_ARG;
    if (!ctx->d_msa) return jl_fail(ctx, JL_ERR_STATE, "no resident matrix: call jl_msa_upload/alloc/adopt first");
    ctx->phase_export = false;
    ctx->direct.on = 0;
    if (prm->tail != 0 && prm->tail != 1) return jl_fail(ctx, JL_ERR_ARG, "tail must be 0 (one-sided greater) or 1 (two-sided)");
    if (!(prm->alpha > 0.0) || !(prm->err.match > 0.0) || !(prm->err.substitution >= 0.0))
        return jl_fail(ctx, JL_ERR_ARG, "alpha/match must be > 0 and substitution >= 0");
    // The device result block is double-buffered by run parity; an uncollected exchange still reads the block of its
    // run.  The next run (runs_launched + 1) writes the block of its own parity: refuse when that is a block in use.
    for (uint32_t r : ctx->exch_runs)
        if (((ctx->runs_launched + 1u - r) & 1u) == 0u)
            return jl_fail(ctx, JL_ERR_STATE, "the exchange of run %u is not collected yet and reads the result block this run "
                                              "would write: collect it (jl_allgather_variants) first", r);
    JL_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if (!same_plan(ctx, genes, n_genes, refseq, ref_len) && (rc = build_plan(ctx, genes, n_genes, refseq, ref_len))) return rc;
    if (drm_masks && ctx->P) {
        JL_HIP(ctx, hipMemcpyAsync(ctx->d_drm, drm_masks, (size_t)ctx->P * 8, hipMemcpyHostToDevice, ctx->stream));
        JL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (phasing) {
        if ((rc = reserve_phase(ctx, ctx->keys_words > 4 ? ctx->keys_words : 4))) return rc;   // (with the half-key tables once the context is in two-word mode)
        if (want_read_hap && ctx->h_read_hap_cap < (size_t)ctx->col_stride * 2) {
            if (ctx->h_read_hap) hipHostFree(ctx->h_read_hap);
            ctx->h_read_hap = nullptr;
            ctx->alloc_version++;
            JL_HIP(ctx, hipHostMalloc(&ctx->h_read_hap, (size_t)ctx->col_stride * 4, hipHostMallocDefault));
            ctx->h_read_hap_cap = (size_t)ctx->col_stride * 2;
        }
    }
#ifdef JL_TUNING
    if (!ctx->d_timeline && getenv("JL_TIMELINE")) {   // tuning aid, see stamp_kernel
        JL_HIP(ctx, hipMalloc(&ctx->d_timeline, (size_t)JL_TIMELINE_ROWS * JL_TIMELINE_SLOTS * 8));
        JL_HIP(ctx, hipMemset(ctx->d_timeline, 0, (size_t)JL_TIMELINE_ROWS * JL_TIMELINE_SLOTS * 8));
        ctx->alloc_version++;
    }
#endif
    *n_tests_out = prm->n_tests > 0.0 ? prm->n_tests : ctx->default_n_tests;
    ctx->last_min_reads = min_reads;
    jl_prepare_pileup(ctx);
    ctx->pack_mirror = ctx->h_pack;
    ctx->read_hap_out = (phasing && want_read_hap) ? ctx->h_read_hap : nullptr;
    return JL_OK;
}

void jl_run_finish(jl_ctx *ctx, int phasing, int want_read_hap)
{
    ctx->runs_launched++;
    ctx->pileup_done = ctx->call_done = true;
    ctx->phase_done = phasing != 0;
    ctx->pack_valid = true;
    ctx->run_read_hap = phasing && want_read_hap;
}

int jl_run_async(jl_ctx *ctx, const jl_gene *genes, uint32_t n_genes, const uint8_t *refseq, uint32_t ref_len,
                 const jl_params *prm, const uint64_t *drm_masks, int phasing, uint32_t min_reads, int want_read_hap)
{
    double n_tests = 0.0;
    int rc = jl_run_prepare(ctx, genes, n_genes, refseq, ref_len, prm, drm_masks, phasing, min_reads, want_read_hap, &n_tests);
    if (rc) return rc;
    if (ctx->run_stream && ctx->run_stream != ctx->stream) {
        // the previous run of this context was a group run on the group's stream: this one, on the context's own
        // stream, must not overtake it (same buffers, and the result block's parity is counted on the device)
        JL_HIP(ctx, hipStreamSynchronize(ctx->run_stream));
        ctx->run_stream = ctx->stream;
    }

    // signature of everything a captured graph bakes in
    struct { uint64_t alloc, plan; jl_params prm; double n_tests; uint32_t drm, phasing, min_reads, rh, generic, pad; } sig;
    memset(&sig, 0, sizeof sig);
    sig.alloc = ctx->alloc_version; sig.plan = ctx->plan_version; sig.prm = *prm; sig.n_tests = n_tests;
    sig.drm = drm_masks != nullptr; sig.phasing = phasing != 0; sig.min_reads = min_reads; sig.rh = want_read_hap != 0; sig.generic = (ctx->phase_generic ? 1u : 0u) | (ctx->phase_two ? 2u : 0u); sig.pad = (uint32_t)(uintptr_t)ctx->read_hap_out;
    static const bool graphs_on = !getenv("JL_NO_GRAPH");   // read once; eager launches are a debugging aid
    bool launched = false;
    // A graph replay reaches the queue 10-16 us after the call, a plain launch 3-5 us (MI355X guide, graph-replay-floor);
    // the remaining launches of an eager run are enqueued while the first kernel runs.  For a window whose counting
    // stage alone takes hundreds of microseconds the replay's head start is all a graph changes: such runs go eagerly.
    const bool long_run = (uint64_t)ctx->col_stride * ctx->n_cols >= ((uint64_t)512 << 20);

    if (graphs_on && !long_run) {
        const bool hit = ctx->graph_exec && ctx->graph_sig.size() == sizeof sig && memcmp(ctx->graph_sig.data(), &sig, sizeof sig) == 0;
        // A configuration is captured the SECOND time it is run: capture + instantiation cost about 20 ms, which a
        // one-shot caller (the juliet front end: one run per process) would pay for nothing.
        const bool seen = ctx->graph_seen.size() == sizeof sig && memcmp(ctx->graph_seen.data(), &sig, sizeof sig) == 0;
        if (!hit) {
            if (ctx->graph_exec) { hipGraphExecDestroy(ctx->graph_exec); ctx->graph_exec = nullptr; }
            if (ctx->graph) { hipGraphDestroy(ctx->graph); ctx->graph = nullptr; }
            ctx->graph_sig.clear();
            if (!seen) ctx->graph_seen.assign((const uint8_t *)&sig, (const uint8_t *)&sig + sizeof sig);
            else if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                enqueue_path(ctx, prm, n_tests, drm_masks != nullptr, phasing != 0, min_reads, want_read_hap != 0);
                hipGraph_t g = nullptr;
                if (hipStreamEndCapture(ctx->stream, &g) == hipSuccess && g &&
                    hipGraphInstantiate(&ctx->graph_exec, g, nullptr, nullptr, 0) == hipSuccess) {
                    ctx->graph = g;
                    ctx->graph_sig.assign((const uint8_t *)&sig, (const uint8_t *)&sig + sizeof sig);
                } else {
                    if (g) hipGraphDestroy(g);
                    ctx->graph_exec = nullptr;
                }
            }
            (void)hipGetLastError();
        }
        if (ctx->graph_exec && hipGraphLaunch(ctx->graph_exec, ctx->stream) == hipSuccess) launched = true;
    }
    if (!launched) {
        enqueue_path(ctx, prm, n_tests, drm_masks != nullptr, phasing != 0, min_reads, want_read_hap != 0);
        JL_HIP(ctx, hipGetLastError());
    }
    ctx->runs_launched++;
    ctx->clock_run = ctx->pileup_clock ? ctx->runs_launched : 0u;   // (the run whose pileup the two clock nodes bracket)
    ctx->pileup_done = ctx->call_done = true;
    ctx->phase_done = phasing != 0;
    ctx->pack_valid = true;
    ctx->run_read_hap = phasing && want_read_hap;
    return JL_OK;
}

// Spin on the pinned sequence word of the last run (see done_kernel).  A device fault would leave it unset for
// ever, so after a long wait the stream is asked directly.
int jl_run_wait_impl(jl_ctx *ctx)
{
    int rc = jl_run_wait_seq(ctx, ctx->runs_launched);
    if (rc == JL_OK && ctx->ing_check_pending) rc = jl_ingest_verdict(ctx);   // the run read a matrix an enqueued ingest made
    // a result block without its magic behind a phasing run: the folded launch timed out (see jl_phase_rerun_unfolded)
    if (rc == JL_OK && ctx->pack_valid && ctx->phase_done && ctx->h_pack && ctx->h_pack->magic != JL_PACK_MAGIC && !ctx->no_fold)
        rc = jl_phase_rerun_unfolded(ctx);
    return rc;
}

// `want`: the value of runs_launched right after the run of interest was launched.  1: the stream failed, 2: it went
// idle without the word.  Touches nothing of the context but the pinned word: any thread may wait.
static int run_wait_word(volatile uint32_t *p, uint32_t want, hipStream_t stream, hipError_t *err)
{
    uint64_t spins = 0;
    while ((int32_t)(*p - want) < 0) {
        __builtin_ia32_pause();
        if ((++spins & 0x3FFFFFu) == 0) {
            // A long wait (tens of ms): a device fault would leave the word unset for ever, so ask the stream.  A
            // blocking synchronize, not a query: the first launch of a freshly instantiated graph can sit in the
            // runtime for milliseconds before it reaches the queue, during which a query calls the stream idle.
            *err = hipStreamSynchronize(stream);
            if (*err != hipSuccess) return 1;
            for (int k = 0; k < 1000000 && (int32_t)(*p - want) < 0; ++k) __builtin_ia32_pause();
            if ((int32_t)(*p - want) < 0) return 2;
            break;
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return 0;
}

int jl_phase_rerun_unfolded(jl_ctx *ctx)
{
    hipStream_t st = ctx->run_stream ? ctx->run_stream : ctx->stream;
    JL_HIP(ctx, hipSetDevice(ctx->device));
    JL_HIP(ctx, hipStreamSynchronize(st));
    uint32_t ovf = 0;
    JL_HIP(ctx, hipMemcpyAsync(&ovf, &ctx->d_meta->overflow, 4, hipMemcpyDeviceToHost, st));
    JL_HIP(ctx, hipStreamSynchronize(st));
    if (!(ovf & 32u)) return jl_fail(ctx, JL_ERR_DEVICE, "the run's result block was not written (no folded launch timed out)");
    ctx->no_fold = true;
    ctx->fold_reruns++;
    ctx->alloc_version++;   // captured graphs of this context folded
    // the Fisher stage's masks and rows are resident: the plan comes out of them again, the ids from phase_assign_kernel,
    // the completion word from a node of its own (only the fused launches fold, so the run is one of theirs).  The run
    // counters are the Fisher launch's to zero (call_kernel): here a memset stands in for it.
    JL_HIP(ctx, hipMemsetAsync(ctx->d_meta, 0, sizeof(jl_phase_meta), st));
    jl_launch_phase(ctx, st, ctx->last_min_reads, true, true, false);
    jl_launch_done_on(ctx, st);
    JL_HIP(ctx, hipGetLastError());
    ctx->runs_launched++;
    int rc = jl_run_wait_seq(ctx, ctx->runs_launched);
    if (rc) return rc;
    if (ctx->h_pack->magic != JL_PACK_MAGIC) return jl_fail(ctx, JL_ERR_DEVICE, "the unfolded phase launch left no result block either");
    return JL_OK;
}

int jl_run_wait_seq(jl_ctx *ctx, uint32_t want)
{
    hipError_t e = hipSuccess;
    const int r = run_wait_word(ctx->h_seq, want, ctx->run_stream ? ctx->run_stream : ctx->stream, &e);
    if (r == 1) return jl_fail(ctx, JL_ERR_DEVICE, "run failed: %s", hipGetErrorString(e));
    if (r == 2) return jl_fail(ctx, JL_ERR_DEVICE, "run finished without its completion word (%u of %u)", *ctx->h_seq, want);
    return JL_OK;
}
}  // extern "C"

// for threads other than the context's owner (the communicator's worker): no write to ctx->err, and the stream the run
// was enqueued on is the one the requesting thread saw
int jl_run_wait_seq_quiet(jl_ctx *ctx, uint32_t want, hipStream_t stream)
{
    hipError_t e = hipSuccess;
    return run_wait_word(ctx->h_seq, want, stream, &e) ? JL_ERR_DEVICE : JL_OK;
}

extern "C" {

#ifdef JL_TUNING
// tuning aid, not part of the ABI header (tools_tuning/timeline.py): the device-clock stamps of the last
// JL_TIMELINE_ROWS runs of this context (JL_TIMELINE=1), 100 MHz ticks
__attribute__((visibility("default"))) int jl_debug_timeline(jl_ctx *ctx, uint64_t *out)
{
    if (!ctx || !out) return JL_ERR_ARG;
    if (!ctx->d_timeline) return jl_fail(ctx, JL_ERR_STATE, "run with JL_TIMELINE=1");
    JL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    JL_HIP(ctx, hipMemcpy(out, ctx->d_timeline, (size_t)JL_TIMELINE_ROWS * JL_TIMELINE_SLOTS * 8, hipMemcpyDeviceToHost));
    return JL_OK;
}
#endif

int jl_run_wait(jl_ctx *ctx)
{
    if (!ctx) return JL_ERR_ARG;
    if (!ctx->pack_valid) return jl_fail(ctx, JL_ERR_STATE, "jl_run_wait needs jl_run_async first");
    return jl_run_wait_impl(ctx);
}

int jl_run_done(jl_ctx *ctx)
{
    if (!ctx || !ctx->pack_valid) return 0;
    if ((int32_t)(*ctx->h_seq - ctx->runs_launched) < 0) return 0;
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return 1;
}

int jl_expand_read_hap(const void *packed, uint32_t bits, uint64_t n_reads, uint16_t *out)
{
    if (!packed || !out || (bits != 4 && bits != 8 && bits != 16)) return JL_ERR_ARG;
    jl_expand_ids(packed, bits, n_reads, out);
    return JL_OK;
}

int jl_run_view_get(jl_ctx *ctx, jl_run_view *out)
{
    if (!ctx || !out) return JL_ERR_ARG;
    if (!ctx->pack_valid) return jl_fail(ctx, JL_ERR_STATE, "jl_run_view_get needs jl_run_async first");
    int rc = jl_run_wait_impl(ctx);
    if (rc) return rc;
    const jl_pack *pk = ctx->h_pack;
    memset(out, 0, sizeof *out);
    if (pk->magic != JL_PACK_MAGIC) return jl_fail(ctx, JL_ERR_STATE, "result block not written");
    out->n_variants = pk->nvar_total;
    out->phased = pk->phase_ran;
    out->n_reads = ctx->n_reads;
    out->variants = pk->variants;
    bool ok = pk->fits_call != 0;
    if (pk->phase_ran) {
        ok = ok && pk->fits_phase;
        out->n_positions = pk->vp;
        out->n_haplotypes = pk->H;
        out->n_var_phase = pk->nv_phase;
        out->summary = pk->summary;
        out->pos_cols = pk->pos_cols;
        out->hap_count = pk->hap_count;
        out->hap_pattern = pk->hap_pattern;
        out->hit = pk->hit;
        out->cooc = pk->cooc_fits ? pk->cooc : nullptr;
        if (ctx->run_read_hap) {
            out->read_hap_packed = ctx->h_read_hap;
            out->read_hap_bits = pk->id_bits;
            out->read_hap = pk->id_bits == 16 ? ctx->h_read_hap : nullptr;
        }
    }
    out->complete = ok ? 1u : 0u;
    return JL_OK;
}

/* ---------------------------------------------------------------- numerics self-check */

int jl_fisher_eval(jl_ctx *ctx, const uint32_t *a, const uint32_t *c, const uint32_t *cov, uint32_t n, double *p,
                   double *log_p)
{
    return jl_fisher_eval_tail(ctx, a, c, cov, n, 0, p, log_p);
}

int jl_fisher_eval_tail(jl_ctx *ctx, const uint32_t *a, const uint32_t *c, const uint32_t *cov, uint32_t n, int tail, double *p,
                        double *log_p)
{
    if (!ctx || !a || !c || !cov || !p || !log_p || (tail != 0 && tail != 1)) return JL_ERR_ARG;
    if (n == 0) return JL_OK;
    for (uint32_t i = 0; i < n; ++i)
        if (a[i] > cov[i] || c[i] > cov[i]) return jl_fail(ctx, JL_ERR_ARG, "table %u: a and c must be <= cov", i);
    JL_HIP(ctx, hipSetDevice(ctx->device));
    uint32_t *d_in = nullptr;
    double *d_out = nullptr;
    JL_HIP(ctx, hipMalloc(&d_in, (size_t)n * 12));
    if (hipMalloc(&d_out, (size_t)n * 16) != hipSuccess) { hipFree(d_in); return jl_fail(ctx, JL_ERR_MEMORY, "fisher_eval buffers"); }
    hipStream_t st = ctx->stream;
    hipError_t e = hipMemcpyAsync(d_in, a, (size_t)n * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_in + n, c, (size_t)n * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_in + 2 * (size_t)n, cov, (size_t)n * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        jl_launch_fisher_eval(ctx, n, d_in, d_in + n, d_in + 2 * (size_t)n, tail, d_out, d_out + n);
        e = hipMemcpyAsync(p, d_out, (size_t)n * 8, hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(log_p, d_out + n, (size_t)n * 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    hipFree(d_in);
    hipFree(d_out);
    if (e != hipSuccess) return jl_fail(ctx, JL_ERR_DEVICE, "fisher_eval: %s", hipGetErrorString(e));
    return JL_OK;
}

/* ---------------------------------------------------------------- timing */

int jl_run_pileup_clock(jl_ctx *ctx, int on)
{
    if (!ctx) return JL_ERR_ARG;
    if (ctx->pileup_clock != (on != 0)) {
        ctx->pileup_clock = on != 0;
        ctx->alloc_version++;      // (the captured graphs of this context do not have / have the two nodes)
    }
    return JL_OK;
}

int jl_run_pileup_ms(jl_ctx *ctx, float *ms, uint64_t *begin_ticks)
{
    if (!ctx || !ms) return JL_ERR_ARG;
    if (!ctx->pileup_clock) return jl_fail(ctx, JL_ERR_STATE, "jl_run_pileup_ms: the clock is off (jl_run_pileup_clock)");
    if (!ctx->pack_valid) return jl_fail(ctx, JL_ERR_STATE, "jl_run_pileup_ms needs a run");
    // the two stamps belong to the last jl_run_async with the clock on; a group run, a run of another path or one enqueued
    // before the clock was switched on carries no clock nodes: its stamps would be an older run's
    if (ctx->clock_run == 0u || ctx->clock_run != ctx->runs_launched)
        return jl_fail(ctx, JL_ERR_STATE, "jl_run_pileup_ms: the context's last run carried no clock nodes (only jl_run_async runs with the clock on do; group runs do not)");
    if (int rc = jl_run_wait_impl(ctx)) return rc;
    const volatile unsigned long long *t = reinterpret_cast<const volatile unsigned long long *>(const_cast<uint32_t *>(ctx->h_seq) + 8);
    *ms = (float)((double)(t[1] - t[0]) * 1e-5);      // 100 MHz ticks
    if (begin_ticks) *begin_ticks = t[0];
    return JL_OK;
}

// Latency of the whole path at THIS boundary: `reps` runs one after the other, each launched when the one before has put
// its results into the pinned block and the caller has looked at them (jl_run_async -> jl_run_wait -> jl_run_view_get);
// host clock around the loop.  What a C or C++ caller sees; a Python caller adds its interpreter on top (bench.py
// reports both).
int jl_time_run(jl_ctx *ctx, const jl_gene *genes, uint32_t n_genes, const uint8_t *refseq, uint32_t ref_len,
                const jl_params *prm, const uint64_t *drm_masks, int phasing, uint32_t min_reads, int want_read_hap,
                uint32_t reps, float *ms_avg)
{
    if (!ctx || !ms_avg || reps == 0) return JL_ERR_ARG;
    jl_run_view v;
    for (int warm = 0; warm < 2; ++warm) {      // (the second run of a configuration captures its graph)
        if (int rc = jl_run_async(ctx, genes, n_genes, refseq, ref_len, prm, drm_masks, phasing, min_reads, want_read_hap)) return rc;
        if (int rc = jl_run_view_get(ctx, &v)) return rc;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t r = 0; r < reps; ++r) {
        if (int rc = jl_run_async(ctx, genes, n_genes, refseq, ref_len, prm, drm_masks, phasing, min_reads, want_read_hap)) return rc;
        if (int rc = jl_run_view_get(ctx, &v)) return rc;     // (waits for the completion word)
    }
    *ms_avg = (float)(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / (double)reps);
    return JL_OK;
}

int jl_time_pileup(jl_ctx *ctx, uint32_t reps, float *ms_avg)
{
    if (!ctx || !ms_avg || reps == 0) return JL_ERR_ARG;
    if (!ctx->plan_valid) return jl_fail(ctx, JL_ERR_STATE, "jl_time_pileup needs a plan: call jl_pileup_async once first");
    JL_HIP(ctx, hipSetDevice(ctx->device));
    float total = 0.f;
    if (!jl_pileup_needs_zero(ctx)) {
        // nothing but the kernel goes on the stream: one pair of events around `reps` back-to-back launches, so the
        // events' own latency (~3 us around a single launch) is not charged to the kernel; the figure includes the
        // gaps between consecutive launches and agrees with rocprofv3's per-dispatch average to ~1 us
        JL_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
        for (uint32_t r = 0; r < reps; ++r) jl_launch_pileup(ctx, ctx->stream);
        JL_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
        JL_HIP(ctx, hipEventSynchronize(ctx->ev1));
        JL_HIP(ctx, hipEventElapsedTime(&total, ctx->ev0, ctx->ev1));
    } else {
        for (uint32_t r = 0; r < reps; ++r) {
            // the counters are cleared outside the timed interval; only the kernel sits between the events
            JL_HIP(ctx, hipMemsetAsync(ctx->d_counts, 0, ctx->counts_words * sizeof(uint32_t), ctx->stream));
            JL_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
            jl_launch_pileup(ctx, ctx->stream);
            JL_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
            JL_HIP(ctx, hipEventSynchronize(ctx->ev1));
            float ms = 0.f;
            JL_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
            total += ms;
        }
    }
    *ms_avg = total / (float)reps;
    ctx->pileup_done = true;
    ctx->call_done = ctx->phase_done = false;
    return JL_OK;
}

// The same measurement over several resident windows in rotation (all launches on the first context's stream): with
// four 150 MB windows no launch finds its input in the 256 MiB Infinity Cache, as in the pipelined bench loop.
int jl_time_pileup_set(jl_ctx *const *ctxs, uint32_t n_ctx, uint32_t reps, float *ms_avg)
{
    if (!ctxs || n_ctx == 0 || !ms_avg || reps == 0) return JL_ERR_ARG;
    jl_ctx *c0 = ctxs[0];
    if (!c0) return JL_ERR_ARG;
    for (uint32_t k = 0; k < n_ctx; ++k) {
        if (!ctxs[k] || ctxs[k]->device != c0->device) return jl_fail(c0, JL_ERR_ARG, "contexts must share a device");
        if (!ctxs[k]->plan_valid) return jl_fail(c0, JL_ERR_STATE, "jl_time_pileup_set needs a plan on every context");
        if (jl_pileup_needs_zero(ctxs[k])) return jl_fail(c0, JL_ERR_STATE, "this launch shape needs zeroed counters: use jl_time_pileup");
    }
    JL_HIP(c0, hipSetDevice(c0->device));
    for (uint32_t k = 0; k < n_ctx; ++k) JL_HIP(c0, hipStreamSynchronize(ctxs[k]->stream));
    for (uint32_t k = 0; k < n_ctx; ++k) jl_launch_pileup(ctxs[k], c0->stream);   // warm-up, one per window
    JL_HIP(c0, hipEventRecord(c0->ev0, c0->stream));
    for (uint32_t r = 0; r < reps; ++r) jl_launch_pileup(ctxs[r % n_ctx], c0->stream);
    JL_HIP(c0, hipEventRecord(c0->ev1, c0->stream));
    JL_HIP(c0, hipEventSynchronize(c0->ev1));
    float total = 0.f;
    JL_HIP(c0, hipEventElapsedTime(&total, c0->ev0, c0->ev1));
    *ms_avg = total / (float)reps;
    for (uint32_t k = 0; k < n_ctx; ++k) {
        ctxs[k]->pileup_done = true;
        ctxs[k]->call_done = ctxs[k]->phase_done = false;
        ctxs[k]->pack_valid = false;
    }
    return JL_OK;
}

}  // extern "C"
