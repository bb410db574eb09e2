// capi_comm.hip — the multi-GPU part of the C ABI: RCCL communicator and the all-gather of the variant
// table.  One communicator per (rank, device); the asynchronous exchanges run on the communicator's own stream and are
// issued by its worker thread; the blocking cross-window exchanges (capi_xwin.hip) take the communicator for the
// duration of the call (jl_comm_direct_begin).
#include <stddef.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <new>

#include "jl_comm_internal.h"

#ifndef JL_COMM_TIMEOUT_S
#define JL_COMM_TIMEOUT_S 60
#endif

// fixed-stride table (+ row counts) over RCCL/xGMI, on the communicator's stream; blocks the worker until the rows are
// in the caller's arrays
static void comm_full_gather(jl_comm *c, jl_comm_full *f)
{
    jl_ctx *ctx = f->ctx;
    int st = JL_OK;
    if (f->wait_seq && jl_run_wait_seq_quiet(ctx, f->wait_seq, f->run_stream) != JL_OK) st = JL_ERR_DEVICE;
    {
        // The collective is issued whatever happened on this rank: the peers have issued theirs and would wait for ever.
        // A failed rank sends an impossible row count, so every rank takes the same error branch.
        const void *cnt_src = st == JL_OK ? (const void *)ctx->d_nvar : (const void *)c->d_poison;
        ncclResult_t r = ncclGroupStart();
        if (r == ncclSuccess) r = ncclAllGather(ctx->d_variants, c->d_all, sizeof(jl_variant) * (size_t)f->cap_rows, ncclUint8, c->comm, c->stream);
        if (r == ncclSuccess) r = ncclAllGather(cnt_src, c->d_counts, 8, ncclUint8, c->comm, c->stream);
        if (r == ncclSuccess) r = ncclGroupEnd();
        if (r != ncclSuccess && st == JL_OK) st = JL_ERR_COMM;
    }
    std::vector<uint32_t> cnt(2 * (size_t)c->world);
    if (st == JL_OK &&
        (hipMemcpyAsync(f->all_rows, c->d_all, sizeof(jl_variant) * (size_t)f->cap_rows * c->world, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
         hipMemcpyAsync(cnt.data(), c->d_counts, 8 * (size_t)c->world, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
         hipStreamSynchronize(c->stream) != hipSuccess))
        st = JL_ERR_DEVICE;
    if (st == JL_OK)
        for (int k = 0; k < c->world; ++k) {
            f->all_counts[k] = cnt[2 * k];
            if (cnt[2 * k] == 0xFFFFFFFFu) st = JL_ERR_COMM;   // that rank's run failed
            else if (cnt[2 * k] > f->cap_rows && st == JL_OK) st = JL_ERR_OVERFLOW;
        }
    f->status = st;
}

static void comm_worker(jl_comm *c)
{
    hipSetDevice(c->device);
    for (;;) {
        jl_comm_job job;
        {
            std::unique_lock<std::mutex> lk(c->mu);
            // a blocking call of another thread may hold the communicator (jl_comm_direct_begin): nothing starts meanwhile
            c->cv.wait(lk, [&] { return (c->stop || !c->queue.empty()) && !c->direct_busy; });
            if (c->queue.empty()) return;  // stop requested and nothing left
            job = std::move(c->queue.front());
            c->queue.pop_front();
            c->worker_busy = true;
        }
        if (job.full) {
            comm_full_gather(c, job.full);
            {
                std::lock_guard<std::mutex> lk(c->mu);
                job.full->done = true;
                c->worker_busy = false;
            }
            c->cv.notify_all();
            continue;
        }
        std::vector<jl_comm_slot *> &batch = job.batch;
        int st = JL_OK;
        // the producing runs are complete when their sequence words are in pinned memory (jl_run_wait): no events
        bool run_ok = true;
        for (jl_comm_slot *s : batch)
            if (jl_run_wait_seq_quiet(s->ctx, s->run_seq, s->run_stream) != JL_OK) run_ok = false;
        if (!run_ok) st = JL_ERR_DEVICE;
        // A batch whose slots are consecutive in the arena (jl_allgather_variants_async_many reserves them so) is ONE
        // exchange: a small kernel puts the heads of its windows next to each other, one all-gather moves them
        // ([rank][window][head] on every rank), one copy brings them to pinned memory, one event says so.  Otherwise
        // an all-gather, a copy and an event per window.
        // Whatever failed on THIS rank before the collective, the collective is issued: the peers have issued theirs.
        // The heads of a failed rank are zeros (no magic word), which every rank reads as that rank's failure.
        const size_t stride = JL_PACK_HEAD_BYTES * (size_t)c->world;
        bool consecutive = batch.size() > 1 && batch.size() <= JL_GATHER_MAX;
        for (size_t k = 1; k < batch.size(); ++k)
            if (batch[k]->d_heads != batch[k - 1]->d_heads + stride) consecutive = false;
        if (consecutive) {
            const uint32_t n = (uint32_t)batch.size();
            const size_t i0 = (size_t)(batch[0] - c->slots);
            uint8_t *send = c->d_send + i0 * JL_PACK_HEAD_BYTES;
            const uint8_t *srcs[JL_GATHER_MAX];
            for (uint32_t k = 0; k < n; ++k) srcs[k] = batch[k]->d_src;
            jl_comm_slot *last = batch.back();
            const uint8_t *sendbuf = send;
            if (st == JL_OK) {
                jl_launch_gather_heads(srcs, n, send, c->stream);
                if (hipGetLastError() != hipSuccess) st = JL_ERR_DEVICE;
            }
            if (st != JL_OK) sendbuf = c->d_zero;
            if (ncclAllGather(sendbuf, batch[0]->d_heads, JL_PACK_HEAD_BYTES * (size_t)n, ncclUint8, c->comm, c->stream) != ncclSuccess && st == JL_OK) st = JL_ERR_COMM;
            if (hipMemcpyAsync(batch[0]->h_heads, batch[0]->d_heads, stride * n, hipMemcpyDeviceToHost, c->stream) != hipSuccess && st == JL_OK) st = JL_ERR_DEVICE;
            if (hipEventRecord(last->done, c->stream) != hipSuccess && st == JL_OK) st = JL_ERR_DEVICE;
            for (uint32_t k = 0; k < n; ++k) {
                batch[k]->done_at = last;
                batch[k]->h_base = batch[0]->h_heads;
                batch[k]->batch_n = n;
                batch[k]->batch_k = k;
            }
        } else {
            for (jl_comm_slot *s : batch) {
                const uint8_t *sendbuf = st == JL_OK ? s->d_src : c->d_zero;
                if (ncclAllGather(sendbuf, s->d_heads, JL_PACK_HEAD_BYTES, ncclUint8, c->comm, c->stream) != ncclSuccess && st == JL_OK) st = JL_ERR_COMM;
                if (hipMemcpyAsync(s->h_heads, s->d_heads, stride, hipMemcpyDeviceToHost, c->stream) != hipSuccess && st == JL_OK) st = JL_ERR_DEVICE;
                if (hipEventRecord(s->done, c->stream) != hipSuccess && st == JL_OK) st = JL_ERR_DEVICE;
                s->done_at = s;
                s->h_base = s->h_heads;
                s->batch_n = 1;
                s->batch_k = 0;
            }
        }
        {
            std::lock_guard<std::mutex> lk(c->mu);
            for (jl_comm_slot *s : batch) {
                s->status = st;
                s->enqueued = true;
            }
            c->worker_busy = false;
        }
        c->cv.notify_all();
    }
}

int jl_comm_direct_begin(jl_comm *c)
{
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->direct_busy || c->worker_busy || !c->queue.empty()) return JL_ERR_STATE;
    for (const jl_comm_slot &s : c->slots)
        if (s.pending) return JL_ERR_STATE;
    c->direct_busy = true;
    return JL_OK;
}

void jl_comm_direct_end(jl_comm *c)
{
    {
        std::lock_guard<std::mutex> lk(c->mu);
        c->direct_busy = false;
    }
    c->cv.notify_all();
}

static void comm_wait_enqueued(jl_comm *c, jl_comm_slot *s)
{
    std::unique_lock<std::mutex> lk(c->mu);
    c->cv.wait(lk, [&] { return s->enqueued; });
}


extern "C" {

/* ---------------------------------------------------------------- multi-GPU */



int jl_comm_unique_id(uint8_t id[128])
{
    static_assert(sizeof(ncclUniqueId) == 128, "RCCL unique id is 128 bytes");
    ncclUniqueId u;
    if (ncclGetUniqueId(&u) != ncclSuccess) return JL_ERR_COMM;
    memcpy(id, &u, 128);
    return JL_OK;
}

int jl_comm_create(jl_ctx *ctx, const uint8_t id[128], int rank, int world, jl_comm **out)
{
    if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world) return JL_ERR_ARG;
    JL_HIP(ctx, hipSetDevice(ctx->device));
    jl_comm *c = new (std::nothrow) jl_comm();
    if (!c) return JL_ERR_MEMORY;
    c->rank = rank;
    c->world = world;
    c->device = ctx->device;
    ncclUniqueId u;
    memcpy(&u, id, 128);
    ncclResult_t r = ncclCommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) {
        delete c;
        return jl_fail(ctx, JL_ERR_COMM, "ncclCommInitRank: %s", ncclGetErrorString(r));
    }
    const size_t stride = JL_PACK_HEAD_BYTES * (size_t)world;
    bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess &&
              hipMalloc(&c->d_all, sizeof(jl_variant) * JL_VARIANT_CAP * world) == hipSuccess &&
              hipMalloc(&c->d_counts, 8 * world) == hipSuccess &&
              hipMalloc(&c->d_arena, stride * JL_COMM_SLOTS) == hipSuccess &&
              hipMalloc(&c->d_send, (size_t)JL_PACK_HEAD_BYTES * JL_COMM_SLOTS) == hipSuccess &&
              hipMalloc(&c->d_zero, (size_t)JL_PACK_HEAD_BYTES * JL_GATHER_MAX) == hipSuccess &&
              hipMalloc(&c->d_poison, 8) == hipSuccess &&
              hipMemset(c->d_zero, 0, (size_t)JL_PACK_HEAD_BYTES * JL_GATHER_MAX) == hipSuccess &&
              hipMemset(c->d_poison, 0xFF, 8) == hipSuccess &&
              hipHostMalloc(&c->h_arena, stride * JL_COMM_SLOTS, hipHostMallocDefault) == hipSuccess;
    for (int k = 0; ok && k < JL_COMM_SLOTS; ++k) {
        c->slots[k].d_heads = c->d_arena + stride * (size_t)k;
        c->slots[k].h_heads = c->h_arena + stride * (size_t)k;
        ok = hipEventCreateWithFlags(&c->slots[k].done, hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) {
        jl_comm_destroy(c);
        return jl_fail(ctx, JL_ERR_MEMORY, "comm buffers");
    }
    c->worker = std::thread(comm_worker, c);
    *out = c;
    return JL_OK;
}

void jl_comm_destroy(jl_comm *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    if (c->worker.joinable()) {
        {
            std::lock_guard<std::mutex> lk(c->mu);
            c->stop = true;
        }
        c->cv.notify_all();
        c->worker.join();
    }
    if (c->stream) hipStreamSynchronize(c->stream);
    for (jl_comm_slot &s : c->slots)
        if (s.done) hipEventDestroy(s.done);
    if (c->d_arena) hipFree(c->d_arena);
    if (c->d_send) hipFree(c->d_send);
    if (c->d_zero) hipFree(c->d_zero);
    if (c->d_poison) hipFree(c->d_poison);
    if (c->h_arena) hipHostFree(c->h_arena);
    if (c->comm) ncclCommDestroy(c->comm);
    if (c->d_all) hipFree(c->d_all);
    if (c->d_counts) hipFree(c->d_counts);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

// A context may have several exchanges in flight, each in its own slot.  Slots are not tied to contexts: a slot is free
// when no uncollected BATCH uses it; `n` asks for the first of n consecutive free slots (the exchanges of a batch then
// land next to each other in the arena); `oldest` = a context's uncollected slot with the smallest sequence number.
static jl_comm_slot *comm_slot_free(jl_comm *c, uint32_t n = 1)
{
    uint32_t run = 0;
    for (uint32_t k = 0; k < JL_COMM_SLOTS; ++k) {
        run = c->slots[k].pending ? 0 : run + 1;
        if (run == n) return &c->slots[k + 1 - n];
    }
    return nullptr;
}

static jl_comm_slot *comm_slot_oldest(jl_ctx *ctx, jl_comm *c)
{
    jl_comm_slot *best = nullptr;
    for (jl_comm_slot &s : c->slots)
        if (s.ctx == ctx && s.pending && !s.collected && (!best || s.seq < best->seq)) best = &s;
    return best;
}

// A member of a batch was collected: its slot (its bytes of the batch's ONE result region, the batch's event) stays
// reserved until every member is, then the whole run of slots is free again.
static void comm_slot_release(jl_comm_slot *s)
{
    s->collected = true;
    jl_comm_slot *lead = s->leader ? s->leader : s;
    if (lead->batch_left > 0) --lead->batch_left;
    if (lead->batch_left == 0)
        for (uint32_t k = 0; k < lead->batch_size; ++k) lead[k].pending = false;
}

// Enqueue-only half: after jl_run_async, all-gather the head of the run's device result block (header + up to 128
// rows = 6.2 KB per rank) on the communicator's stream.  No HIP call here: the request goes to the worker thread,
// which waits for the run's completion word and then issues the collective.  The result block is double-buffered
// by run parity, so the run that follows on this context does not disturb the exchange; a context can therefore
// have TWO exchanges pending, a third is refused.
static int comm_request(jl_ctx *ctx, jl_comm *c, jl_comm_slot *at, jl_comm_slot **out)
{
    if (!ctx->pack_valid) return jl_fail(ctx, JL_ERR_STATE, "jl_allgather_variants_async needs jl_run_async first");
    if (ctx->device != c->device) return jl_fail(ctx, JL_ERR_ARG, "context and communicator are on different devices");
    if (ctx->exch_runs.size() >= 2) return jl_fail(ctx, JL_ERR_STATE, "two exchanges of this context are pending: collect one first");
    for (uint32_t r : ctx->exch_runs)
        if (r == ctx->runs_launched) return jl_fail(ctx, JL_ERR_STATE, "this run's exchange was already requested");
    jl_comm_slot *s = at ? at : comm_slot_free(c);
    if (!s || s->pending) return jl_fail(ctx, JL_ERR_MEMORY, "no free exchange slot (%d per communicator): collect pending exchanges first", JL_COMM_SLOTS);
    s->ctx = ctx;
    s->done_at = s;
    s->leader = s;
    s->batch_left = 1;
    s->batch_size = 1;
    s->collected = false;
    s->run_seq = ctx->runs_launched;
    s->run_stream = ctx->run_stream ? ctx->run_stream : ctx->stream;
    s->d_src = reinterpret_cast<const uint8_t *>(ctx->d_pack + ((ctx->runs_launched - 1u) & 1u));
    s->seq = c->next_seq++;
    s->event_seen = false;
    s->enqueued = false;   // not yet visible to the worker: no lock needed
    s->status = JL_OK;
    ctx->exch_runs.push_back(s->run_seq);
    ctx->exch_pending = (uint32_t)ctx->exch_runs.size();
    *out = s;
    return JL_OK;
}

static void comm_forget_run(jl_ctx *ctx, uint32_t run_seq)
{
    for (size_t k = 0; k < ctx->exch_runs.size(); ++k)
        if (ctx->exch_runs[k] == run_seq) { ctx->exch_runs.erase(ctx->exch_runs.begin() + (long)k); break; }
    ctx->exch_pending = (uint32_t)ctx->exch_runs.size();
}

int jl_allgather_variants_async(jl_ctx *ctx, jl_comm *c)
{
    if (!ctx || !c) return JL_ERR_ARG;
    jl_comm_slot *s = nullptr;
    int rc = comm_request(ctx, c, nullptr, &s);
    if (rc) return rc;
    s->pending = true;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        jl_comm_job job;
        job.batch.assign(1, s);
        c->queue.push_back(std::move(job));
    }
    c->cv.notify_all();
    return JL_OK;
}

// The exchanges of several contexts (the windows of one group run) as ONE all-gather: one collective launch instead
// of n.  Every rank must pass the same number of contexts in the same call order.
int jl_allgather_variants_async_many(jl_ctx *const *ctxs, uint32_t n, jl_comm *c)
{
    if (!ctxs || !c || n == 0 || n > JL_GATHER_MAX) return JL_ERR_ARG;
    std::vector<jl_comm_slot *> batch;
    // n consecutive slots: the batch is then ONE all-gather.  Which protocol a batch uses must not depend on what
    // happens to be free (every rank has to issue the same collectives), so there is no fall-back to n all-gathers.
    jl_comm_slot *run0 = comm_slot_free(c, n);
    if (!run0) return jl_fail(ctxs[0], JL_ERR_MEMORY, "no %u consecutive free exchange slots (%d per communicator): collect pending exchanges first", n, JL_COMM_SLOTS);
    for (uint32_t k = 0; k < n; ++k) {
        if (!ctxs[k]) return JL_ERR_ARG;
        jl_comm_slot *s = nullptr;
        int rc = comm_request(ctxs[k], c, run0 + k, &s);
        if (rc) {
            for (jl_comm_slot *b : batch) { b->pending = false; comm_forget_run(b->ctx, b->run_seq); }
            return rc;
        }
        s->pending = true;   // reserves the slot for the following comm_slot_free calls
        s->leader = run0;
        batch.push_back(s);
    }
    run0->batch_left = n;
    run0->batch_size = n;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        jl_comm_job job;
        job.batch = batch;
        c->queue.push_back(std::move(job));
    }
    c->cv.notify_all();
    return JL_OK;
}

// The full fixed stride, through the worker's FIFO like every other collective of the communicator (every rank reaches
// this at the same point of its program, so the job takes the same place in every rank's queue).
static int allgather_full(jl_ctx *ctx, jl_comm *c, jl_variant *all_rows, uint32_t *all_counts, uint32_t cap_rows, uint32_t wait_seq)
{
    jl_comm_full f;
    f.ctx = ctx;
    f.cap_rows = cap_rows;
    f.wait_seq = wait_seq;
    f.run_stream = ctx->run_stream ? ctx->run_stream : ctx->stream;
    f.all_rows = all_rows;
    f.all_counts = all_counts;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        jl_comm_job job;
        job.full = &f;
        c->queue.push_back(std::move(job));
    }
    c->cv.notify_all();
    {
        std::unique_lock<std::mutex> lk(c->mu);
        c->cv.wait(lk, [&] { return f.done; });
    }
    if (f.status == JL_ERR_OVERFLOW) return jl_fail(ctx, JL_ERR_OVERFLOW, "a rank produced more than %u variant rows", cap_rows);
    if (f.status == JL_ERR_COMM) return jl_fail(ctx, JL_ERR_COMM, "the exchange failed: RCCL error, or a rank's run did not complete (it says so in its own error)");
    if (f.status != JL_OK) return jl_fail(ctx, f.status, "full-stride all-gather failed on the communicator thread");
    return JL_OK;
}

// The one collective of the path.  After jl_run_async the exchange is the 6.2 KB head of each rank's result
// block (enqueued here unless jl_allgather_variants_async already did); tables with more than 128 rows on any
// rank, or stage-by-stage callers, use the full fixed stride.  The decision is made from the gathered headers,
// so every rank takes the same branch.
int jl_allgather_variants(jl_ctx *ctx, jl_comm *c, jl_variant *all_rows, uint32_t *all_counts, uint32_t cap_rows)
{
    if (!ctx || !c || !all_rows || !all_counts) return JL_ERR_ARG;
    if (!ctx->call_done) return jl_fail(ctx, JL_ERR_STATE, "jl_allgather_variants before jl_call_async");
    if (cap_rows == 0 || cap_rows > JL_VARIANT_CAP) return jl_fail(ctx, JL_ERR_ARG, "cap_rows must be 1..%u", JL_VARIANT_CAP);
    JL_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->pack_valid) {
        // stage-by-stage caller: the table is final once the context's stream is idle
        JL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return allgather_full(ctx, c, all_rows, all_counts, cap_rows, 0);
    }
    jl_comm_slot *s = comm_slot_oldest(ctx, c);
    if (!s) {
        int rc = jl_allgather_variants_async(ctx, c);
        if (rc) return rc;
        s = comm_slot_oldest(ctx, c);
    }
    const uint32_t run_seq = s->run_seq;
    comm_forget_run(ctx, run_seq);
    comm_wait_enqueued(c, s);
    struct release { jl_comm_slot *s; ~release() { comm_slot_release(s); } } rel{s};   // free for reuse once the heads were read
    if (s->status != JL_OK) return jl_fail(ctx, s->status, "all-gather enqueue failed on the communicator thread");
    if (!s->done_at->event_seen) {   // spin on the event: a blocking hipEventSynchronize costs ~15 us of wake-up latency per step
        hipError_t q;
        // ... but not for ever: a peer that died before its collective leaves this one incomplete
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(JL_COMM_TIMEOUT_S);
        uint32_t polls = 0;
        while ((q = hipEventQuery(s->done_at->done)) == hipErrorNotReady)
            if ((++polls & 0xFFFu) == 0 && std::chrono::steady_clock::now() > deadline)
                return jl_fail(ctx, JL_ERR_COMM, "all-gather not complete after %d s: a peer has not issued its collective", JL_COMM_TIMEOUT_S);
        if (q != hipSuccess) return jl_fail(ctx, JL_ERR_DEVICE, "all-gather: %s", hipGetErrorString(q));
        s->done_at->event_seen = true;
    }
    bool compact = true;
    for (int k = 0; k < c->world; ++k) {
        const jl_pack *pk = reinterpret_cast<const jl_pack *>(s->h_base + ((size_t)k * s->batch_n + s->batch_k) * JL_PACK_HEAD_BYTES);
        // a head without the magic word is what a rank sends whose run failed (comm_worker): every rank sees it and
        // returns the same error; nobody goes on to a second collective
        if (pk->magic != JL_PACK_MAGIC) return jl_fail(ctx, JL_ERR_COMM, "rank %d's run did not complete: its head of the exchange is empty", k);
        if (!pk->fits_call) compact = false;
    }
    if (compact) {
        int rc = JL_OK;
        for (int k = 0; k < c->world; ++k) {
            const jl_pack *pk = reinterpret_cast<const jl_pack *>(s->h_base + ((size_t)k * s->batch_n + s->batch_k) * JL_PACK_HEAD_BYTES);
            all_counts[k] = pk->nvar_total;
            if (pk->nvar_total > cap_rows) { rc = JL_ERR_OVERFLOW; continue; }
            memcpy(all_rows + (size_t)k * cap_rows, pk->variants, (size_t)pk->nvar_total * sizeof(jl_variant));
        }
        if (rc) return jl_fail(ctx, rc, "a rank produced more than %u variant rows", cap_rows);
        return JL_OK;
    }
    // Some rank's table does not fit the 128-row head: the full stride, from the resident table.  That table is NOT
    // double-buffered — if this context has launched another run since, it holds that run's rows.  Every rank sees the
    // same headers and runs the same program, so every rank refuses here together.
    if (ctx->runs_launched != run_seq)
        return jl_fail(ctx, JL_ERR_STATE, "a rank called more than %u variants: such a run's exchange must be collected before the "
                                          "context's next run (the full table is not double-buffered)", JL_PACK_MAX_VAR);
    return allgather_full(ctx, c, all_rows, all_counts, cap_rows, run_seq);
}


// The exchanges of several contexts collected in one call (the windows of one launch): all_rows [n_ctx][world][cap_rows],
// all_counts [n_ctx][world].  Stops at the first failure.
int jl_allgather_variants_many(jl_ctx *const *ctxs, uint32_t n_ctx, jl_comm *c, jl_variant *all_rows, uint32_t *all_counts,
                               uint32_t cap_rows)
{
    if (!ctxs || !c || !all_rows || !all_counts) return JL_ERR_ARG;
    for (uint32_t k = 0; k < n_ctx; ++k) {
        const int rc = jl_allgather_variants(ctxs[k], c, all_rows + (size_t)k * c->world * cap_rows, all_counts + (size_t)k * c->world, cap_rows);
        if (rc) return rc;
    }
    return JL_OK;
}


}  // extern "C"
