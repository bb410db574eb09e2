// capi_comm.hip — the multi-GPU part of the C ABI: RCCL communicator, the all-gather of the variant
// table, and the cross-window column exchange.  One communicator per (rank, device); collectives run on the
// communicator's own stream and are issued by a worker thread.
#include <rccl/rccl.h>
#include <stddef.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "jl_internal.h"

#define JL_COMM_SLOTS 128

// One communicator per (rank, device).  Collectives run on the communicator's OWN stream, issued by a worker
// thread once the producing run's completion word has arrived in pinned memory, so that several contexts (batches
// in flight) never have an RCCL launch — nor any HIP call it would imply — on the thread that launches batches.
struct jl_comm_slot {
    jl_ctx *ctx = nullptr;
    const uint8_t *d_src = nullptr;  // this rank's contribution: the run's device result block (double-buffered by run parity)
    uint32_t run_seq = 0;        // the run whose results are exchanged: the worker waits for its completion word
    uint64_t seq = 0;            // enqueue order: jl_allgather_variants collects a context's OLDEST pending exchange
    uint8_t *d_heads = nullptr;  // [world][JL_PACK_HEAD_BYTES]: slot k's share of the communicator's arena
    uint8_t *h_heads = nullptr;  // pinned mirror (same layout: consecutive slots are consecutive in memory)
    hipEvent_t done = nullptr;
    jl_comm_slot *done_at = nullptr;   // the slot whose event covers this exchange (the last one of its batch)
    // where rank r's head of this exchange lies in pinned memory: h_base + (r * batch_n + batch_k) * JL_PACK_HEAD_BYTES
    const uint8_t *h_base = nullptr;
    uint32_t batch_n = 1, batch_k = 0;
    bool pending = false;        // the slot is reserved: requested, and its batch not yet collected completely (host thread only)
    bool collected = false;      // this member was collected; the slot stays reserved until the whole batch is (its
                                 // region of the arena is ONE [rank][window][head] block, its event the batch's)
    jl_comm_slot *leader = nullptr;   // first slot of the batch
    uint32_t batch_left = 0;     // leader only: members not yet collected
    uint32_t batch_size = 1;     // leader only
    bool event_seen = false;     // host thread only: `done` was seen complete (a batch's members share one event: the
                                 // first collector pays for the query, 5-10 us in the runtime, the others do not)
    bool enqueued = false;       // the worker has issued it and recorded `done` (guarded by jl_comm::mu)
    int status = 0;              // ncclResult_t / hip error of the enqueue, as jl_status
};

// the full-stride gather (tables of more than 128 rows, stage-by-stage callers): issued by the worker too — the
// communicator is never used from two threads
struct jl_comm_full {
    jl_ctx *ctx = nullptr;
    uint32_t cap_rows = 0;
    uint32_t wait_seq = 0;       // != 0: the run whose completion word the worker waits for first
    jl_variant *all_rows = nullptr;
    uint32_t *all_counts = nullptr;
    int status = 0;
    bool done = false;           // guarded by jl_comm::mu
};

struct jl_comm_job {
    std::vector<jl_comm_slot *> batch;
    jl_comm_full *full = nullptr;
};

struct jl_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;
    uint8_t *d_arena = nullptr, *h_arena = nullptr;   // [JL_COMM_SLOTS][world][JL_PACK_HEAD_BYTES]
    uint8_t *d_send = nullptr;                        // [JL_COMM_SLOTS][JL_PACK_HEAD_BYTES]: send buffers of batches
    jl_variant *d_all = nullptr;   // [world][JL_VARIANT_CAP]   (full-stride fallback)
    uint32_t *d_counts = nullptr;  // [world][2]
    jl_comm_slot slots[JL_COMM_SLOTS];
    uint64_t next_seq = 1;
    // RCCL enqueues cost the host ~20 us each; a worker thread issues them (FIFO, so every rank keeps the
    // same collective order) while the caller's thread goes on launching the next batch
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<jl_comm_job> queue;   // FIFO: batches (the exchanges of one batch go out as ONE all-gather) and full-stride gathers
    bool stop = false;
};

// fixed-stride table (+ row counts) over RCCL/xGMI, on the communicator's stream; blocks the worker until the rows are
// in the caller's arrays
static void comm_full_gather(jl_comm *c, jl_comm_full *f)
{
    jl_ctx *ctx = f->ctx;
    int st = JL_OK;
    if (f->wait_seq && jl_run_wait_seq(ctx, f->wait_seq) != JL_OK) st = JL_ERR_DEVICE;
    if (st == JL_OK) {
        ncclResult_t r = ncclGroupStart();
        if (r == ncclSuccess) r = ncclAllGather(ctx->d_variants, c->d_all, sizeof(jl_variant) * (size_t)f->cap_rows, ncclUint8, c->comm, c->stream);
        if (r == ncclSuccess) r = ncclAllGather(ctx->d_nvar, c->d_counts, 8, ncclUint8, c->comm, c->stream);
        if (r == ncclSuccess) r = ncclGroupEnd();
        if (r != ncclSuccess) st = JL_ERR_COMM;
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
            if (cnt[2 * k] > f->cap_rows) st = JL_ERR_OVERFLOW;
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
            c->cv.wait(lk, [&] { return c->stop || !c->queue.empty(); });
            if (c->queue.empty()) return;  // stop requested and nothing left
            job = std::move(c->queue.front());
            c->queue.pop_front();
        }
        if (job.full) {
            comm_full_gather(c, job.full);
            {
                std::lock_guard<std::mutex> lk(c->mu);
                job.full->done = true;
            }
            c->cv.notify_all();
            continue;
        }
        std::vector<jl_comm_slot *> &batch = job.batch;
        int st = JL_OK;
        // the producing runs are complete when their sequence words are in pinned memory (jl_run_wait): no events
        for (jl_comm_slot *s : batch)
            if (jl_run_wait_seq(s->ctx, s->run_seq) != JL_OK) st = JL_ERR_DEVICE;
        // A batch whose slots are consecutive in the arena (jl_allgather_variants_async_many reserves them so) is ONE
        // exchange: a small kernel puts the heads of its windows next to each other, one all-gather moves them
        // ([rank][window][head] on every rank), one copy brings them to pinned memory, one event says so.  Otherwise
        // an all-gather, a copy and an event per window.
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
            if (st == JL_OK) {
                jl_launch_gather_heads(srcs, n, send, c->stream);
                if (hipGetLastError() != hipSuccess) st = JL_ERR_DEVICE;
            }
            if (st == JL_OK && ncclAllGather(send, batch[0]->d_heads, JL_PACK_HEAD_BYTES * (size_t)n, ncclUint8, c->comm, c->stream) != ncclSuccess) st = JL_ERR_COMM;
            if (st == JL_OK && hipMemcpyAsync(batch[0]->h_heads, batch[0]->d_heads, stride * n, hipMemcpyDeviceToHost, c->stream) != hipSuccess) st = JL_ERR_DEVICE;
            if (hipEventRecord(last->done, c->stream) != hipSuccess && st == JL_OK) st = JL_ERR_DEVICE;
            for (uint32_t k = 0; k < n; ++k) {
                batch[k]->done_at = last;
                batch[k]->h_base = batch[0]->h_heads;
                batch[k]->batch_n = n;
                batch[k]->batch_k = k;
            }
        } else {
            for (jl_comm_slot *s : batch) {
                if (st == JL_OK && ncclAllGather(s->d_src, s->d_heads, JL_PACK_HEAD_BYTES, ncclUint8, c->comm, c->stream) != ncclSuccess) st = JL_ERR_COMM;
                if (st == JL_OK && hipMemcpyAsync(s->h_heads, s->d_heads, stride, hipMemcpyDeviceToHost, c->stream) != hipSuccess) st = JL_ERR_DEVICE;
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
        }
        c->cv.notify_all();
    }
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
        while ((q = hipEventQuery(s->done_at->done)) == hipErrorNotReady) {}
        if (q != hipSuccess) return jl_fail(ctx, JL_ERR_DEVICE, "all-gather: %s", hipGetErrorString(q));
        s->done_at->event_seen = true;
    }
    bool compact = true;
    for (int k = 0; k < c->world; ++k) {
        const jl_pack *pk = reinterpret_cast<const jl_pack *>(s->h_base + ((size_t)k * s->batch_n + s->batch_k) * JL_PACK_HEAD_BYTES);
        if (pk->magic != JL_PACK_MAGIC || !pk->fits_call) compact = false;
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


/* ---------------------------------------------------------------- cross-window phasing */

// Distinct variant positions of the merged (global-column) table, ascending, and the remapped table whose
// columns index the compact matrix: position k lives in compact columns 3k..3k+2.
static uint32_t xwin_remap(const jl_variant *merged, uint32_t n_var, jl_variant *remapped, uint32_t *pos_global)
{
    std::vector<uint32_t> cols;
    cols.reserve(n_var);
    for (uint32_t v = 0; v < n_var; ++v) cols.push_back(merged[v].col);
    std::sort(cols.begin(), cols.end());
    cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
    for (uint32_t v = 0; v < n_var; ++v) {
        const uint32_t k = (uint32_t)(std::lower_bound(cols.begin(), cols.end(), merged[v].col) - cols.begin());
        if (remapped) {
            remapped[v] = merged[v];
            remapped[v].col = 3u * k;
        }
    }
    if (pos_global) std::copy(cols.begin(), cols.end(), pos_global);
    return (uint32_t)cols.size();
}

static int xwin_owner(const uint32_t *win_begin, const uint32_t *win_ncols, uint32_t n_windows, uint32_t col)
{
    for (uint32_t w = 0; w < n_windows; ++w)
        if (col >= win_begin[w] && (uint64_t)col + 3 <= (uint64_t)win_begin[w] + win_ncols[w]) return (int)w;
    return -1;
}

// Host-only plan of the column exchange (no device, no communicator): the distinct variant positions of the merged
// table in ascending global order, the table remapped onto the compact matrix (position k -> columns 3k..3k+2), and
// for every position the window that holds its three columns entirely (-1: none does).
int jl_xwin_plan(const uint32_t *win_begin, const uint32_t *win_ncols, uint32_t n_windows, const jl_variant *merged, uint32_t n_var,
                 jl_variant *remapped, uint32_t *pos_global, int32_t *owner, uint32_t *vp_total)
{
    if (!win_begin || !win_ncols || n_windows == 0 || (!merged && n_var) || !vp_total) return JL_ERR_ARG;
    std::vector<uint32_t> pos(n_var ? n_var : 1);
    const uint32_t vp = xwin_remap(merged, n_var, remapped, pos.data());
    *vp_total = vp;
    for (uint32_t k = 0; k < vp; ++k) {
        if (pos_global) pos_global[k] = pos[k];
        if (owner) owner[k] = xwin_owner(win_begin, win_ncols, n_windows, pos[k]);
    }
    return JL_OK;
}

// All windows on THIS device (a 288 GB GPU holds many): device-to-device copies of 3 columns per position.
int jl_xwin_assemble_local(jl_ctx *pc, jl_ctx *const *windows, uint32_t n_windows, const jl_variant *merged, uint32_t n_var,
                           jl_variant *remapped, uint32_t *pos_global, uint32_t *vp_total)
{
    if (!pc || !windows || n_windows == 0 || (!merged && n_var) || !vp_total) return JL_ERR_ARG;
    if (n_var > JL_VARIANT_CAP) return jl_fail(pc, JL_ERR_OVERFLOW, "%u variants, table holds %u", n_var, JL_VARIANT_CAP);
    std::vector<uint32_t> wb(n_windows), wn(n_windows);
    for (uint32_t w = 0; w < n_windows; ++w) {
        if (!windows[w] || !windows[w]->d_msa) return jl_fail(pc, JL_ERR_ARG, "window %u has no resident matrix", w);
        if (windows[w]->n_reads != windows[0]->n_reads || windows[w]->col_stride != windows[0]->col_stride)
            return jl_fail(pc, JL_ERR_ARG, "windows must hold the same reads (window %u differs)", w);
        wb[w] = windows[w]->win_begin;
        wn[w] = windows[w]->n_cols;
    }
    std::vector<uint32_t> pos(n_var ? n_var : 1);
    const uint32_t vp = xwin_remap(merged, n_var, remapped, pos.data());
    *vp_total = vp;
    if (pos_global) std::copy(pos.begin(), pos.begin() + vp, pos_global);
    if (vp == 0) return JL_OK;
    const uint64_t stride = windows[0]->col_stride;
    int rc = jl_msa_alloc_strided(pc, windows[0]->n_reads, 3u * vp, stride, 0);   // the windows' stride, whatever it is
    if (rc) return rc;
    for (uint32_t k = 0; k < vp; ++k) {
        const int w = xwin_owner(wb.data(), wn.data(), n_windows, pos[k]);
        if (w < 0) return jl_fail(pc, JL_ERR_ARG, "variant column %u is not fully inside any window", pos[k]);
        JL_HIP(pc, hipStreamSynchronize(windows[w]->stream));
        JL_HIP(pc, hipMemcpyAsync(pc->d_msa + (uint64_t)3 * k * stride, windows[w]->d_msa + (uint64_t)(pos[k] - wb[w]) * stride,
                                  3 * stride, hipMemcpyDeviceToDevice, pc->stream));
    }
    JL_HIP(pc, hipStreamSynchronize(pc->stream));
    return JL_OK;
}

// One window per rank: the owner of each position broadcasts its 3 columns over RCCL/xGMI into every rank's
// compact matrix (the second exchange of a cross-window run; 3*Vp*col_stride bytes in total).
int jl_xwin_assemble_rccl(jl_ctx *pc, jl_ctx *window, jl_comm *c, const uint32_t *win_begin, const uint32_t *win_ncols,
                          const jl_variant *merged, uint32_t n_var, jl_variant *remapped, uint32_t *pos_global,
                          uint32_t *vp_total)
{
    if (!pc || !window || !c || !win_begin || !win_ncols || (!merged && n_var) || !vp_total) return JL_ERR_ARG;
    if (!window->d_msa) return jl_fail(pc, JL_ERR_ARG, "window has no resident matrix");
    if (n_var > JL_VARIANT_CAP) return jl_fail(pc, JL_ERR_OVERFLOW, "%u variants, table holds %u", n_var, JL_VARIANT_CAP);
    std::vector<uint32_t> pos(n_var ? n_var : 1);
    const uint32_t vp = xwin_remap(merged, n_var, remapped, pos.data());
    *vp_total = vp;
    if (pos_global) std::copy(pos.begin(), pos.begin() + vp, pos_global);
    if (vp == 0) return JL_OK;
    const uint64_t stride = window->col_stride;   // every rank's window must use the same stride (same reads)
    int rc = jl_msa_alloc_strided(pc, window->n_reads, 3u * vp, stride, 0);
    if (rc) return rc;
    JL_HIP(pc, hipStreamSynchronize(window->stream));
    ncclResult_t r = ncclGroupStart();
    for (uint32_t k = 0; k < vp && r == ncclSuccess; ++k) {
        const int w = xwin_owner(win_begin, win_ncols, (uint32_t)c->world, pos[k]);
        if (w < 0) { ncclGroupEnd(); return jl_fail(pc, JL_ERR_ARG, "variant column %u is not fully inside any window", pos[k]); }
        const uint8_t *src = (w == c->rank) ? window->d_msa + (uint64_t)(pos[k] - win_begin[w]) * stride : pc->d_msa;
        r = ncclBroadcast(src, pc->d_msa + (uint64_t)3 * k * stride, 3 * stride, ncclUint8, w, c->comm, pc->stream);
    }
    if (r == ncclSuccess) r = ncclGroupEnd();
    if (r != ncclSuccess) return jl_fail(pc, JL_ERR_COMM, "ncclBroadcast: %s", ncclGetErrorString(r));
    JL_HIP(pc, hipStreamSynchronize(pc->stream));
    return JL_OK;
}

// ---- the same with the READS sharded (SURVEY §8e option A): the compact matrix holds reads [read_begin,
// read_begin + n_slice) only, so a rank phases 1/world of the reads and the second exchange moves 1/world of the bytes.
// read_begin must be a multiple of 256 (a slice starts on a 128-byte line of every column).
static int xwin_slice_alloc(jl_ctx *pc, uint64_t n_slice, uint32_t vp)
{
    int rc = jl_msa_alloc(pc, n_slice, 3u * vp, 0);
    if (rc) return rc;
    // padding nibbles of a column (reads past the slice) are 'not covered'
    JL_HIP(pc, hipMemsetAsync(pc->d_msa, 0x66, (size_t)pc->col_stride * 3u * vp, pc->stream));
    return JL_OK;
}

int jl_xwin_assemble_slice_local(jl_ctx *pc, jl_ctx *const *windows, uint32_t n_windows, const jl_variant *merged, uint32_t n_var,
                                 uint64_t read_begin, uint64_t n_slice, jl_variant *remapped, uint32_t *pos_global,
                                 uint32_t *vp_total)
{
    if (!pc || !windows || n_windows == 0 || (!merged && n_var) || !vp_total) return JL_ERR_ARG;
    if (n_var > JL_VARIANT_CAP) return jl_fail(pc, JL_ERR_OVERFLOW, "%u variants, table holds %u", n_var, JL_VARIANT_CAP);
    std::vector<uint32_t> wb(n_windows), wn(n_windows);
    for (uint32_t w = 0; w < n_windows; ++w) {
        if (!windows[w] || !windows[w]->d_msa) return jl_fail(pc, JL_ERR_ARG, "window %u has no resident matrix", w);
        if (windows[w]->n_reads != windows[0]->n_reads || windows[w]->col_stride != windows[0]->col_stride)
            return jl_fail(pc, JL_ERR_ARG, "windows must hold the same reads (window %u differs)", w);
        wb[w] = windows[w]->win_begin;
        wn[w] = windows[w]->n_cols;
    }
    const uint64_t n_reads = windows[0]->n_reads;
    if ((n_slice && (read_begin & 255u)) || read_begin > n_reads || n_slice > n_reads - read_begin)
        return jl_fail(pc, JL_ERR_ARG, "slice [%llu, +%llu) of %llu reads: the start must be a multiple of 256 and the slice inside",
                       (unsigned long long)read_begin, (unsigned long long)n_slice, (unsigned long long)n_reads);
    std::vector<uint32_t> pos(n_var ? n_var : 1);
    const uint32_t vp = xwin_remap(merged, n_var, remapped, pos.data());
    *vp_total = vp;
    if (pos_global) std::copy(pos.begin(), pos.begin() + vp, pos_global);
    if (vp == 0 || n_slice == 0) return JL_OK;
    int rc = xwin_slice_alloc(pc, n_slice, vp);
    if (rc) return rc;
    const uint64_t src_stride = windows[0]->col_stride, bytes = (n_slice + 1u) / 2u;
    for (uint32_t k = 0; k < vp; ++k) {
        const int w = xwin_owner(wb.data(), wn.data(), n_windows, pos[k]);
        if (w < 0) return jl_fail(pc, JL_ERR_ARG, "variant column %u is not fully inside any window", pos[k]);
        JL_HIP(pc, hipStreamSynchronize(windows[w]->stream));
        JL_HIP(pc, hipMemcpy2DAsync(pc->d_msa + (uint64_t)3 * k * pc->col_stride, pc->col_stride,
                                    windows[w]->d_msa + (uint64_t)(pos[k] - wb[w]) * src_stride + read_begin / 2u, src_stride, bytes, 3,
                                    hipMemcpyDeviceToDevice, pc->stream));
    }
    JL_HIP(pc, hipStreamSynchronize(pc->stream));
    return JL_OK;
}

// One window per rank, reads sharded: rank s phases reads [slice_begin[s], slice_begin[s + 1]) (world + 1 entries, the
// same on every rank, multiples of 256 except the last = n_reads).  The owner of each variant position sends rank s
// that slice of its three columns (ncclSend / ncclRecv in one group; its own slice by a device copy).
int jl_xwin_assemble_slice_rccl(jl_ctx *pc, jl_ctx *window, jl_comm *c, const uint32_t *win_begin, const uint32_t *win_ncols,
                                const jl_variant *merged, uint32_t n_var, const uint64_t *slice_begin, jl_variant *remapped,
                                uint32_t *pos_global, uint32_t *vp_total)
{
    if (!pc || !window || !c || !win_begin || !win_ncols || (!merged && n_var) || !slice_begin || !vp_total) return JL_ERR_ARG;
    if (!window->d_msa) return jl_fail(pc, JL_ERR_ARG, "window has no resident matrix");
    if (n_var > JL_VARIANT_CAP) return jl_fail(pc, JL_ERR_OVERFLOW, "%u variants, table holds %u", n_var, JL_VARIANT_CAP);
    const int world = c->world, me = c->rank;
    for (int s = 0; s < world; ++s)
        if ((slice_begin[s + 1] > slice_begin[s] && (slice_begin[s] & 255u)) || slice_begin[s + 1] < slice_begin[s] ||
            slice_begin[s + 1] > window->n_reads)
            return jl_fail(pc, JL_ERR_ARG, "slice %d of the reads is not 256-aligned or not inside the %llu reads", s,
                           (unsigned long long)window->n_reads);
    std::vector<uint32_t> pos(n_var ? n_var : 1);
    const uint32_t vp = xwin_remap(merged, n_var, remapped, pos.data());
    *vp_total = vp;
    if (pos_global) std::copy(pos.begin(), pos.begin() + vp, pos_global);
    const uint64_t n_mine = slice_begin[me + 1] - slice_begin[me];
    if (vp == 0) return JL_OK;
    if (n_mine) {
        int rc = xwin_slice_alloc(pc, n_mine, vp);
        if (rc) return rc;
    }
    const uint64_t src_stride = window->col_stride;
    JL_HIP(pc, hipStreamSynchronize(window->stream));
    ncclResult_t r = ncclGroupStart();
    for (uint32_t k = 0; k < vp && r == ncclSuccess; ++k) {
        const int w = xwin_owner(win_begin, win_ncols, (uint32_t)world, pos[k]);
        if (w < 0) { ncclGroupEnd(); return jl_fail(pc, JL_ERR_ARG, "variant column %u is not fully inside any window", pos[k]); }
        for (uint32_t j = 0; j < 3u && r == ncclSuccess; ++j) {
            if (w == me) {
                const uint8_t *col = window->d_msa + (uint64_t)(pos[k] - win_begin[w] + j) * src_stride;
                for (int s = 0; s < world && r == ncclSuccess; ++s) {
                    const uint64_t bytes = (slice_begin[s + 1] - slice_begin[s] + 1u) / 2u;
                    if (!bytes) continue;
                    if (s == me) {
                        if (hipMemcpyAsync(pc->d_msa + (uint64_t)(3u * k + j) * pc->col_stride, col + slice_begin[s] / 2u, bytes,
                                           hipMemcpyDeviceToDevice, pc->stream) != hipSuccess) r = ncclUnhandledCudaError;
                    } else {
                        r = ncclSend(col + slice_begin[s] / 2u, bytes, ncclUint8, s, c->comm, pc->stream);
                    }
                }
            } else if (n_mine) {
                r = ncclRecv(pc->d_msa + (uint64_t)(3u * k + j) * pc->col_stride, (n_mine + 1u) / 2u, ncclUint8, w, c->comm, pc->stream);
            }
        }
    }
    if (r == ncclSuccess) r = ncclGroupEnd();
    else ncclGroupEnd();
    if (r != ncclSuccess) return jl_fail(pc, JL_ERR_COMM, "column slices: %s", ncclGetErrorString(r));
    JL_HIP(pc, hipStreamSynchronize(pc->stream));
    return JL_OK;
}

}  // extern "C"
