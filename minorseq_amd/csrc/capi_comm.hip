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

#include <map>
#include <memory>

/* ---------------------------------------------------------------- the transport: RCCL, or device copies between rank threads */

// The ranks of an in-process communicator.  An exchange = every rank posts what it offers (pointers, an event that says
// "ready" on its stream), a barrier of the rank threads, every rank copies what is addressed to it on its own stream
// (same device: device-to-device; another device: hipMemcpyPeerAsync, i.e. xGMI), waits for its copies, a second barrier
// (only then may a sender reuse its buffers).  A rank that does not show up within JL_COMM_TIMEOUT_S fails the exchange on
// every rank (the world stays failed: the ranks' programs have diverged).
struct jl_inproc_world {
    int world = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    bool failed = false;
    struct post_t {
        int device = 0;
        hipEvent_t ready = nullptr;
        const void *send[2] = {nullptr, nullptr};       // all-gather(s)
        size_t bytes[2] = {0, 0};
        std::vector<jl_tp_msg> msgs;                    // exchange
        std::vector<jl_tp_bcast> bcasts;                // broadcasts
    };
    std::vector<post_t> post;
    int members = 0;   // communicators alive
};

namespace {
std::mutex g_inproc_mu;
std::map<std::string, std::shared_ptr<jl_inproc_world>> g_inproc;   // by the 128-byte id

int tp_fail(jl_comm *c, int status, const std::string &what)
{
    c->tp_error = what;
    return status;
}

bool inproc_barrier(jl_inproc_world *w)
{
    std::unique_lock<std::mutex> lk(w->mu);
    if (w->failed) return false;
    const uint64_t gen = w->generation;
    if (++w->arrived == w->world) {
        w->arrived = 0;
        ++w->generation;
        w->cv.notify_all();
        return true;
    }
    if (!w->cv.wait_for(lk, std::chrono::seconds(JL_COMM_TIMEOUT_S), [&] { return w->generation != gen || w->failed; })) {
        w->failed = true;   // a rank is missing: nobody may go on as if the exchange had happened
        w->cv.notify_all();
        return false;
    }
    return !w->failed;
}

hipError_t inproc_copy(jl_comm *c, void *dst, const void *src, size_t bytes, int src_device, hipStream_t st)
{
    if (!bytes) return hipSuccess;
    if (dst == src) return hipSuccess;   // an exchange in place: this rank's own part
    if (src_device == c->device) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, st);   // (either end may be pinned host memory)
    return hipMemcpyPeerAsync(dst, c->device, src, src_device, bytes, st);
}

// the frame of every in-process exchange: post (the caller filled its post_t), barrier, `copy_mine` on st, wait, barrier
template <class F>
int inproc_run(jl_comm *c, hipStream_t st, F copy_mine)
{
    jl_inproc_world *w = c->inproc;
    jl_inproc_world::post_t &me = w->post[(size_t)c->rank];
    me.device = c->device;
    me.ready = c->ready;
    hipError_t e = hipEventRecord(c->ready, st);
    const bool b1 = inproc_barrier(w);   // reached whatever happened locally: the peers are waiting in theirs
    if (!b1) return tp_fail(c, JL_ERR_COMM, "in-process exchange: a rank did not arrive (or the world has failed before)");
    if (e == hipSuccess) {
        for (int p = 0; p < w->world && e == hipSuccess; ++p)
            if (p != c->rank) e = hipStreamWaitEvent(st, w->post[(size_t)p].ready, 0);
        if (e == hipSuccess) e = copy_mine(w);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    const bool b2 = inproc_barrier(w);
    if (e != hipSuccess) return tp_fail(c, JL_ERR_DEVICE, std::string("in-process exchange: ") + hipGetErrorString(e));
    if (!b2) return tp_fail(c, JL_ERR_COMM, "in-process exchange: a rank did not finish");
    return JL_OK;
}
}  // namespace

int jl_tp_allgather2(jl_comm *c, const void *send_a, void *recv_a, size_t bytes_a, const void *send_b, void *recv_b, size_t bytes_b,
                     hipStream_t st)
{
    if (!c->inproc) {
        ncclResult_t r = ncclGroupStart();
        if (r == ncclSuccess) r = ncclAllGather(send_a, recv_a, bytes_a, ncclUint8, c->comm, st);
        if (r == ncclSuccess && send_b) r = ncclAllGather(send_b, recv_b, bytes_b, ncclUint8, c->comm, st);
        const ncclResult_t e = ncclGroupEnd();
        if (r == ncclSuccess) r = e;
        return r == ncclSuccess ? JL_OK : tp_fail(c, JL_ERR_COMM, std::string("ncclAllGather: ") + ncclGetErrorString(r));
    }
    jl_inproc_world::post_t &me = c->inproc->post[(size_t)c->rank];
    me.send[0] = send_a; me.bytes[0] = bytes_a;
    me.send[1] = send_b; me.bytes[1] = send_b ? bytes_b : 0;
    return inproc_run(c, st, [&](jl_inproc_world *w) {
        hipError_t e = hipSuccess;
        for (int p = 0; p < w->world && e == hipSuccess; ++p) {
            const jl_inproc_world::post_t &q = w->post[(size_t)p];
            if (q.bytes[0] != bytes_a || q.bytes[1] != (send_b ? bytes_b : 0)) return hipErrorInvalidValue;   // the ranks disagree
            e = inproc_copy(c, (uint8_t *)recv_a + (size_t)p * bytes_a, q.send[0], bytes_a, q.device, st);
            if (e == hipSuccess && send_b) e = inproc_copy(c, (uint8_t *)recv_b + (size_t)p * bytes_b, q.send[1], bytes_b, q.device, st);
        }
        return e;
    });
}

int jl_tp_allgather(jl_comm *c, const void *send, void *recv, size_t bytes, hipStream_t st)
{
    return jl_tp_allgather2(c, send, recv, bytes, nullptr, nullptr, 0, st);
}

int jl_tp_exchange(jl_comm *c, const jl_tp_msg *msgs, size_t n, hipStream_t st)
{
    if (!c->inproc) {
        ncclResult_t r = ncclGroupStart();
        for (size_t i = 0; i < n && r == ncclSuccess; ++i)
            r = msgs[i].send ? ncclSend(msgs[i].ptr, msgs[i].bytes, ncclUint8, msgs[i].peer, c->comm, st)
                             : ncclRecv(msgs[i].ptr, msgs[i].bytes, ncclUint8, msgs[i].peer, c->comm, st);
        const ncclResult_t e = ncclGroupEnd();
        if (r == ncclSuccess) r = e;
        return r == ncclSuccess ? JL_OK : tp_fail(c, JL_ERR_COMM, std::string("ncclSend / ncclRecv: ") + ncclGetErrorString(r));
    }
    c->inproc->post[(size_t)c->rank].msgs.assign(msgs, msgs + n);
    return inproc_run(c, st, [&](jl_inproc_world *w) {
        // my i-th receive from peer p takes p's i-th send to me: messages of a pair match in list order, as in RCCL
        std::vector<size_t> next((size_t)w->world, 0);
        for (size_t i = 0; i < n; ++i) {
            if (msgs[i].send) continue;
            const int p = msgs[i].peer;
            if (p < 0 || p >= w->world) return hipErrorInvalidValue;
            const jl_inproc_world::post_t &q = w->post[(size_t)p];
            size_t &k = next[(size_t)p];
            while (k < q.msgs.size() && !(q.msgs[k].send && q.msgs[k].peer == c->rank)) ++k;
            if (k == q.msgs.size() || q.msgs[k].bytes != msgs[i].bytes) return hipErrorInvalidValue;   // no send for this receive
            const hipError_t e = inproc_copy(c, msgs[i].ptr, q.msgs[k].ptr, msgs[i].bytes, q.device, st);
            if (e != hipSuccess) return e;
            ++k;
        }
        return hipSuccess;
    });
}

int jl_tp_broadcasts(jl_comm *c, const jl_tp_bcast *b, size_t n, hipStream_t st)
{
    if (!c->inproc) {
        ncclResult_t r = ncclGroupStart();
        for (size_t i = 0; i < n && r == ncclSuccess; ++i) r = ncclBroadcast(b[i].buf, b[i].buf, b[i].bytes, ncclUint8, b[i].root, c->comm, st);
        const ncclResult_t e = ncclGroupEnd();
        if (r == ncclSuccess) r = e;
        return r == ncclSuccess ? JL_OK : tp_fail(c, JL_ERR_COMM, std::string("ncclBroadcast: ") + ncclGetErrorString(r));
    }
    c->inproc->post[(size_t)c->rank].bcasts.assign(b, b + n);
    return inproc_run(c, st, [&](jl_inproc_world *w) {
        for (size_t i = 0; i < n; ++i) {
            if (b[i].root == c->rank) continue;
            if (b[i].root < 0 || b[i].root >= w->world) return hipErrorInvalidValue;
            const jl_inproc_world::post_t &q = w->post[(size_t)b[i].root];
            if (i >= q.bcasts.size() || q.bcasts[i].bytes != b[i].bytes || q.bcasts[i].root != b[i].root) return hipErrorInvalidValue;
            const hipError_t e = inproc_copy(c, b[i].buf, q.bcasts[i].buf, b[i].bytes, q.device, st);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    });
}

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
        const int rc = jl_tp_allgather2(c, ctx->d_variants, c->d_all, sizeof(jl_variant) * (size_t)f->cap_rows, cnt_src, c->d_counts, 8, c->stream);
        if (rc != JL_OK && st == JL_OK) st = JL_ERR_COMM;
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
            if (jl_tp_allgather(c, sendbuf, batch[0]->d_heads, JL_PACK_HEAD_BYTES * (size_t)n, c->stream) != JL_OK && st == JL_OK) st = JL_ERR_COMM;
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
                if (jl_tp_allgather(c, sendbuf, s->d_heads, JL_PACK_HEAD_BYTES, c->stream) != JL_OK && st == JL_OK) st = JL_ERR_COMM;
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

void jl_comm_direct_wait_begin(jl_comm *c)
{
    std::unique_lock<std::mutex> lk(c->mu);
    c->cv.wait(lk, [&] { return !c->direct_busy && !c->worker_busy && c->queue.empty(); });
    c->direct_busy = true;
}

int jl_comm_host_gather(jl_comm *c)
{
    if (c->host_gather >= 0) return c->host_gather;
    static const char *forced = getenv("JL_EXCHANGE_STAGED");   // tuning / tests: take the staged form
    const size_t kPart = 64;
    uint8_t *t = nullptr;
    bool ok = !(forced && forced[0] == '1') && hipHostMalloc(&t, kPart * (size_t)c->world, hipHostMallocDefault) == hipSuccess;
    jl_comm_direct_wait_begin(c);
    if (ok) {
        memset(t, 0, kPart * (size_t)c->world);
        for (size_t i = 0; i < kPart; ++i) t[kPart * (size_t)c->rank + i] = (uint8_t)(0xA5u ^ (uint8_t)(31 * c->rank + (int)i));
        ok = jl_tp_allgather(c, t + kPart * (size_t)c->rank, t, kPart, c->stream) == JL_OK && hipStreamSynchronize(c->stream) == hipSuccess;
        for (int r = 0; ok && r < c->world; ++r)
            for (size_t i = 0; i < kPart; ++i)
                if (t[kPart * (size_t)r + i] != (uint8_t)(0xA5u ^ (uint8_t)(31 * r + (int)i))) ok = false;
    }
    else {
        // every rank issues the same collectives in the same order whatever happened to it: a rank that cannot try host memory
        // (no pinned block, or the staged form forced on it alone) runs the trial in device memory — the arena is idle here —
        // and says "not here" below; skipping the trial would pair its peers' 64-byte all-gather with the 8-byte one that follows
        (void)jl_tp_allgather(c, c->d_arena + kPart * (size_t)c->rank, c->d_arena, kPart, c->stream);
        (void)hipStreamSynchronize(c->stream);
    }
    (void)hipGetLastError();
    // the verdicts of all ranks, in device memory (zeros: fine, ones: not here)
    std::vector<uint32_t> v(2 * (size_t)c->world, 1u);
    bool agreed = jl_tp_allgather(c, ok ? (const void *)c->d_zero : (const void *)c->d_poison, c->d_counts, 8, c->stream) == JL_OK &&
                  hipMemcpyAsync(v.data(), c->d_counts, 8 * (size_t)c->world, hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
                  hipStreamSynchronize(c->stream) == hipSuccess;
    jl_comm_direct_end(c);
    for (uint32_t x : v) agreed = agreed && x == 0u;
    if (t) hipHostFree(t);
    c->host_gather = agreed ? 1 : 0;
    return c->host_gather;
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

static int comm_create(jl_ctx *ctx, const uint8_t id[128], int rank, int world, bool inproc, jl_comm **out)
{
    if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world) return JL_ERR_ARG;
    *out = nullptr;
    JL_HIP(ctx, hipSetDevice(ctx->device));
    jl_comm *c = new (std::nothrow) jl_comm();
    if (!c) return JL_ERR_MEMORY;
    c->rank = rank;
    c->world = world;
    c->device = ctx->device;
    if (inproc) {
        std::shared_ptr<jl_inproc_world> w;
        {
            std::lock_guard<std::mutex> lk(g_inproc_mu);
            std::shared_ptr<jl_inproc_world> &slot = g_inproc[std::string((const char *)id, 128)];
            if (!slot) {
                slot = std::make_shared<jl_inproc_world>();
                slot->world = world;
                slot->post.resize((size_t)world);
            }
            w = slot;
            if (w->world != world || w->post[(size_t)rank].ready) {   // (ready: the rank is taken)
                delete c;
                return jl_fail(ctx, JL_ERR_ARG, "in-process communicator: rank %d of %d does not fit the ranks that joined under this id", rank, world);
            }
            ++w->members;
        }
        c->inproc = w.get();
        if (hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) != hipSuccess) {
            jl_comm_destroy(c);
            return jl_fail(ctx, JL_ERR_DEVICE, "hipEventCreate failed");
        }
        {
            std::lock_guard<std::mutex> lk(g_inproc_mu);
            w->post[(size_t)rank].ready = c->ready;
            w->post[(size_t)rank].device = c->device;
        }
        // peers on other devices are copied from directly
        // (enabled lazily by the runtime for hipMemcpyPeerAsync; nothing to do here)
    } else {
        ncclUniqueId u;
        memcpy(&u, id, 128);
        ncclResult_t r = ncclCommInitRank(&c->comm, world, u, rank);
        if (r != ncclSuccess) {
            delete c;
            return jl_fail(ctx, JL_ERR_COMM, "ncclCommInitRank: %s", ncclGetErrorString(r));
        }
    }
    const size_t stride = JL_PACK_HEAD_BYTES * (size_t)world;
    bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess &&
              hipMalloc(&c->d_all, sizeof(jl_variant) * JL_VARIANT_CAP * world) == hipSuccess &&
              hipMalloc(&c->d_counts, 8 * world) == hipSuccess &&
              hipMalloc(&c->d_arena, stride * JL_COMM_SLOTS) == hipSuccess &&
              hipMalloc(&c->d_send, (size_t)JL_PACK_HEAD_BYTES * JL_COMM_SLOTS) == hipSuccess &&
              hipMalloc(&c->d_zero, (size_t)JL_PACK_HEAD_BYTES * JL_GATHER_MAX) == hipSuccess &&
              hipMalloc(&c->d_poison, 8) == hipSuccess &&
              hipMemsetAsync(c->d_zero, 0, (size_t)JL_PACK_HEAD_BYTES * JL_GATHER_MAX, c->stream) == hipSuccess &&      // (the communicator's own stream: see capi_group.hip)
              hipMemsetAsync(c->d_poison, 0xFF, 8, c->stream) == hipSuccess &&
              hipStreamSynchronize(c->stream) == hipSuccess &&
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

int jl_comm_create(jl_ctx *ctx, const uint8_t id[128], int rank, int world, jl_comm **out) { return comm_create(ctx, id, rank, world, false, out); }

int jl_comm_info(jl_comm *c, int *rccl_ranks, int *rank, int *world, int *exchange_form)
{
    if (!c) return JL_ERR_ARG;
    int n = 0;
    if (c->comm && ncclCommCount(c->comm, &n) != ncclSuccess) return JL_ERR_COMM;
    if (rccl_ranks) *rccl_ranks = n;
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (exchange_form) *exchange_form = c->host_gather;
    return JL_OK;
}

int jl_comm_create_inproc(jl_ctx *ctx, const uint8_t id[128], int rank, int world, jl_comm **out)
{
    return comm_create(ctx, id, rank, world, true, out);
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
    if (c->inproc) {
        std::lock_guard<std::mutex> lk(g_inproc_mu);
        c->inproc->post[(size_t)c->rank].ready = nullptr;
        if (--c->inproc->members == 0)
            for (auto it = g_inproc.begin(); it != g_inproc.end(); ++it)
                if (it->second.get() == c->inproc) { g_inproc.erase(it); break; }
    }
    if (c->ready) hipEventDestroy(c->ready);
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

int jl_comm_allgather_full(jl_ctx *ctx, jl_comm *c, jl_variant *all_rows, uint32_t *all_counts, uint32_t cap_rows, uint32_t wait_seq)
{
    return allgather_full(ctx, c, all_rows, all_counts, cap_rows, wait_seq);
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
