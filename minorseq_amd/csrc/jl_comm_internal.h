// jl_comm_internal.h — the communicator of libjuliet_hip.so (capi_comm.hip), shared with the cross-window session
// (capi_xwin.hip).  Private to the library.
#pragma once
#include <rccl/rccl.h>

#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
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
    hipStream_t run_stream = nullptr;   // the stream the producing run was enqueued on, as the requesting thread saw it
    bool enqueued = false;       // the worker has issued it and recorded `done` (guarded by jl_comm::mu)
    int status = 0;              // ncclResult_t / hip error of the enqueue, as jl_status
};

// the full-stride gather (tables of more than 128 rows, stage-by-stage callers): issued by the worker too — the
// communicator is never used from two threads
struct jl_comm_full {
    jl_ctx *ctx = nullptr;
    uint32_t cap_rows = 0;
    uint32_t wait_seq = 0;       // != 0: the run whose completion word the worker waits for first
    hipStream_t run_stream = nullptr;
    jl_variant *all_rows = nullptr;
    uint32_t *all_counts = nullptr;
    int status = 0;
    bool done = false;           // guarded by jl_comm::mu
};

struct jl_comm_job {
    std::vector<jl_comm_slot *> batch;
    jl_comm_full *full = nullptr;
};

struct jl_inproc_world;   // capi_comm.hip: the ranks of an in-process communicator (threads of this process)

struct jl_comm {
    ncclComm_t comm = nullptr;
    // In-process form (jl_comm_create_inproc): the ranks are threads of ONE process — a host that drives several devices,
    // or several ranks on one device — and every exchange is a set of device copies (same device, or peer devices over
    // xGMI) between two barriers of the rank threads.  No RCCL communicator exists then.
    jl_inproc_world *inproc = nullptr;
    hipEvent_t ready = nullptr;      // in-process form: "what this rank sends is ready" for the exchange at hand
    std::string tp_error;            // the transport's last failure, for the caller's message
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;
    uint8_t *d_arena = nullptr, *h_arena = nullptr;   // [JL_COMM_SLOTS][world][JL_PACK_HEAD_BYTES]
    uint8_t *d_send = nullptr;                        // [JL_COMM_SLOTS][JL_PACK_HEAD_BYTES]: send buffers of batches
    jl_variant *d_all = nullptr;   // [world][JL_VARIANT_CAP]   (full-stride fallback)
    uint32_t *d_counts = nullptr;  // [world][2]
    // what a rank sends when ITS part of an exchange failed before the collective: the collective is issued all the same
    // (a rank that skipped it would leave its peers waiting for ever) and carries the failure to every rank
    uint8_t *d_zero = nullptr;     // [JL_GATHER_MAX][JL_PACK_HEAD_BYTES] zeros: heads without the magic word
    uint32_t *d_poison = nullptr;  // [2] 0xFFFFFFFF: an impossible row count
    jl_comm_slot slots[JL_COMM_SLOTS];
    uint64_t next_seq = 1;
    // RCCL enqueues cost the host ~20 us each; a worker thread issues them (FIFO, so every rank keeps the
    // same collective order) while the caller's thread goes on launching the next batch
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<jl_comm_job> queue;   // FIFO: batches (the exchanges of one batch go out as ONE all-gather) and full-stride gathers
    bool stop = false;
    bool worker_busy = false;    // the worker is issuing a job (guarded by mu)
    bool direct_busy = false;    // a blocking call of another thread has taken the communicator (guarded by mu)
    int host_gather = -1;        // may a collective work in pinned host memory?  -1: not tried yet (jl_comm_host_gather)
};

// A communicator is used by ONE thread at a time.  Asynchronous exchanges are issued by the worker thread; a blocking call
// that issues its collectives itself (the cross-window exchanges) first takes the communicator: refused with
// JL_ERR_STATE while the worker has anything queued, running or uncollected — every rank runs the same program, so every
// rank is refused alike — and the worker starts nothing until it is given back.
int jl_comm_direct_begin(jl_comm *c);
void jl_comm_direct_end(jl_comm *c);
// The same for a caller whose collective belongs BEHIND whatever the worker still has to issue (the exchange that a group
// launch carries, jl_group_exchange_bind): waits until the worker's queue is empty, then takes the communicator.
void jl_comm_direct_wait_begin(jl_comm *c);
// May an all-gather of this communicator work in place in pinned host memory (one device operation per exchange, nothing
// to copy back)?  Tried once — a collective: every rank calls this at the same point of its program — on a 64-byte pattern
// per rank, and agreed on through a second all-gather in device memory, so every rank gets the same answer.  1 / 0.
int jl_comm_host_gather(jl_comm *c);
// the full fixed stride of ctx's resident table through the worker (jl_allgather_variants' fall-back)
extern "C" int jl_comm_allgather_full(jl_ctx *ctx, jl_comm *c, jl_variant *all_rows, uint32_t *all_counts, uint32_t cap_rows, uint32_t wait_seq);

// The transport: the three exchange shapes the library uses, on stream `st`, over RCCL or in process.  0 or JL_ERR_COMM /
// JL_ERR_DEVICE (c->tp_error says what).  Stream-ordered with RCCL; the in-process form also blocks the calling thread
// until every rank has its data (two barriers of the rank threads), so on return the send buffers are free either way once
// the stream has passed the call.  Every rank makes the same calls in the same order (as RCCL demands).
struct jl_tp_msg {
    int peer;
    void *ptr;        // what to send / where to receive
    size_t bytes;
    bool send;
};
int jl_tp_allgather(jl_comm *c, const void *send, void *recv, size_t bytes, hipStream_t st);      // recv = [world][bytes]
int jl_tp_allgather2(jl_comm *c, const void *send_a, void *recv_a, size_t bytes_a, const void *send_b, void *recv_b, size_t bytes_b,
                     hipStream_t st);                                                            // two of them as one group
int jl_tp_exchange(jl_comm *c, const jl_tp_msg *msgs, size_t n, hipStream_t st);                  // sends and receives, one group
struct jl_tp_bcast {
    void *buf;        // the root's data / where the others receive it
    size_t bytes;
    int root;
};
int jl_tp_broadcasts(jl_comm *c, const jl_tp_bcast *b, size_t n, hipStream_t st);                 // one group
