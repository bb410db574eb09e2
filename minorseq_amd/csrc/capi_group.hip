// capi_group.hip — group runs: the whole path for SEVERAL resident windows in a few launches.
//
// Why.  A 3 kb x 100k-read window is a 150 MB stream: too short to hide a launch's ramp and drain (one pileup launch
// reaches 0.61 of the HBM peak, a 1.2 GB stream 0.77), and its phasing stage is a chain of dependent memory round
// trips that keeps a hardware queue busy for tens of microseconds while doing almost nothing.  A group run gives each
// stage ONE launch for up to JL_GROUP_MAX windows (blockIdx.z = window, per-window argument blocks by value):
//   counting            pileup_planes_group_kernel (up to JL_GROUP_WINDOWS_MAX windows)
//   Fisher              call_group_kernel
//   phasing             phase_group_run_kernel (plan out of the call masks, keys, grouping, selection, result block)
//   per-read ids        phase_assign_group_kernel (small groups fold them into the phasing launch)
//   completion words    done_group_kernel
// A group of more windows is cut into chunks of JL_GROUP_MAX that are PIPELINED inside one captured graph: the
// pileups of consecutive chunks run back to back on the group's stream, each chunk's latency-bound tail runs on a
// side stream next to the following chunk's pileup, so only the last (smallest) chunk's tail is ever exposed.
//
// Results are per window, exactly those of jl_run_async on each context: every context keeps its own result block,
// per-read ids and completion word, so jl_run_wait / jl_run_view_get / jl_call_fetch / jl_phase_fetch work unchanged.
#include <string.h>

#include <algorithm>
#include <new>
#include <string>
#include <vector>

#include <chrono>

#include "jl_comm_internal.h"
#include "jl_internal.h"

#define JL_GROUP_SIDE_STREAMS 2
#ifndef JL_COMM_TIMEOUT_S
#define JL_COMM_TIMEOUT_S 60
#endif

struct jl_group {
    int device = -1;
    hipStream_t stream = nullptr;
    hipStream_t side[JL_GROUP_SIDE_STREAMS] = {nullptr, nullptr};
    hipEvent_t ev_end = nullptr;
    hipEvent_t ev_fork[JL_GROUP_WINDOWS_MAX / JL_GROUP_MAX] = {};
    hipEvent_t ev_join[JL_GROUP_SIDE_STREAMS] = {};
    std::vector<jl_ctx *> ctxs;
    jl_done_ent *d_done = nullptr;
    std::vector<jl_win_pileup> h_pile;
    std::vector<jl_win_call> h_call;
    std::vector<jl_win_fold> h_fold;      // the Fisher stage as the pileup launch's epilogue (every chunk is counted by one workgroup)
    std::vector<jl_win_compact> h_compact;
    std::vector<jl_win_phase> h_phase;
    struct chunk_t { uint32_t first, n, max_chunks, max_call_blocks, max_phase_blocks; bool fold; };
    std::vector<chunk_t> chunks;
    bool phasing = true;
    // one captured graph per parity of a bound exchange (the heads' destination is baked in); unbound: [0] only
    hipGraph_t graph[2] = {nullptr, nullptr};
    hipGraphExec_t graph_exec[2] = {nullptr, nullptr};
    bool graph_tried[2] = {false, false};
    std::vector<uint8_t> sig;   // everything the captured graph and the tables bake in
    std::string err;
    // ---- bound exchange (jl_group_exchange_bind): every run carries the all-gather of its windows' table heads
    jl_comm *xc = nullptr;
    bool x_staged = false;           // the all-gather works in device memory and is copied to the pinned region behind it
    uint8_t *x_host = nullptr;       // pinned [2][world][n][JL_PACK_HEAD_BYTES]: by run parity, [rank][window][head] each
    uint8_t *x_dev = nullptr;        // staged form: the same in device memory
    hipEvent_t x_done[2] = {nullptr, nullptr};
    bool x_pending[2] = {false, false};
    uint32_t x_run_seq[2][JL_GROUP_WINDOWS_MAX];   // each window's run number when the exchange went out
    uint64_t x_launched = 0, x_collected = 0;

    size_t x_part() const { return (size_t)JL_PACK_HEAD_BYTES * ctxs.size(); }                  // one rank's heads
    size_t x_region() const { return x_part() * (size_t)(xc ? xc->world : 1); }                 // one exchange
    uint8_t *x_work(uint32_t par) const { return (x_staged ? x_dev : x_host) + x_region() * par; }   // where the collective works
};

static void group_drop_graphs(jl_group *g)
{
    for (int p = 0; p < 2; ++p) {
        if (g->graph_exec[p]) { hipGraphExecDestroy(g->graph_exec[p]); g->graph_exec[p] = nullptr; }
        if (g->graph[p]) { hipGraphDestroy(g->graph[p]); g->graph[p] = nullptr; }
        g->graph_tried[p] = false;
    }
}

static int group_fail(jl_group *g, int status, const char *msg)
{
    if (g) g->err = msg;
    return status;
}

// the latency-bound stages of one chunk, on `st`
static void chunk_tail(jl_group *g, const jl_group::chunk_t &c, hipStream_t st)
{
    // (the Fisher stage ran in the pileup launch's epilogue: kernels_pileup.hip FOLD)
    if (!jl_fold_enabled()) jl_launch_call_group(g->h_call.data() + c.first, c.n, c.max_call_blocks, st);
    if (!g->phasing) {
        jl_launch_compact_group(g->h_compact.data() + c.first, c.n, st);
    } else {
        jl_launch_phase_group(g->h_phase.data() + c.first, c.n, c.max_phase_blocks, st);
        if (!c.fold) {
            bool to_host = false;
            for (uint32_t k = 0; k < c.n; ++k) to_host = to_host || g->ctxs[c.first + k]->read_hap_out != nullptr;
            jl_launch_assign_group(g->h_phase.data() + c.first, c.n, c.max_phase_blocks, to_host, st);
        }
    }
    // completion words of the chunk's windows, behind the end of its last stage (see enqueue_path in capi.hip)
    jl_launch_done_group(g->d_done + c.first, c.n, st);
}

static int group_enqueue(jl_group *g)
{
    for (jl_ctx *c : g->ctxs)
        if (!c->have_ref) jl_launch_guess(c, g->stream);   // majority-codon mode: seed bases per window
    const size_t nc = g->chunks.size();
    bool side_used[JL_GROUP_SIDE_STREAMS] = {false, false};
    for (size_t k = 0; k < nc; ++k) {
        const jl_group::chunk_t &c = g->chunks[k];
        int rc = jl_fold_enabled() ? jl_launch_pileup_fold_group(g->ctxs.data() + c.first, c.n, g->h_pile.data() + c.first, g->h_fold.data() + c.first, c.max_chunks, g->stream)
                                   : jl_launch_pileup_group(g->ctxs.data() + c.first, c.n, g->h_pile.data() + c.first, c.max_chunks, g->stream);
        if (rc) return rc;
        if (k + 1 < nc) {   // the tail runs beside the next chunk's pileup
            hipStream_t st = g->side[k % JL_GROUP_SIDE_STREAMS];
            if (hipEventRecord(g->ev_fork[k], g->stream) != hipSuccess || hipStreamWaitEvent(st, g->ev_fork[k], 0) != hipSuccess)
                return JL_ERR_DEVICE;
            chunk_tail(g, c, st);
            side_used[k % JL_GROUP_SIDE_STREAMS] = true;
        } else {
            chunk_tail(g, c, g->stream);
        }
    }
    for (int i = 0; i < JL_GROUP_SIDE_STREAMS; ++i)
        if (side_used[i] && (hipEventRecord(g->ev_join[i], g->side[i]) != hipSuccess ||
                             hipStreamWaitEvent(g->stream, g->ev_join[i], 0) != hipSuccess))
            return JL_ERR_DEVICE;
    return JL_OK;
}

extern "C" {

int jl_group_create(jl_ctx *const *ctxs, uint32_t n_ctx, jl_group **out)
{
    if (!ctxs || !out || n_ctx == 0 || n_ctx > JL_GROUP_WINDOWS_MAX) return JL_ERR_ARG;
    *out = nullptr;
    for (uint32_t k = 0; k < n_ctx; ++k) {
        if (!ctxs[k] || ctxs[k]->device != ctxs[0]->device) return JL_ERR_ARG;
        for (uint32_t j = 0; j < k; ++j)
            if (ctxs[j] == ctxs[k]) return JL_ERR_ARG;
    }
    jl_group *g = new (std::nothrow) jl_group();
    if (!g) return JL_ERR_MEMORY;
    g->device = ctxs[0]->device;
    g->ctxs.assign(ctxs, ctxs + n_ctx);
    g->h_pile.resize(n_ctx);
    g->h_call.resize(n_ctx);
    g->h_fold.resize(n_ctx);
    g->h_compact.resize(n_ctx);
    g->h_phase.resize(n_ctx);
    bool ok = hipSetDevice(g->device) == hipSuccess &&
              hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&g->ev_end, hipEventDisableTiming) == hipSuccess &&
              hipMalloc(&g->d_done, sizeof(jl_done_ent) * n_ctx) == hipSuccess;
    if (ok && n_ctx > JL_GROUP_MAX) {   // pipelined chunks: side streams for the tails
        for (int i = 0; ok && i < JL_GROUP_SIDE_STREAMS; ++i)
            ok = hipStreamCreateWithFlags(&g->side[i], hipStreamNonBlocking) == hipSuccess &&
                 hipEventCreateWithFlags(&g->ev_join[i], hipEventDisableTiming) == hipSuccess;
        for (auto &e : g->ev_fork)
            ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) {
        jl_group_destroy(g);
        return JL_ERR_MEMORY;
    }
    *out = g;
    return JL_OK;
}

void jl_group_destroy(jl_group *g)
{
    if (!g) return;
    hipSetDevice(g->device);
    if (g->stream) hipStreamSynchronize(g->stream);
    group_drop_graphs(g);
    if (g->x_host) hipHostFree(g->x_host);
    if (g->x_dev) hipFree(g->x_dev);
    for (auto &e : g->x_done)
        if (e) hipEventDestroy(e);
    if (g->d_done) hipFree(g->d_done);
    if (g->ev_end) hipEventDestroy(g->ev_end);
    for (auto &e : g->ev_fork)
        if (e) hipEventDestroy(e);
    for (int i = 0; i < JL_GROUP_SIDE_STREAMS; ++i) {
        if (g->ev_join[i]) hipEventDestroy(g->ev_join[i]);
        if (g->side[i]) hipStreamDestroy(g->side[i]);
    }
    if (g->stream) hipStreamDestroy(g->stream);
    delete g;
}

const char *jl_group_last_error(const jl_group *g) { return g ? g->err.c_str() : ""; }

int jl_group_views(jl_group *g, jl_run_view *out, uint32_t cap, uint32_t *n)
{
    if (!g || !n || (!out && cap)) return JL_ERR_ARG;
    *n = (uint32_t)g->ctxs.size();
    if (g->ctxs.size() > cap) return group_fail(g, JL_ERR_OVERFLOW, "jl_group_views: more windows than the caller's array holds");
    for (size_t k = 0; k < g->ctxs.size(); ++k)
        if (int rc = jl_run_view_get(g->ctxs[k], &out[k])) {
            g->err = "window " + std::to_string(k) + ": " + jl_last_error(g->ctxs[k]);
            return rc;
        }
    return JL_OK;
}

int jl_group_run_async(jl_group *g, const jl_gene *genes, uint32_t n_genes, const uint8_t *refseq, uint32_t ref_len,
                       const jl_params *prm, int phasing, uint32_t min_reads, int want_read_hap)
{
    return jl_group_run_masked_async(g, genes, n_genes, refseq, ref_len, prm, nullptr, phasing, min_reads, want_read_hap);
}

int jl_group_run_masked_async(jl_group *g, const jl_gene *genes, uint32_t n_genes, const uint8_t *refseq, uint32_t ref_len,
                              const jl_params *prm, const uint64_t *const *drm_masks, int phasing, uint32_t min_reads,
                              int want_read_hap)
{
    if (!g || !prm) return JL_ERR_ARG;
    if (hipSetDevice(g->device) != hipSuccess) return group_fail(g, JL_ERR_DEVICE, "hipSetDevice failed");
    const uint32_t n = (uint32_t)g->ctxs.size();
    std::vector<double> n_tests(n);
    const uint32_t par = g->xc ? (uint32_t)(g->x_launched & 1u) : 0u;
    if (g->xc && g->x_pending[par])   // (the same on every rank: nobody issues a collective)
        return group_fail(g, JL_ERR_STATE, "two exchanges of this group are pending: collect one first (jl_group_exchange_collect)");
    // Bound exchange: this run's heads go to the region of its parity (this rank's part), whose magic words are cleared first —
    // whatever goes wrong below on THIS rank, on the host or on the device, the collective is still issued (the peers issue
    // theirs and would wait for ever) and carries empty heads, which every rank reads as this rank's failure.
    bool x_ok = true;
    if (g->xc) {
        uint8_t *mine = g->x_work(par) + g->x_part() * (size_t)g->xc->rank;
        if (!g->x_staged) {
            for (uint32_t k = 0; k < n; ++k) reinterpret_cast<jl_pack *>(mine + (size_t)JL_PACK_HEAD_BYTES * k)->magic = 0u;
        } else if (hipMemsetAsync(mine, 0, g->x_part(), g->stream) != hipSuccess) {
            x_ok = false;
        }
    }
    // ---- the run itself
    auto run_part = [&]() -> int {
    // per-window preparation (plans, buffers, parameter blocks): may allocate and wait, so it comes before any enqueue
    for (uint32_t k = 0; k < n; ++k) {
        jl_ctx *c = g->ctxs[k];
        int rc = jl_run_prepare(c, genes, n_genes, refseq, ref_len, prm, drm_masks ? drm_masks[k] : nullptr, phasing, min_reads,
                                want_read_hap, &n_tests[k]);
        if (rc) return group_fail(g, rc, jl_last_error(c));
        // a group run reads what the window's own stream wrote (uploads, ingest): that stream must be idle
        if (hipStreamSynchronize(c->stream) != hipSuccess) return group_fail(g, JL_ERR_DEVICE, "context stream failed");
        if (phasing && (c->phase_generic || c->phase_two))
            return group_fail(g, JL_ERR_ARG, "a window needs the two-word or the multi-word phasing pipeline: run it with jl_run_async");
    }
    // signature: anything that changes an argument block or the launch shapes
    struct item { uint64_t alloc, plan; double n_tests; void *rh; uint32_t n_dw, drm; };
    std::vector<uint8_t> sig(sizeof(jl_params) + 16 + sizeof(item) * n);
    memcpy(sig.data(), prm, sizeof(jl_params));
    const uint32_t flags[4] = {(uint32_t)(phasing != 0) | (g->xc ? 2u : 0u) | (g->x_staged ? 4u : 0u), min_reads, (uint32_t)(want_read_hap != 0), n};
    memcpy(sig.data() + sizeof(jl_params), flags, 16);
    for (uint32_t k = 0; k < n; ++k) {
        jl_ctx *c = g->ctxs[k];
        item it;
        memset(&it, 0, sizeof it);
        it.alloc = c->alloc_version; it.plan = c->plan_version; it.n_tests = n_tests[k]; it.rh = c->read_hap_out;
        it.n_dw = (uint32_t)(c->col_stride / 4u);
        it.drm = (drm_masks && drm_masks[k]) ? 1u : 0u;
        memcpy(sig.data() + sizeof(jl_params) + 16 + sizeof(item) * k, &it, sizeof it);
    }
    if (sig != g->sig) {
        group_drop_graphs(g);
        g->sig.clear();
        if (hipStreamSynchronize(g->stream) != hipSuccess) return group_fail(g, JL_ERR_DEVICE, "group stream failed");
        g->phasing = phasing != 0;
        // same kernel variant everywhere?  (checked again by the launcher)
        for (uint32_t k = 1; k < n; ++k)
            if (g->ctxs[k]->pileup_w != g->ctxs[0]->pileup_w)
                return group_fail(g, JL_ERR_ARG, "the windows of a group must share the pileup chunk width (gene layouts too different)");
        // chunks of at most JL_GROUP_MAX windows; the remainder goes last (its tail is the exposed one)
        g->chunks.clear();
        for (uint32_t o = 0; o < n; o += JL_GROUP_MAX) {
            jl_group::chunk_t c;
            memset(&c, 0, sizeof c);
            c.first = o;
            c.n = std::min<uint32_t>(JL_GROUP_MAX, n - o);
            // The phase launch writes the per-read ids itself when ALL its workgroups can wait for each other, i.e.
            // are resident at once — also while more such launches run: at most JL_FOLD_MAX_BLOCKS per launch against
            // 1536 places (six 75-register blocks per CU).  Larger chunks take a separate launch for the ids.
            uint32_t total_blocks = 0;
            for (uint32_t k = o; k < o + c.n; ++k) total_blocks += (uint32_t)((g->ctxs[k]->col_stride / 4u + 255u) / 256u) + 1u;
            c.fold = total_blocks <= JL_FOLD_MAX_BLOCKS;
            for (uint32_t k = o; k < o + c.n; ++k) {
                jl_ctx *x = g->ctxs[k];
                jl_fill_win_pileup(x, &g->h_pile[k]);
                jl_fill_win_call(x, prm, n_tests[k], drm_masks && drm_masks[k], phasing != 0, &g->h_call[k]);
                jl_fill_win_fold(x, &g->h_call[k], &g->h_fold[k]);
                jl_fill_win_compact(x, false, true, false, &g->h_compact[k]);
                jl_fill_win_phase(x, min_reads, false, c.fold ? 0xFFFFFFFFu : 0u, true, &g->h_phase[k]);
                c.max_chunks = std::max(c.max_chunks, g->h_pile[k].n_chunks);
                c.max_call_blocks = std::max(c.max_call_blocks, g->h_call[k].n_blocks);
                c.max_phase_blocks = std::max(c.max_phase_blocks, g->h_phase[k].n_blocks);
            }
            g->chunks.push_back(c);
        }
        {
            std::vector<jl_done_ent> ents(n);
            for (uint32_t k = 0; k < n; ++k) { ents[k].seq_dev = g->ctxs[k]->d_sync; ents[k].seq_host = g->ctxs[k]->h_seq; }
            // (on the group's own stream, not the null stream: rank threads of one process run side by side, and a synchronous copy
            // on the null stream while another thread captures its graph failed once in a few hundred runs of the suite)
            hipError_t he = hipMemcpyAsync(g->d_done, ents.data(), sizeof(jl_done_ent) * n, hipMemcpyHostToDevice, g->stream);
            if (he == hipSuccess) he = hipStreamSynchronize(g->stream);
            if (he != hipSuccess) {
                g->err = std::string("argument tables: ") + hipGetErrorString(he);
                return JL_ERR_DEVICE;
            }
        }
        g->sig = sig;
    }
    for (uint32_t k = 0; k < n; ++k) {
        uint8_t *xh = g->xc ? g->x_work(par) + g->x_part() * (size_t)g->xc->rank + (size_t)JL_PACK_HEAD_BYTES * k : nullptr;
        g->h_phase[k].S.xhead = xh;
        g->h_compact[k].xhead = xh;
    }
    static const bool graphs_on = !getenv("JL_NO_GRAPH");
    if (graphs_on && !g->graph_exec[par] && !g->graph_tried[par]) {
        g->graph_tried[par] = true;
        if (hipStreamBeginCapture(g->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            const int erc = group_enqueue(g);
            hipGraph_t gr = nullptr;
            if (hipStreamEndCapture(g->stream, &gr) == hipSuccess && gr && erc == JL_OK &&
                hipGraphInstantiate(&g->graph_exec[par], gr, nullptr, nullptr, 0) == hipSuccess) {
                g->graph[par] = gr;
            } else {
                if (gr) hipGraphDestroy(gr);
                g->graph_exec[par] = nullptr;
            }
            (void)hipGetLastError();
        }
    }
    bool launched = g->graph_exec[par] && hipGraphLaunch(g->graph_exec[par], g->stream) == hipSuccess;
    if (!launched) {
        const int erc = group_enqueue(g);
        if (erc != JL_OK || hipGetLastError() != hipSuccess) return group_fail(g, erc ? erc : JL_ERR_DEVICE, "group launch failed");
    }
    return JL_OK;
    };
    const int run_rc = run_part();
    int xrc = x_ok ? JL_OK : JL_ERR_DEVICE;
    if (g->xc) {
        // The exchange of this run, behind it on the same stream: ONE collective, in place in the region the kernels have
        // written this rank's heads into — pinned host memory (nothing else to do: the event behind it says the heads of
        // all ranks are there), or its device stage and one copy.  Issued whatever happened above: the peers issue theirs.
        jl_comm *c = g->xc;
        uint8_t *work = g->x_work(par);
        jl_comm_direct_wait_begin(c);
        if (jl_tp_allgather(c, work + g->x_part() * (size_t)c->rank, work, g->x_part(), g->stream) != JL_OK && xrc == JL_OK) xrc = JL_ERR_COMM;
        jl_comm_direct_end(c);
        if (g->x_staged) {      // (the end of that kernel is what pushes its stores to the host out; the event behind it says so)
            jl_launch_heads_to_host(work, g->x_host + g->x_region() * par, (uint32_t)(g->x_region() / JL_PACK_HEAD_BYTES), g->stream);
            if (hipGetLastError() != hipSuccess && xrc == JL_OK) xrc = JL_ERR_DEVICE;
        }
        if (hipEventRecord(g->x_done[par], g->stream) != hipSuccess && xrc == JL_OK) xrc = JL_ERR_DEVICE;
        g->x_pending[par] = true;
        ++g->x_launched;
    } else if (g->ev_end && run_rc == JL_OK) {
        // An event behind the launch: the runtime retires a stream whose last command is an event marker without a marker of
        // its own (hipStreamSynchronize / the closing hipDeviceSynchronize of a short run: 12 instead of 20-30 us per stream).
        hipEventRecord(g->ev_end, g->stream);
    }
    if (run_rc != JL_OK) return run_rc;   // (its message stands; the exchange, if any, went out with empty heads and is pending)
    for (uint32_t k = 0; k < n; ++k) {
        jl_ctx *c = g->ctxs[k];
        jl_run_finish(c, phasing, want_read_hap);
        c->run_stream = g->stream;
        if (g->xc) g->x_run_seq[par][k] = c->runs_launched;
    }
    if (xrc != JL_OK) return group_fail(g, xrc, xrc == JL_ERR_COMM ? g->xc->tp_error.c_str() : "the run's exchange could not be enqueued");
    return JL_OK;
}

int jl_group_exchange_bind(jl_group *g, jl_comm *c)
{
    if (!g) return JL_ERR_ARG;
    if (hipSetDevice(g->device) != hipSuccess) return group_fail(g, JL_ERR_DEVICE, "hipSetDevice failed");
    if (g->x_pending[0] || g->x_pending[1]) return group_fail(g, JL_ERR_STATE, "an exchange of this group is pending: collect it first");
    if (c == g->xc) return JL_OK;
    if (hipStreamSynchronize(g->stream) != hipSuccess) return group_fail(g, JL_ERR_DEVICE, "group stream failed");
    group_drop_graphs(g);
    g->sig.clear();
    if (g->x_host) { hipHostFree(g->x_host); g->x_host = nullptr; }
    if (g->x_dev) { hipFree(g->x_dev); g->x_dev = nullptr; }
    g->xc = nullptr;
    g->x_launched = g->x_collected = 0;
    if (!c) return JL_OK;
    if (c->device != g->device) return group_fail(g, JL_ERR_ARG, "group and communicator are on different devices");
    if (g->ctxs.size() > JL_GATHER_MAX) return group_fail(g, JL_ERR_ARG, "a bound exchange carries at most 32 windows");
    g->xc = c;
    g->x_staged = jl_comm_host_gather(c) == 0;   // (a collective the first time a communicator is asked)
    bool ok = hipHostMalloc(&g->x_host, 2 * g->x_region(), hipHostMallocDefault) == hipSuccess;
    if (ok) memset(g->x_host, 0, 2 * g->x_region());
    if (ok && g->x_staged) ok = hipMalloc(&g->x_dev, 2 * g->x_region()) == hipSuccess && hipMemsetAsync(g->x_dev, 0, 2 * g->x_region(), g->stream) == hipSuccess &&
                                hipStreamSynchronize(g->stream) == hipSuccess;
    for (auto &e : g->x_done)
        if (ok && !e) ok = hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        g->xc = nullptr;
        return group_fail(g, JL_ERR_MEMORY, "exchange regions");
    }
    return JL_OK;
}

int jl_group_exchange_collect(jl_group *g, jl_variant *all_rows, uint32_t *all_counts, uint32_t cap_rows)
{
    if (!g || !all_rows || !all_counts) return JL_ERR_ARG;
    if (!g->xc) return group_fail(g, JL_ERR_STATE, "jl_group_exchange_collect: no exchange is bound to this group");
    if (cap_rows == 0 || cap_rows > JL_VARIANT_CAP) return group_fail(g, JL_ERR_ARG, "cap_rows must be 1..4096");
    const uint32_t par = (uint32_t)(g->x_collected & 1u);
    if (!g->x_pending[par]) return group_fail(g, JL_ERR_STATE, "jl_group_exchange_collect: no exchange of this group is pending");
    if (hipSetDevice(g->device) != hipSuccess) return group_fail(g, JL_ERR_DEVICE, "hipSetDevice failed");
    jl_comm *c = g->xc;
    {   // spin on the event (a blocking wait costs ~15 us of wake-up latency), but not for ever: a peer may have died
        hipError_t q;
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(JL_COMM_TIMEOUT_S);
        uint32_t polls = 0;
        while ((q = hipEventQuery(g->x_done[par])) == hipErrorNotReady)
            if ((++polls & 0xFFFu) == 0 && std::chrono::steady_clock::now() > deadline)
                return group_fail(g, JL_ERR_COMM, "the group's exchange is not complete after the time-out: a peer has not issued its collective");
        if (q != hipSuccess) return group_fail(g, JL_ERR_DEVICE, "the group's exchange failed on the device");
    }
    g->x_pending[par] = false;
    ++g->x_collected;
    // a window whose matrix an enqueued record ingest made: what that ingest found wrong with the records is reported before
    // tables computed from it are handed out (the run is over: the verdict word is final)
    for (jl_ctx *ctx : g->ctxs)
        if (ctx->ing_check_pending)
            if (int vrc = jl_ingest_verdict(ctx)) return group_fail(g, vrc, ctx->err.c_str());
    const uint32_t n = (uint32_t)g->ctxs.size();
    const uint8_t *base = g->x_host + g->x_region() * par;
    auto head = [&](int r, uint32_t k) { return reinterpret_cast<const jl_pack *>(base + g->x_part() * (size_t)r + (size_t)JL_PACK_HEAD_BYTES * k); };
    bool compact = true;
    for (int r = 0; r < c->world; ++r)
        for (uint32_t k = 0; k < n; ++k) {
            // a head without the magic word: that rank's run did not reach its result block; every rank sees the same
            if (head(r, k)->magic != JL_PACK_MAGIC) {
                g->err = "rank " + std::to_string(r) + "'s run of window " + std::to_string(k) + " did not complete: its head of the exchange is empty";
                return JL_ERR_COMM;
            }
            if (!head(r, k)->fits_call) compact = false;
        }
    int rc = JL_OK;
    for (uint32_t k = 0; k < n; ++k) {
        jl_variant *rows = all_rows + (size_t)k * c->world * cap_rows;
        uint32_t *counts = all_counts + (size_t)k * c->world;
        if (compact) {
            for (int r = 0; r < c->world; ++r) {
                const jl_pack *pk = head(r, k);
                counts[r] = pk->nvar_total;
                if (pk->nvar_total > cap_rows) { rc = JL_ERR_OVERFLOW; continue; }
                memcpy(rows + (size_t)r * cap_rows, pk->variants, (size_t)pk->nvar_total * sizeof(jl_variant));
            }
            continue;
        }
        // Some rank's table has more rows than a head holds: the full stride from the resident tables, window by window —
        // every rank takes this branch (same heads everywhere).  The resident table is the LAST run's.
        jl_ctx *x = g->ctxs[k];
        if (x->runs_launched != g->x_run_seq[par][k])
            return group_fail(g, JL_ERR_STATE, "a rank called more than 128 variants: such a run's exchange must be collected before the "
                                               "group's next run (the full table is not double-buffered)");
        rc = jl_comm_allgather_full(x, c, rows, counts, cap_rows, g->x_run_seq[par][k]);
        if (rc) return group_fail(g, rc, jl_last_error(x));
    }
    if (rc) return group_fail(g, rc, "a rank produced more rows than cap_rows");
    return JL_OK;
}

// Average device time in ms of the grouped pileup launch alone (the launch of a group's FIRST chunk, with the Fisher
// stage in its epilogue as in a run): `reps` back-to-back launches rotating over the given groups (all on the first
// group's stream), one pair of HIP events around them.  Every group must have run at least once (its argument
// tables are what the launch reads).
int jl_group_time_pileup(jl_group *const *groups, uint32_t n_groups, uint32_t reps, float *ms_avg, uint64_t *bytes_per_launch)
{
    if (!groups || n_groups == 0 || !ms_avg || reps == 0) return JL_ERR_ARG;
    jl_group *g0 = groups[0];
    for (uint32_t k = 0; k < n_groups; ++k)
        if (!groups[k] || groups[k]->sig.empty() || groups[k]->device != g0->device)
            return group_fail(g0, JL_ERR_STATE, "jl_group_time_pileup: every group must have run once");
    if (hipSetDevice(g0->device) != hipSuccess) return group_fail(g0, JL_ERR_DEVICE, "hipSetDevice failed");
    for (uint32_t k = 0; k < n_groups; ++k)
        if (hipStreamSynchronize(groups[k]->stream) != hipSuccess) return group_fail(g0, JL_ERR_DEVICE, "group stream failed");
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return group_fail(g0, JL_ERR_DEVICE, "events");
    auto launch = [&](jl_group *g) {
        const jl_group::chunk_t &c = g->chunks[0];
        // (the launch the group's runs make: the pileup with the Fisher stage in its epilogue)
        if (!jl_fold_enabled()) return jl_launch_pileup_group(g->ctxs.data() + c.first, c.n, g->h_pile.data() + c.first, c.max_chunks, g0->stream);
        return jl_launch_pileup_fold_group(g->ctxs.data() + c.first, c.n, g->h_pile.data() + c.first, g->h_fold.data() + c.first, c.max_chunks, g0->stream);
    };
    int rc = JL_OK;
    for (uint32_t k = 0; k < n_groups && rc == JL_OK; ++k) rc = launch(groups[k]);   // warm-up, once per group
    hipEventRecord(e0, g0->stream);
    for (uint32_t r = 0; r < reps && rc == JL_OK; ++r) rc = launch(groups[r % n_groups]);
    hipEventRecord(e1, g0->stream);
    float total = 0.f;
    if (rc == JL_OK && (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&total, e0, e1) != hipSuccess)) rc = JL_ERR_DEVICE;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    if (rc) return group_fail(g0, rc, "grouped pileup launch failed");
    *ms_avg = total / (float)reps;
    if (bytes_per_launch) {
        uint64_t b = 0;
        const jl_group::chunk_t &c = g0->chunks[0];
        for (uint32_t k = c.first; k < c.first + c.n; ++k) b += (uint64_t)g0->ctxs[k]->n_reads * g0->ctxs[k]->n_cols * 3u / 8u;   // 3 bits per cell
        *bytes_per_launch = b;
    }
    for (uint32_t k = 0; k < n_groups; ++k)
        for (jl_ctx *c : groups[k]->ctxs) {
            c->pileup_done = true;
            c->call_done = c->phase_done = false;
            c->pack_valid = false;
        }
    return JL_OK;
}

}  // extern "C"
