// capi_group.hip — group runs: the whole path for SEVERAL resident windows in three launches.
//
// Why.  A 3 kb x 100k-read window is a 150 MB stream: too short to hide a launch's ramp and drain (one pileup launch
// reaches 0.61 of the HBM peak, a 600 MB stream 0.78), and its Fisher / phasing stages are chains of dependent memory
// round trips that keep a hardware queue busy for tens of microseconds while doing almost nothing.  The chip runs
// at most four queues at once (more are time-sliced: measured), so with one graph per window the queues' time — not
// the HBM — bounds the throughput.  A group run gives each stage ONE launch for all windows of the group
// (blockIdx.z = window, per-window argument blocks in device memory): the pileup becomes one long stream, and the
// latency chains of the other stages run side by side instead of one after the other.
//
// Results are per window, exactly those of jl_run_async on each context: every context keeps its own result block,
// per-read ids and completion word, so jl_run_wait / jl_run_view_get / jl_call_fetch / jl_phase_fetch work unchanged.
#include <string.h>

#include <algorithm>
#include <new>
#include <string>
#include <vector>

#include "jl_internal.h"

struct jl_group {
    int device = -1;
    hipStream_t stream = nullptr;
    std::vector<jl_ctx *> ctxs;
    jl_done_ent *d_done = nullptr;
    std::vector<jl_win_pileup> h_pile;
    std::vector<jl_win_call> h_call;
    std::vector<jl_win_phase> h_phase;
    uint32_t max_chunks = 0, max_call_blocks = 0, max_phase_blocks = 0, max_read_blocks = 0;
    bool fold = true;   // the phase launch also writes the per-read ids (its workgroups wait for each other)
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    std::vector<uint8_t> sig;   // everything the captured graph and the tables bake in
    std::string err;
};

static int group_fail(jl_group *g, int status, const char *msg)
{
    if (g) g->err = msg;
    return status;
}

static void group_enqueue(jl_group *g, bool phasing)
{
    const uint32_t n = (uint32_t)g->ctxs.size();
    for (jl_ctx *c : g->ctxs)
        if (!c->have_ref) jl_launch_guess(c, g->stream);   // majority-codon mode: seed bases per window
    // JL_GROUP_SKIP (tuning only, results are then stale): 1 = pileup alone, 2 = no phasing stages, 3 = no id launch
    static const int skip = getenv("JL_GROUP_SKIP") ? atoi(getenv("JL_GROUP_SKIP")) : 0;
    jl_launch_pileup_group(g->ctxs.data(), n, g->h_pile.data(), g->max_chunks, g->stream);
    // the later stages take their windows' argument blocks (250-350 bytes each) by value too: JL_GROUP_MAX per launch
    bool to_host = false;
    for (jl_ctx *c : g->ctxs) to_host = to_host || c->read_hap_out != nullptr;
    for (uint32_t o = 0; o < n; o += JL_GROUP_MAX) {
        const uint32_t m = std::min<uint32_t>(JL_GROUP_MAX, n - o);
        if (skip != 1) jl_launch_call_group(g->h_call.data() + o, m, g->max_call_blocks, g->stream);
        if (phasing && skip != 1 && skip != 2) {
            jl_launch_phase_group(g->h_phase.data() + o, m, g->max_phase_blocks, g->stream);
            if (!g->fold && skip != 3) jl_launch_assign_group(g->h_phase.data() + o, m, g->max_read_blocks, to_host, g->stream);
        }
    }
    // completion words of all windows, behind the end of the last stage (see enqueue_path in capi.hip)
    jl_launch_done_group(g->d_done, n, g->stream);
}

extern "C" {

int jl_group_create(jl_ctx *const *ctxs, uint32_t n_ctx, jl_group **out)
{
    if (!ctxs || !out || n_ctx == 0 || n_ctx > JL_GROUP_WINDOWS_MAX) return JL_ERR_ARG;   // the argument blocks travel by value
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
    g->h_phase.resize(n_ctx);
    bool ok = hipSetDevice(g->device) == hipSuccess &&
              hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) == hipSuccess &&
              hipMalloc(&g->d_done, sizeof(jl_done_ent) * n_ctx) == hipSuccess;
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
    if (g->graph_exec) hipGraphExecDestroy(g->graph_exec);
    if (g->graph) hipGraphDestroy(g->graph);
    if (g->d_done) hipFree(g->d_done);
    if (g->stream) hipStreamDestroy(g->stream);
    delete g;
}

const char *jl_group_last_error(const jl_group *g) { return g ? g->err.c_str() : ""; }

int jl_group_run_async(jl_group *g, const jl_gene *genes, uint32_t n_genes, const uint8_t *refseq, uint32_t ref_len,
                       const jl_params *prm, int phasing, uint32_t min_reads, int want_read_hap)
{
    if (!g || !prm) return JL_ERR_ARG;
    if (hipSetDevice(g->device) != hipSuccess) return group_fail(g, JL_ERR_DEVICE, "hipSetDevice failed");
    const uint32_t n = (uint32_t)g->ctxs.size();
    std::vector<double> n_tests(n);
    // per-window preparation (plans, buffers): may allocate and wait, so it comes before any enqueue
    for (uint32_t k = 0; k < n; ++k) {
        jl_ctx *c = g->ctxs[k];
        int rc = jl_run_prepare(c, genes, n_genes, refseq, ref_len, prm, nullptr, phasing, min_reads, want_read_hap, &n_tests[k]);
        if (rc) return group_fail(g, rc, jl_last_error(c));
        // a group run reads what the window's own stream wrote (uploads, ingest): that stream must be idle
        if (hipStreamSynchronize(c->stream) != hipSuccess) return group_fail(g, JL_ERR_DEVICE, "context stream failed");
        if (phasing && c->phase_generic)
            return group_fail(g, JL_ERR_ARG, "a window needs the multi-word phasing pipeline: run it with jl_run_async");
    }
    // signature: anything that changes an argument block or the launch shapes
    struct item { uint64_t alloc, plan; double n_tests; void *rh; uint32_t n_dw, pad; };
    std::vector<uint8_t> sig(sizeof(jl_params) + 16 + sizeof(item) * n);
    memcpy(sig.data(), prm, sizeof(jl_params));
    const uint32_t flags[4] = {(uint32_t)(phasing != 0), min_reads, (uint32_t)(want_read_hap != 0), n};
    memcpy(sig.data() + sizeof(jl_params), flags, 16);
    for (uint32_t k = 0; k < n; ++k) {
        jl_ctx *c = g->ctxs[k];
        item it;
        memset(&it, 0, sizeof it);
        it.alloc = c->alloc_version; it.plan = c->plan_version; it.n_tests = n_tests[k]; it.rh = c->read_hap_out;
        it.n_dw = (uint32_t)(c->col_stride / 4u);
        memcpy(sig.data() + sizeof(jl_params) + 16 + sizeof(item) * k, &it, sizeof it);
    }
    if (sig != g->sig) {
        if (g->graph_exec) { hipGraphExecDestroy(g->graph_exec); g->graph_exec = nullptr; }
        if (g->graph) { hipGraphDestroy(g->graph); g->graph = nullptr; }
        g->sig.clear();
        if (hipStreamSynchronize(g->stream) != hipSuccess) return group_fail(g, JL_ERR_DEVICE, "group stream failed");
        g->max_chunks = g->max_call_blocks = g->max_phase_blocks = 0;
        // The phase launch writes the per-read ids itself when ALL its workgroups can wait for each other, i.e. are
        // resident at once — also while more such launches run: at most JL_FOLD_MAX_BLOCKS per launch against 1536
        // places (six 75-register blocks per CU).  Larger groups take a separate launch for the ids: nothing waits
        // for anything then.
        uint32_t total_blocks = 0;
        for (uint32_t k = 0; k < n; ++k) total_blocks += (uint32_t)((g->ctxs[k]->col_stride / 4u + 255u) / 256u);
        g->fold = total_blocks <= JL_FOLD_MAX_BLOCKS && !getenv("JL_NO_FOLD");
        g->max_read_blocks = 0;
        for (uint32_t k = 0; k < n; ++k) {
            jl_ctx *c = g->ctxs[k];
            jl_fill_win_pileup(c, &g->h_pile[k]);
            jl_fill_win_call(c, prm, n_tests[k], false, phasing != 0, &g->h_call[k]);
            jl_fill_win_phase(c, min_reads, false, g->fold ? 0xFFFFFFFFu : 0u, &g->h_phase[k]);
            if (!phasing) g->h_call[k].meta = nullptr;
            g->max_chunks = std::max(g->max_chunks, g->h_pile[k].n_chunks);
            g->max_call_blocks = std::max(g->max_call_blocks, g->h_call[k].n_blocks);
            g->max_phase_blocks = std::max(g->max_phase_blocks, g->h_phase[k].n_blocks);
            g->max_read_blocks = std::max(g->max_read_blocks, g->h_phase[k].n_blocks);   // 8 reads per lane there too
        }
        if (!phasing)
            return group_fail(g, JL_ERR_ARG, "group runs are built for call + phase; run call-only windows with jl_run_async");
        {
            std::vector<jl_done_ent> ents(n);
            for (uint32_t k = 0; k < n; ++k) { ents[k].seq_dev = g->ctxs[k]->d_sync; ents[k].seq_host = g->ctxs[k]->h_seq; }
            if (hipMemcpy(g->d_done, ents.data(), sizeof(jl_done_ent) * n, hipMemcpyHostToDevice) != hipSuccess)
                return group_fail(g, JL_ERR_DEVICE, "argument tables");
        }
        // same variant / launch shape everywhere?  (checked by the launcher; probe it outside the capture)
        for (uint32_t k = 1; k < n; ++k)
            if (g->ctxs[k]->pileup_w != g->ctxs[0]->pileup_w)
                return group_fail(g, JL_ERR_ARG, "the windows of a group must share the pileup chunk width (gene layouts too different)");
        if (!getenv("JL_NO_GRAPH") && hipStreamBeginCapture(g->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            group_enqueue(g, phasing != 0);
            hipGraph_t gr = nullptr;
            if (hipStreamEndCapture(g->stream, &gr) == hipSuccess && gr &&
                hipGraphInstantiate(&g->graph_exec, gr, nullptr, nullptr, 0) == hipSuccess) {
                g->graph = gr;
            } else {
                if (gr) hipGraphDestroy(gr);
                g->graph_exec = nullptr;
            }
            (void)hipGetLastError();
        }
        g->sig = sig;
    }
    bool launched = g->graph_exec && hipGraphLaunch(g->graph_exec, g->stream) == hipSuccess;
    if (!launched) {
        group_enqueue(g, phasing != 0);
        if (hipGetLastError() != hipSuccess) return group_fail(g, JL_ERR_DEVICE, "group launch failed");
    }
    for (jl_ctx *c : g->ctxs) {
        jl_run_finish(c, phasing, want_read_hap);
        c->run_stream = g->stream;
    }
    return JL_OK;
}

// Average device time in ms of the grouped pileup launch alone: `reps` back-to-back launches rotating over the
// given groups (all on the first group's stream), one pair of HIP events around them.  Every group must have run
// at least once (its argument table is what the launch reads).
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
        return jl_launch_pileup_group(g->ctxs.data(), (uint32_t)g->ctxs.size(), g->h_pile.data(), g->max_chunks, g0->stream);
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
        for (jl_ctx *c : g0->ctxs) b += (uint64_t)c->n_reads * c->n_cols / 2u;
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
