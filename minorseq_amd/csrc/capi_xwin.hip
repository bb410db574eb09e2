// capi_xwin.hip — cross-window phasing behind the C ABI (BASELINE.json configs[3]/[4]: ONE reference whose
// reads span every column window).  Behaviour: doc/JULIET.md:192-211 (phasing, haplotype_hit), :253-254 (>= 10 reads:
// applied to the merged count), :261-264 (each gene separately: the split into windows never shows in the output).
//
//   plan        jl_xwin_plan / jl_xwin_slice_plan (host only; xwin_schedule.h, capi_merge.hip)
//   exchange    the variant columns of every window gathered into a compact matrix (position k = columns 3k..3k+2):
//               whole columns (replicated form, jl_xwin_assemble_local / _rccl) or one slice of the reads per rank
//               (jl_xwin_assemble_slice_local / _rccl): ONE pack launch + ONE packed send per peer
//   session     jl_xwin_phase_sharded: tables -> merge -> plan -> exchange -> grouping + export -> gather -> merge +
//               selection -> per-read ids, with no host synchronisation call: every hand-off to the host is a word in
//               pinned memory behind the data it announces.
// Collectives are issued from the calling thread while it holds the communicator (jl_comm_direct_begin).
//
// NOTHING here has run with more than one rank on hardware (one GPU per box in this project's pool); the schedule is
// exercised as data over gloo (tests/test_sharding_gloo.py) and with a one-rank communicator on the GPU.
#include <string.h>

#include <algorithm>
#include <chrono>
#include <new>
#include <string>
#include <vector>

#include "jl_comm_internal.h"
#include "xwin_schedule.h"

namespace {

struct xw_layout {   // the column windows of the whole run and who holds them
    std::vector<uint32_t> win_begin, win_ncols;
    std::vector<int32_t> win_rank;
    uint32_t first_local = 0;   // global index of this rank's first window
};

// What one exchange needs besides the schedule.  `send`: grown on demand, holds the packed messages to the peers.
struct xw_send_buf {
    uint8_t *d = nullptr;
    size_t cap = 0;
};

int xw_fail(jl_ctx *pc, std::string *err, int status, const std::string &msg)
{
    if (err) *err = msg;
    if (pc) jl_fail(pc, status, "%s", msg.c_str());
    return status;
}

// Pack launches for this rank's owned positions: destination s gets slice s of the 9 * k_count plane rows of the owned
// columns, laid out as rows of stride dst_stride[s] (the own slice: straight into the compact matrix).  `plan`: non-null for the first
// launch of a step, which then also writes the compact matrix's phasing plan.
// room for what this rank sends its peers (nothing is enqueued: a rank that cannot allocate says so BEFORE the exchange)
int xw_reserve_send(jl_ctx *pc, const xwin_schedule &sch, const uint64_t *slice_begin, int world, int rank, xw_send_buf *send, std::string *err)
{
    const uint32_t kn = sch.k_count[(size_t)rank];
    size_t need = 0;
    if (kn)
        for (int s = 0; s < world; ++s) {
            const uint64_t n_s = slice_begin[s + 1] - slice_begin[s];
            if (s != rank && n_s) need += (size_t)9 * kn * xwin_stride(n_s);
        }
    if (need > send->cap) {
        if (send->d) hipFree(send->d);
        send->d = nullptr;
        send->cap = 0;
        if (hipMalloc(&send->d, need) != hipSuccess) {
            (void)hipGetLastError();
            return xw_fail(pc, err, JL_ERR_MEMORY, "send buffer of the column exchange");
        }
        send->cap = need;
    }
    return JL_OK;
}

int xw_pack(jl_ctx *pc, jl_ctx *const *wins, const xw_layout &lay, const xwin_schedule &sch, const uint64_t *slice_begin, int world,
            int rank, const uint64_t *dst_stride_override, xw_send_buf *send, const jl_xw_pack_args *plan, std::string *err)
{
    const uint32_t k0 = sch.k_begin[(size_t)rank], kn = sch.k_count[(size_t)rank];
    const uint64_t src_stride = wins ? wins[0]->plane_stride : 0;
    struct dst_t { uint8_t *dst; uint64_t stride, byte_begin, bytes; uint32_t tail_mask; };
    std::vector<dst_t> dsts;
    if (kn) {
        if (int rc = xw_reserve_send(pc, sch, slice_begin, world, rank, send, err)) return rc;
        size_t off = 0;
        for (int s = 0; s < world; ++s) {
            const uint64_t n_s = slice_begin[s + 1] - slice_begin[s];
            if (!n_s) continue;
            dst_t d;
            d.stride = (s == rank && dst_stride_override) ? *dst_stride_override : xwin_stride(n_s);
            d.byte_begin = slice_begin[s] / 8u;
            d.bytes = (n_s + 7u) / 8u;
            d.tail_mask = (n_s & 7u) ? (1u << (n_s & 7u)) - 1u : 0xFFu;
            if (s == rank) d.dst = pc->d_msa + (uint64_t)9 * k0 * d.stride;
            else { d.dst = send->d + off; off += (size_t)9 * kn * d.stride; }
            dsts.push_back(d);
        }
    }
    bool plan_done = plan == nullptr;
    for (uint32_t p0 = 0; p0 < kn; p0 += JL_XW_POS_MAX) {
        const uint32_t np = std::min<uint32_t>(JL_XW_POS_MAX, kn - p0);
        for (size_t d0 = 0; d0 < dsts.size(); d0 += JL_XW_DST_MAX) {
            jl_xw_pack_args a;
            memset(&a, 0, sizeof a);
            if (!plan_done) { a = *plan; plan_done = true; }
            for (uint32_t p = 0; p < np; ++p) {
                const uint32_t k = k0 + p0 + p;
                const uint32_t w = (uint32_t)sch.owner_win[k];
                const jl_ctx *wc = wins[w - lay.first_local];
                a.src[p] = wc->d_msa + (uint64_t)(sch.pos[k] - lay.win_begin[w]) * 3u * src_stride;
            }
            a.src_stride = src_stride;
            a.n_pos = np;
            a.n_dst = (uint32_t)std::min<size_t>(JL_XW_DST_MAX, dsts.size() - d0);
            for (uint32_t j = 0; j < a.n_dst; ++j) {
                const dst_t &d = dsts[d0 + j];
                a.d[j].dst = d.dst + (uint64_t)9 * p0 * d.stride;
                a.d[j].dst_stride = d.stride;
                a.d[j].byte_begin = d.byte_begin;
                a.d[j].bytes = d.bytes;
                a.d[j].tail_mask = d.tail_mask;
            }
            jl_launch_xw_pack(&a, pc->stream);
        }
    }
    if (!plan_done) jl_launch_xw_pack(plan, pc->stream);   // nothing to pack on this rank: the plan alone
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) return xw_fail(pc, err, JL_ERR_DEVICE, std::string("pack launch of the column exchange: ") + hipGetErrorString(le));
    return JL_OK;
}

// The peers' part: the op list of this rank, executed in order inside one RCCL group on pc's stream.  The caller holds
// the communicator.
int xw_send_recv(jl_ctx *pc, jl_comm *c, const xwin_schedule &sch, const uint64_t *slice_begin, const xw_send_buf &send, std::string *err)
{
    std::vector<jl_xwin_op> ops;
    xwin_ops_of_rank(sch, slice_begin, c->world, c->rank, &ops);
    std::vector<jl_tp_msg> msgs;
    size_t off = 0;
    for (const jl_xwin_op &o : ops) {
        if (o.op == JL_XWIN_OP_SEND) {
            msgs.push_back({o.peer, send.d + off, (size_t)o.bytes, true});
            off += o.bytes;
        } else if (o.op == JL_XWIN_OP_RECV) {
            msgs.push_back({o.peer, pc->d_msa + o.dst_offset, (size_t)o.bytes, false});
        }
    }
    if (jl_tp_exchange(c, msgs.data(), msgs.size(), pc->stream) != JL_OK) return xw_fail(pc, err, JL_ERR_COMM, "column slices: " + c->tp_error);
    return JL_OK;
}

int xw_check_windows(jl_ctx *pc, jl_ctx *const *windows, uint32_t n_windows, xw_layout *lay)
{
    for (uint32_t w = 0; w < n_windows; ++w) {
        if (!windows[w] || !windows[w]->d_msa) return jl_fail(pc, JL_ERR_ARG, "window %u has no resident matrix", w);
        if (windows[w]->n_reads != windows[0]->n_reads || windows[w]->col_stride != windows[0]->col_stride)
            return jl_fail(pc, JL_ERR_ARG, "windows must hold the same reads (window %u differs)", w);
        if (windows[w]->device != pc->device) return jl_fail(pc, JL_ERR_ARG, "window %u is on another device", w);
        lay->win_begin.push_back(windows[w]->win_begin);
        lay->win_ncols.push_back(windows[w]->n_cols);
        lay->win_rank.push_back(0);
    }
    for (uint32_t w = 1; w < n_windows; ++w)
        if (lay->win_begin[w] < lay->win_begin[w - 1]) return jl_fail(pc, JL_ERR_ARG, "windows must come in ascending column order");
    lay->first_local = 0;
    return JL_OK;
}

}  // namespace

extern "C" {

/* ---------------------------------------------------------------- plan (host only) */

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

/* ---------------------------------------------------------------- the exchange alone */

// All windows on THIS device (a 288 GB GPU holds many), whole columns: one pack launch.
int jl_xwin_assemble_local(jl_ctx *pc, jl_ctx *const *windows, uint32_t n_windows, const jl_variant *merged, uint32_t n_var,
                           jl_variant *remapped, uint32_t *pos_global, uint32_t *vp_total)
{
    if (!pc || !windows || n_windows == 0 || (!merged && n_var) || !vp_total) return JL_ERR_ARG;
    if (n_var > JL_VARIANT_CAP) return jl_fail(pc, JL_ERR_OVERFLOW, "%u variants, table holds %u", n_var, JL_VARIANT_CAP);
    xw_layout lay;
    int rc = xw_check_windows(pc, windows, n_windows, &lay);
    if (rc) return rc;
    const uint64_t n_reads = windows[0]->n_reads, sb[2] = {0, n_reads};
    xwin_schedule sch;
    std::string err;
    rc = xwin_make_schedule(lay.win_begin.data(), lay.win_ncols.data(), lay.win_rank.data(), n_windows, merged, n_var, sb, 1, remapped, &sch, &err);
    if (rc) return jl_fail(pc, rc, "%s", err.c_str());
    const uint32_t vp = (uint32_t)sch.pos.size();
    *vp_total = vp;
    if (pos_global) std::copy(sch.pos.begin(), sch.pos.end(), pos_global);
    if (vp == 0) return JL_OK;
    JL_HIP(pc, hipSetDevice(pc->device));
    const uint64_t stride = windows[0]->plane_stride;   // the windows' plane stride, whatever it is
    rc = jl_msa_alloc_strided(pc, n_reads, 3u * vp, stride, 0);
    if (rc) return rc;
    for (uint32_t w = 0; w < n_windows; ++w) JL_HIP(pc, hipStreamSynchronize(windows[w]->stream));
    xw_send_buf none;
    rc = xw_pack(pc, windows, lay, sch, sb, 1, 0, &stride, &none, nullptr, nullptr);
    if (rc) return rc;
    JL_HIP(pc, hipStreamSynchronize(pc->stream));
    return JL_OK;
}

// One window per rank, whole columns on every rank (the replicated form): every owner packs its columns into its own
// compact matrix and broadcasts that run of columns in place — one ncclBroadcast per owning rank.
int jl_xwin_assemble_rccl(jl_ctx *pc, jl_ctx *window, jl_comm *c, const uint32_t *win_begin, const uint32_t *win_ncols,
                          const jl_variant *merged, uint32_t n_var, jl_variant *remapped, uint32_t *pos_global,
                          uint32_t *vp_total)
{
    if (!pc || !window || !c || !win_begin || !win_ncols || (!merged && n_var) || !vp_total) return JL_ERR_ARG;
    if (!window->d_msa) return jl_fail(pc, JL_ERR_ARG, "window has no resident matrix");
    if (n_var > JL_VARIANT_CAP) return jl_fail(pc, JL_ERR_OVERFLOW, "%u variants, table holds %u", n_var, JL_VARIANT_CAP);
    const int world = c->world;
    xw_layout lay;
    lay.win_begin.assign(win_begin, win_begin + world);
    lay.win_ncols.assign(win_ncols, win_ncols + world);
    for (int r = 0; r < world; ++r) lay.win_rank.push_back(r);
    lay.first_local = (uint32_t)c->rank;
    const uint64_t n_reads = window->n_reads;
    // as a schedule: ONE slice (all reads) that belongs to this rank; the peers get theirs by the broadcasts below
    std::vector<uint64_t> sb((size_t)world + 1, 0);
    for (int r = c->rank + 1; r <= world; ++r) sb[(size_t)r] = n_reads;
    xwin_schedule sch;
    std::string err;
    int rc = xwin_make_schedule(lay.win_begin.data(), lay.win_ncols.data(), lay.win_rank.data(), (uint32_t)world, merged, n_var, sb.data(),
                                world, remapped, &sch, &err);
    if (rc) return jl_fail(pc, rc, "%s", err.c_str());
    const uint32_t vp = (uint32_t)sch.pos.size();
    *vp_total = vp;
    if (pos_global) std::copy(sch.pos.begin(), sch.pos.end(), pos_global);
    if (vp == 0) return JL_OK;
    JL_HIP(pc, hipSetDevice(pc->device));
    const uint64_t stride = window->plane_stride;   // every rank's window must use the same stride (same reads)
    rc = jl_msa_alloc_strided(pc, n_reads, 3u * vp, stride, 0);
    if (rc) return rc;
    JL_HIP(pc, hipStreamSynchronize(window->stream));
    xw_send_buf none;
    jl_ctx *wins[1] = {window};
    rc = xw_pack(pc, wins, lay, sch, sb.data(), world, c->rank, &stride, &none, nullptr, nullptr);
    if (rc) return rc;
    if (jl_comm_direct_begin(c) != JL_OK)
        return jl_fail(pc, JL_ERR_STATE, "the communicator has asynchronous exchanges queued or uncollected: collect them first");
    std::vector<jl_tp_bcast> bc;
    for (int o = 0; o < world; ++o) {
        if (!sch.k_count[(size_t)o]) continue;
        bc.push_back({pc->d_msa + (uint64_t)9 * sch.k_begin[(size_t)o] * stride, (size_t)9 * sch.k_count[(size_t)o] * stride, o});
    }
    const int brc = jl_tp_broadcasts(c, bc.data(), bc.size(), pc->stream);
    jl_comm_direct_end(c);
    if (brc != JL_OK) return jl_fail(pc, JL_ERR_COMM, "broadcast of the variant columns: %s", c->tp_error.c_str());
    JL_HIP(pc, hipStreamSynchronize(pc->stream));
    return JL_OK;
}

// ---- the same with the READS sharded (SURVEY §8e option A): the compact matrix holds reads [read_begin,
// read_begin + n_slice) only, so a rank phases 1/world of the reads and the second exchange moves 1/world of the bytes.
// read_begin must be a multiple of 256 (a slice starts on a 128-byte line of every column).
int jl_xwin_assemble_slice_local(jl_ctx *pc, jl_ctx *const *windows, uint32_t n_windows, const jl_variant *merged, uint32_t n_var,
                                 uint64_t read_begin, uint64_t n_slice, jl_variant *remapped, uint32_t *pos_global,
                                 uint32_t *vp_total)
{
    if (!pc || !windows || n_windows == 0 || (!merged && n_var) || !vp_total) return JL_ERR_ARG;
    if (n_var > JL_VARIANT_CAP) return jl_fail(pc, JL_ERR_OVERFLOW, "%u variants, table holds %u", n_var, JL_VARIANT_CAP);
    xw_layout lay;
    int rc = xw_check_windows(pc, windows, n_windows, &lay);
    if (rc) return rc;
    const uint64_t n_reads = windows[0]->n_reads;
    if ((n_slice && (read_begin & 255u)) || read_begin > n_reads || n_slice > n_reads - read_begin)
        return jl_fail(pc, JL_ERR_ARG, "slice [%llu, +%llu) of %llu reads: the start must be a multiple of 256 and the slice inside",
                       (unsigned long long)read_begin, (unsigned long long)n_slice, (unsigned long long)n_reads);
    const uint64_t sb[2] = {read_begin, read_begin + n_slice};
    xwin_schedule sch;
    std::string err;
    rc = xwin_make_schedule(lay.win_begin.data(), lay.win_ncols.data(), lay.win_rank.data(), n_windows, merged, n_var, sb, 1, remapped, &sch, &err);
    if (rc) return jl_fail(pc, rc, "%s", err.c_str());
    const uint32_t vp = (uint32_t)sch.pos.size();
    *vp_total = vp;
    if (pos_global) std::copy(sch.pos.begin(), sch.pos.end(), pos_global);
    if (vp == 0 || n_slice == 0) return JL_OK;
    JL_HIP(pc, hipSetDevice(pc->device));
    rc = jl_msa_alloc(pc, n_slice, 3u * vp, 0);
    if (rc) return rc;
    for (uint32_t w = 0; w < n_windows; ++w) JL_HIP(pc, hipStreamSynchronize(windows[w]->stream));
    xw_send_buf none;
    rc = xw_pack(pc, windows, lay, sch, sb, 1, 0, nullptr, &none, nullptr, nullptr);
    if (rc) return rc;
    JL_HIP(pc, hipStreamSynchronize(pc->stream));
    return JL_OK;
}

// One window per rank, reads sharded: rank s phases reads [slice_begin[s], slice_begin[s + 1]) (world + 1 entries, the
// same on every rank, multiples of 256 except the last = n_reads).  The schedule is jl_xwin_slice_plan's: one pack
// launch, then one packed ncclSend per peer and one ncclRecv per owning peer in a single group.
int jl_xwin_assemble_slice_rccl(jl_ctx *pc, jl_ctx *window, jl_comm *c, const uint32_t *win_begin, const uint32_t *win_ncols,
                                const jl_variant *merged, uint32_t n_var, const uint64_t *slice_begin, jl_variant *remapped,
                                uint32_t *pos_global, uint32_t *vp_total)
{
    if (!pc || !window || !c || !win_begin || !win_ncols || (!merged && n_var) || !slice_begin || !vp_total) return JL_ERR_ARG;
    if (!window->d_msa) return jl_fail(pc, JL_ERR_ARG, "window has no resident matrix");
    if (n_var > JL_VARIANT_CAP) return jl_fail(pc, JL_ERR_OVERFLOW, "%u variants, table holds %u", n_var, JL_VARIANT_CAP);
    const int world = c->world, me = c->rank;
    if (slice_begin[world] > window->n_reads)
        return jl_fail(pc, JL_ERR_ARG, "the slices cover %llu reads, the window holds %llu", (unsigned long long)slice_begin[world],
                       (unsigned long long)window->n_reads);
    xw_layout lay;
    lay.win_begin.assign(win_begin, win_begin + world);
    lay.win_ncols.assign(win_ncols, win_ncols + world);
    for (int r = 0; r < world; ++r) lay.win_rank.push_back(r);
    lay.first_local = (uint32_t)me;
    xwin_schedule sch;
    std::string err;
    int rc = xwin_make_schedule(lay.win_begin.data(), lay.win_ncols.data(), lay.win_rank.data(), (uint32_t)world, merged, n_var, slice_begin,
                                world, remapped, &sch, &err);
    if (rc) return jl_fail(pc, rc, "%s", err.c_str());
    const uint32_t vp = (uint32_t)sch.pos.size();
    *vp_total = vp;
    if (pos_global) std::copy(sch.pos.begin(), sch.pos.end(), pos_global);
    if (vp == 0) return JL_OK;
    JL_HIP(pc, hipSetDevice(pc->device));
    const uint64_t n_mine = slice_begin[me + 1] - slice_begin[me];
    // (allocation comes before anything is issued: a rank that fails here has issued nothing its peers wait for in THIS
    // call; they find out at their next collective — DESIGN.md (e) says what a one-sided failure does)
    if (n_mine && (rc = jl_msa_alloc(pc, n_mine, 3u * vp, 0))) return rc;
    JL_HIP(pc, hipStreamSynchronize(window->stream));
    xw_send_buf send;
    jl_ctx *wins[1] = {window};
    // the communicator is taken whatever its size: the call is refused alike on one rank and on eight
    if (jl_comm_direct_begin(c) != JL_OK)
        return jl_fail(pc, JL_ERR_STATE, "the communicator has asynchronous exchanges queued or uncollected: collect them first");
    rc = xw_pack(pc, wins, lay, sch, slice_begin, world, me, nullptr, &send, nullptr, nullptr);
    if (rc == JL_OK && world > 1) rc = xw_send_recv(pc, c, sch, slice_begin, send, nullptr);
    jl_comm_direct_end(c);
    const hipError_t e = hipStreamSynchronize(pc->stream);
    if (send.d) hipFree(send.d);
    if (rc) return rc;
    if (e != hipSuccess) return jl_fail(pc, JL_ERR_DEVICE, "column exchange: %s", hipGetErrorString(e));
    return JL_OK;
}

}  // extern "C"

/* ---------------------------------------------------------------- the session */

#define JL_XW_TAB_MAGIC 0x4A4C5854u   // head of a rank's table block
#define JL_XW_GCAP0 1024u             // groups a rank's export block holds before it has to grow
#define JL_XW_XROWS0 256u             // variant rows a rank's table block holds before it has to grow

struct jl_xwin {
    int device = 0, rank = 0, world = 1;
    jl_comm *comm = nullptr;
    std::vector<jl_ctx *> wins;
    xw_layout lay;
    std::vector<uint64_t> slice_begin;
    uint64_t n_reads = 0, n_mine = 0;
    jl_ctx *pc = nullptr;            // the compact matrix of this rank's slice and its phasing buffers
    std::string err;

    // group tables: one block per rank = jl_exp_head + counts[gcap] + patterns[gcap][pstride]
    uint32_t gcap = 0, pstride = 0;
    size_t blk = 0;
    uint8_t *h_blk = nullptr;        // pinned [world][blk]: where the host reads every rank's groups
    uint8_t *d_blk_send = nullptr, *d_blk_recv = nullptr;   // world > 1: this rank's block in HBM, the gathered ones
    // variant tables (world > 1): one block per rank = 16-byte head + rows[xrows]
    uint32_t xrows = 0;
    size_t tblk = 0;
    uint8_t *h_tsend = nullptr, *h_trecv = nullptr, *d_tsend = nullptr, *d_trecv = nullptr;
    xw_send_buf send;
    uint16_t *h_tab = nullptr, *d_tab = nullptr;   // haplotype of each exported group, for more than JL_XW_TAB_MAX groups
    size_t tab_cap = 0;

    // results (host), valid until the next call
    std::vector<std::vector<jl_variant>> scratch;
    std::vector<jl_variant> merged, remapped;
    xwin_schedule sch;
    std::vector<uint8_t> mpat, hap_pattern, hit;
    std::vector<uint64_t> mcount;
    std::vector<uint32_t> index_me, hap_count, cooc, pos_cols;
    std::vector<uint16_t> hap_of_merged, tab;
    jl_phase_summary summary{};
    uint32_t n_haplotypes = 0, n_merged_groups = 0, my_groups = 0, bits = 4;
    bool ids_on_device = false;
    float stage_us[JL_XWIN_STAGES] = {};   // host time of the last call, by stage (jl_xwin_stage_us)
};

namespace {

int xs_fail(jl_xwin *x, int status, const std::string &msg)
{
    x->err = msg;
    return status;
}

int xs_fail_ctx(jl_xwin *x, int status, jl_ctx *c)
{
    x->err = jl_last_error(c);
    return status;
}

void xs_free_blocks(jl_xwin *x)
{
    if (x->h_blk) hipHostFree(x->h_blk);
    if (x->d_blk_send) hipFree(x->d_blk_send);
    if (x->d_blk_recv) hipFree(x->d_blk_recv);
    x->h_blk = x->d_blk_send = x->d_blk_recv = nullptr;
    x->blk = 0;
}

// room for `gcap` groups of `pstride` pattern bytes per rank
int xs_reserve_blocks(jl_xwin *x, uint32_t gcap, uint32_t pstride)
{
    if (x->gcap >= gcap && x->pstride >= pstride && x->blk) return JL_OK;
    gcap = std::max(gcap, x->gcap);
    pstride = std::max(pstride, x->pstride);
    xs_free_blocks(x);
    const size_t blk = (sizeof(jl_exp_head) + (size_t)gcap * 4u + (size_t)gcap * pstride + 15u) / 16u * 16u;
    bool ok = hipHostMalloc(&x->h_blk, blk * (size_t)x->world, hipHostMallocDefault) == hipSuccess;
    if (ok && x->comm)
        ok = hipMalloc(&x->d_blk_send, blk) == hipSuccess && hipMalloc(&x->d_blk_recv, blk * (size_t)x->world) == hipSuccess;
    if (!ok) {
        xs_free_blocks(x);
        return xs_fail(x, JL_ERR_MEMORY, "group-table blocks");
    }
    x->gcap = gcap;
    x->pstride = pstride;
    x->blk = blk;
    return JL_OK;
}

void xs_free_tables(jl_xwin *x)
{
    if (x->h_tsend) hipHostFree(x->h_tsend);
    if (x->h_trecv) hipHostFree(x->h_trecv);
    if (x->d_tsend) hipFree(x->d_tsend);
    if (x->d_trecv) hipFree(x->d_trecv);
    x->h_tsend = x->h_trecv = x->d_tsend = x->d_trecv = nullptr;
    x->tblk = 0;
}

int xs_reserve_tables(jl_xwin *x, uint32_t xrows)
{
    if (x->xrows >= xrows && x->tblk) return JL_OK;
    xs_free_tables(x);
    const size_t tblk = 16u + (size_t)xrows * sizeof(jl_variant);
    const bool ok = hipHostMalloc(&x->h_tsend, tblk, hipHostMallocDefault) == hipSuccess &&
                    hipHostMalloc(&x->h_trecv, tblk * (size_t)x->world, hipHostMallocDefault) == hipSuccess &&
                    hipMalloc(&x->d_tsend, tblk) == hipSuccess && hipMalloc(&x->d_trecv, tblk * (size_t)x->world) == hipSuccess;
    if (!ok) {
        xs_free_tables(x);
        return xs_fail(x, JL_ERR_MEMORY, "variant-table blocks");
    }
    x->xrows = xrows;
    x->tblk = tblk;
    return JL_OK;
}

// `bytes` of HBM into pinned host memory and a completion word behind them, then wait for the word
int xs_fetch_and_wait(jl_xwin *x, const void *d_src, void *h_dst, size_t bytes)
{
    jl_ctx *pc = x->pc;
    jl_launch_xw_fetch(d_src, h_dst, bytes, pc->d_sync + 6, pc->d_sync, pc->h_seq, pc->stream);
    if (hipGetLastError() != hipSuccess) return xs_fail(x, JL_ERR_DEVICE, "fetch launch failed");
    pc->runs_launched++;
    if (jl_run_wait_seq(pc, pc->runs_launched)) return xs_fail_ctx(x, JL_ERR_DEVICE, pc);
    return JL_OK;
}

// world > 1: every rank's rows (global columns) to every rank; ONE all-gather of fixed-stride blocks, a second one only
// when some rank's table did not fit (every rank sees every head and takes the same branch)
int xs_gather_tables(jl_xwin *x, const std::vector<const jl_variant *> &rows, const std::vector<uint32_t> &counts)
{
    jl_ctx *pc = x->pc;
    uint32_t mine = 0;
    for (uint32_t n : counts) mine += n;
    int rc = xs_reserve_tables(x, std::max<uint32_t>(JL_XW_XROWS0, x->xrows));
    if (rc) return rc;
    for (int attempt = 0; attempt < 2; ++attempt) {
        uint32_t *head = reinterpret_cast<uint32_t *>(x->h_tsend);
        head[0] = JL_XW_TAB_MAGIC; head[1] = mine; head[2] = 0; head[3] = 0;
        if (mine <= x->xrows) {
            jl_variant *dst = reinterpret_cast<jl_variant *>(x->h_tsend + 16);
            for (size_t w = 0; w < rows.size(); ++w)
                for (uint32_t r = 0; r < counts[w]; ++r) {
                    *dst = rows[w][r];
                    dst->col += x->lay.win_begin[x->lay.first_local + w];
                    ++dst;
                }
        }
        const size_t used = 16u + (size_t)std::min(mine, x->xrows) * sizeof(jl_variant);
        if (hipMemcpyAsync(x->d_tsend, x->h_tsend, used, hipMemcpyHostToDevice, pc->stream) != hipSuccess)
            return xs_fail(x, JL_ERR_DEVICE, "table upload");
        if (jl_tp_allgather(x->comm, x->d_tsend, x->d_trecv, x->tblk, pc->stream) != JL_OK)
            return xs_fail(x, JL_ERR_COMM, "all-gather of the variant tables: " + x->comm->tp_error);
        if ((rc = xs_fetch_and_wait(x, x->d_trecv, x->h_trecv, x->tblk * (size_t)x->world))) return rc;
        uint32_t most = 0;
        uint64_t total = 0;
        for (int r = 0; r < x->world; ++r) {
            const uint32_t *h = reinterpret_cast<const uint32_t *>(x->h_trecv + x->tblk * (size_t)r);
            if (h[0] != JL_XW_TAB_MAGIC) return xs_fail(x, JL_ERR_COMM, "rank " + std::to_string(r) + " sent no variant table");
            most = std::max(most, h[1]);
            total += h[1];
        }
        if (most <= x->xrows) {
            x->merged.resize(total ? total : 1);
            std::vector<const jl_variant *> tabs((size_t)x->world);
            std::vector<uint32_t> cnt((size_t)x->world), zero((size_t)x->world, 0);
            for (int r = 0; r < x->world; ++r) {
                tabs[(size_t)r] = reinterpret_cast<const jl_variant *>(x->h_trecv + x->tblk * (size_t)r + 16);
                cnt[(size_t)r] = reinterpret_cast<const uint32_t *>(x->h_trecv + x->tblk * (size_t)r)[1];
            }
            uint32_t n = 0;
            rc = jl_merge_tables(tabs.data(), cnt.data(), zero.data(), (uint32_t)x->world, x->merged.data(), (uint32_t)x->merged.size(), &n);
            if (rc) return xs_fail(x, rc, "merge of the variant tables");
            x->merged.resize(n);
            return JL_OK;
        }
        // some rank called more rows than a block holds: every rank grows to the same size and gathers again
        if ((rc = xs_reserve_tables(x, (most + 127u) / 128u * 128u))) return rc;
    }
    return xs_fail(x, JL_ERR_COMM, "the ranks disagree about the size of the variant tables");
}

}  // namespace

extern "C" {

int jl_xwin_create(jl_ctx *const *windows, uint32_t n_local, jl_comm *comm, const uint32_t *win_begin, const uint32_t *win_ncols,
                   const int32_t *win_rank, uint32_t n_windows, const uint64_t *slice_begin, jl_xwin **out)
{
    if (!out) return JL_ERR_ARG;
    *out = nullptr;
    if (!windows || !n_local || !win_begin || !win_ncols || !win_rank || !n_windows || !slice_begin) return JL_ERR_ARG;
    jl_xwin *x = new (std::nothrow) jl_xwin();
    if (!x) return JL_ERR_MEMORY;
    x->comm = comm;
    x->world = comm ? comm->world : 1;
    x->rank = comm ? comm->rank : 0;
    x->device = windows[0] ? windows[0]->device : 0;
    auto bail = [&](int st) { jl_xwin_destroy(x); return st; };
    if (comm && comm->device != x->device) return bail(JL_ERR_ARG);
    x->lay.win_begin.assign(win_begin, win_begin + n_windows);
    x->lay.win_ncols.assign(win_ncols, win_ncols + n_windows);
    x->lay.win_rank.assign(win_rank, win_rank + n_windows);
    uint32_t mine = 0, first = n_windows;
    for (uint32_t w = 0; w < n_windows; ++w) {
        if (win_rank[w] < 0 || win_rank[w] >= x->world || (w && (win_rank[w] < win_rank[w - 1] || win_begin[w] < win_begin[w - 1]))) return bail(JL_ERR_ARG);
        if (win_rank[w] == x->rank) { ++mine; first = std::min(first, w); }
    }
    if (mine != n_local) return bail(JL_ERR_ARG);
    x->lay.first_local = first;
    for (uint32_t w = 0; w < n_local; ++w) {
        jl_ctx *c = windows[w];
        if (!c || !c->d_msa || c->device != x->device || c->n_reads != windows[0]->n_reads || c->col_stride != windows[0]->col_stride ||
            c->win_begin != win_begin[first + w] || c->n_cols != win_ncols[first + w])
            return bail(JL_ERR_ARG);
        x->wins.push_back(c);
    }
    x->n_reads = windows[0]->n_reads;
    x->slice_begin.assign(slice_begin, slice_begin + x->world + 1);
    for (int s = 0; s < x->world; ++s)
        if (slice_begin[s + 1] < slice_begin[s] || (slice_begin[s + 1] > slice_begin[s] && (slice_begin[s] & 255u))) return bail(JL_ERR_ARG);
    if (slice_begin[0] != 0 || slice_begin[x->world] != x->n_reads) return bail(JL_ERR_ARG);
    x->n_mine = slice_begin[x->rank + 1] - slice_begin[x->rank];
    x->scratch.resize(n_local);
    if (hipSetDevice(x->device) != hipSuccess) return bail(JL_ERR_DEVICE);
    // the session's own context (assembled columns, grouping, ids) orders its work on the first window's stream: the
    // session's launches come after that window's tables are on the host anyway, and a stream of its own is a hardware
    // queue the runtime takes 8 ms to create — a sixth of `juliet --windows 8` on a 100k-read BAM
    int rc = jl_ctx_create(x->device, windows[0]->stream, &x->pc);
    if (rc) return bail(rc);
    *out = x;
    return JL_OK;
}

void jl_xwin_destroy(jl_xwin *x)
{
    if (!x) return;
    hipSetDevice(x->device);
    if (x->pc) jl_ctx_destroy(x->pc);
    xs_free_blocks(x);
    xs_free_tables(x);
    if (x->send.d) hipFree(x->send.d);
    if (x->h_tab) hipHostFree(x->h_tab);
    if (x->d_tab) hipFree(x->d_tab);
    delete x;
}

const char *jl_xwin_last_error(const jl_xwin *x) { return x ? x->err.c_str() : ""; }

// Before an exchange every rank says how it fares: an all-gather of {magic, status, group-block shape}.  A rank that failed
// on its own between two collectives (an allocation, a launch) still takes part in THIS one, so its peers leave the step
// with an error instead of waiting inside the exchange for a message that never comes; ranks whose group blocks have
// different shapes (their histories diverged after an earlier local error) stop here too instead of decoding each other's
// blocks with the wrong layout.  16 bytes per rank through the table blocks, which are free at this point.
#define JL_XW_AGREE_MAGIC 0x4A4C4147u
static int xs_agree(jl_xwin *x, int local_rc)
{
    jl_ctx *pc = x->pc;
    if (!x->d_tsend || !x->d_trecv || x->tblk < 16u) return local_rc ? local_rc : xs_fail(x, JL_ERR_STATE, "no table blocks for the status exchange");
    uint32_t *h = reinterpret_cast<uint32_t *>(x->h_tsend);
    h[0] = JL_XW_AGREE_MAGIC; h[1] = (uint32_t)(-local_rc); h[2] = x->gcap; h[3] = x->pstride;
    // (a rank whose upload fails still takes part in the all-gather — with whatever its block holds: no magic word, which ends the
    // step on every rank — instead of leaving its peers inside a collective it never joins)
    const bool uploaded = hipMemcpyAsync(x->d_tsend, x->h_tsend, 16, hipMemcpyHostToDevice, pc->stream) == hipSuccess;
    if (!uploaded) (void)hipMemsetAsync(x->d_tsend, 0, 16, pc->stream);
    if (jl_tp_allgather(x->comm, x->d_tsend, x->d_trecv, 16, pc->stream) != JL_OK)
        return xs_fail(x, JL_ERR_COMM, "status exchange before the column slices: " + x->comm->tp_error);
    if (!uploaded) return xs_fail(x, JL_ERR_DEVICE, "status upload");
    if (int rc = xs_fetch_and_wait(x, x->d_trecv, x->h_trecv, 16u * (size_t)x->world)) return rc;
    for (int r = 0; r < x->world; ++r) {
        const uint32_t *q = reinterpret_cast<const uint32_t *>(x->h_trecv + 16u * (size_t)r);
        if (q[0] != JL_XW_AGREE_MAGIC) return xs_fail(x, JL_ERR_COMM, "rank " + std::to_string(r) + " sent no status");
        if (q[1] != 0u)
            return local_rc ? local_rc
                            : xs_fail(x, JL_ERR_COMM, "rank " + std::to_string(r) + " failed before the exchange (status -" + std::to_string(q[1]) + "): the step ends on every rank");
        if (q[2] != x->gcap || q[3] != x->pstride)
            return xs_fail(x, JL_ERR_COMM, "rank " + std::to_string(r) + " holds group blocks of another shape (" + std::to_string(q[2]) + " x " +
                                               std::to_string(q[3]) + " against " + std::to_string(x->gcap) + " x " + std::to_string(x->pstride) + ")");
    }
    return local_rc;
}

int jl_xwin_phase_sharded(jl_xwin *x, uint32_t min_reads, jl_xwin_result *out)
{
    if (!x || !out) return JL_ERR_ARG;
    memset(out, 0, sizeof *out);
    x->err.clear();
    if (hipSetDevice(x->device) != hipSuccess) return xs_fail(x, JL_ERR_DEVICE, "hipSetDevice failed");
    jl_ctx *pc = x->pc;
    const int world = x->world, me = x->rank;
    // with a communicator every collective is issued, also when it has one rank (all but the wire, on one GPU)
    const bool collective = x->comm != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    memset(x->stage_us, 0, sizeof x->stage_us);
    auto lap = [&](int stage) {
        const auto t = std::chrono::steady_clock::now();
        x->stage_us[stage] += std::chrono::duration<float, std::micro>(t - t_prev).count();
        t_prev = t;
    };

    // ---- 1. this rank's tables: in pinned memory already when the windows ran through jl_run_async / a group run
    std::vector<const jl_variant *> rows(x->wins.size());
    std::vector<uint32_t> counts(x->wins.size());
    for (size_t w = 0; w < x->wins.size(); ++w)
        if (int rc = jl_ctx_table_host(x->wins[w], &x->scratch[w], &rows[w], &counts[w])) return xs_fail_ctx(x, rc, x->wins[w]);

    lap(0);
    struct guard_t {   // the communicator is this thread's for the rest of the call
        jl_comm *c;
        ~guard_t() { if (c) jl_comm_direct_end(c); }
    } guard{nullptr};
    if (collective) {
        if (jl_comm_direct_begin(x->comm) != JL_OK)
            return xs_fail(x, JL_ERR_STATE, "the communicator has asynchronous exchanges queued or uncollected: collect them first");
        guard.c = x->comm;
    }

    // ---- 2. one table for the whole reference
    if (!collective) {
        uint64_t total = 0;
        for (uint32_t n : counts) total += n;
        x->merged.resize(total ? total : 1);
        uint32_t n = 0;
        int rc = jl_merge_tables(rows.data(), counts.data(), x->lay.win_begin.data() + x->lay.first_local, (uint32_t)rows.size(),
                                 x->merged.data(), (uint32_t)x->merged.size(), &n);
        if (rc) return xs_fail(x, rc, "merge of the variant tables");
        x->merged.resize(n);
    } else if (int rc = xs_gather_tables(x, rows, counts)) {
        return rc;
    }
    const uint32_t n_var = (uint32_t)x->merged.size();
    lap(1);

    // ---- 3. plan: positions, owners, the exchange
    x->remapped.resize(n_var ? n_var : 1);
    std::string err;
    int rc = xwin_make_schedule(x->lay.win_begin.data(), x->lay.win_ncols.data(), x->lay.win_rank.data(), (uint32_t)x->lay.win_begin.size(),
                                x->merged.data(), n_var, x->slice_begin.data(), world, x->remapped.data(), &x->sch, &err);
    if (rc) return xs_fail(x, rc, err);
    const uint32_t vp = (uint32_t)x->sch.pos.size();
    x->pos_cols.resize(vp);
    for (uint32_t k = 0; k < vp; ++k) x->pos_cols[k] = 3u * k;

    out->n_variants = n_var;
    out->merged = x->merged.data();
    out->n_positions = vp;
    out->pos_global = x->sch.pos.data();
    out->slice_begin = x->slice_begin[(size_t)me];
    out->slice_reads = x->n_mine;
    x->ids_on_device = false;
    x->n_haplotypes = 0;
    x->bits = 4;
    memset(&x->summary, 0, sizeof x->summary);
    lap(2);
    if (vp == 0) {   // nothing to phase (every rank sees the same table and leaves here together)
        out->read_hap_bits = 4;
        return JL_OK;
    }

    // ---- 4 + 5. column slices into the compact matrix, keys + grouping, the groups exported
    const uint32_t kwords = (vp + JL_POS_PER_WORD - 1) / JL_POS_PER_WORD;
    if ((rc = xs_reserve_blocks(x, std::max<uint32_t>(JL_XW_GCAP0, x->gcap), (kwords * JL_POS_PER_WORD + 7u) / 8u * 8u))) return rc;
    for (int attempt = 0;; ++attempt) {
        uint8_t *blk_dev = collective ? x->d_blk_send : x->h_blk;   // where THIS rank's kernels write its block
        // everything that can fail on this rank alone comes first; with peers the verdict is shared before anything is sent
        int local_rc = JL_OK;
        if (x->n_mine) {
            if ((local_rc = jl_msa_alloc(pc, x->n_mine, 3u * vp, 0))) xs_fail_ctx(x, local_rc, pc);
            else if ((local_rc = jl_phase_groups_prepare(pc, vp))) xs_fail_ctx(x, local_rc, pc);
        }
        if (!local_rc && world > 1) local_rc = xw_reserve_send(pc, x->sch, x->slice_begin.data(), world, me, &x->send, &x->err);
        if (collective && world > 1) {
            if ((rc = xs_agree(x, local_rc))) return rc;
        } else if (local_rc) {
            return local_rc;
        }
        if (x->n_mine) {
            pc->exp_ext_head = reinterpret_cast<uint32_t *>(blk_dev);
            pc->exp_ext_count = reinterpret_cast<uint32_t *>(blk_dev + sizeof(jl_exp_head));
            pc->exp_ext_pattern = blk_dev + sizeof(jl_exp_head) + (size_t)x->gcap * 4u;
            pc->exp_ext_cap = x->gcap;
            pc->exp_ext_stride = x->pstride;
        }
        // Every position in a window of THIS device and one key word: the phase launch reads the columns where they lie
        // (no compact matrix, no pack launch, no plan in front).  Otherwise: pack (+ the plan), then the peers' slices.
        const bool direct = x->n_mine && vp <= JL_POS_PER_WORD && x->sch.k_count[(size_t)me] == vp && world == 1;
        if (direct) {
            jl_direct_cols &d = pc->direct;
            memset(&d, 0, sizeof d);
            const uint64_t src_stride = x->wins[0]->plane_stride;
            for (uint32_t k = 0; k < vp; ++k) {
                const uint32_t w = (uint32_t)x->sch.owner_win[k];
                d.col[k] = x->wins[w - x->lay.first_local]->d_msa + (uint64_t)(x->sch.pos[k] - x->lay.win_begin[w]) * 3u * src_stride +
                           x->slice_begin[(size_t)me] / 8u;
            }
            d.stride = src_stride;
            d.vp = vp;
            d.on = 1;
        } else {
            jl_xw_pack_args plan;
            memset(&plan, 0, sizeof plan);
            if (x->n_mine) {
                plan.meta = pc->d_meta; plan.vpcols = pc->d_vpcols; plan.col2pos = pc->d_col2pos;
                plan.vp_total = vp; plan.kwords = kwords; plan.n_var = std::min<uint32_t>(n_var, JL_VARIANT_CAP);
            }
            if ((rc = xw_pack(pc, x->wins.data(), x->lay, x->sch, x->slice_begin.data(), world, me, nullptr, &x->send,
                              x->n_mine ? &plan : nullptr, &x->err)))
                return rc;
            if (world > 1 && (rc = xw_send_recv(pc, x->comm, x->sch, x->slice_begin.data(), x->send, &x->err))) return rc;
        }
        lap(3);
        if (x->n_mine) {
            // the selection of this launch ends a "run" of pc when nothing else follows on the device (no gather): its
            // last workgroup stores the completion word behind the exported block
            jl_launch_phase(pc, pc->stream, 0xFFFFFFFFu, true, false, !collective);
            if (hipGetLastError() != hipSuccess) return xs_fail(x, JL_ERR_DEVICE, "phase launch failed");
            if (!collective) {
                pc->runs_launched++;
                if (jl_run_wait_seq(pc, pc->runs_launched)) return xs_fail_ctx(x, JL_ERR_DEVICE, pc);
            }
        } else if (collective) {
            // a rank without reads takes part in the gather with an empty table
            if (hipMemsetAsync(x->d_blk_send, 0, sizeof(jl_exp_head), pc->stream) != hipSuccess) return xs_fail(x, JL_ERR_DEVICE, "memset");
        } else {
            memset(x->h_blk, 0, sizeof(jl_exp_head));
        }
        if (collective) {
            if (jl_tp_allgather(x->comm, x->d_blk_send, x->d_blk_recv, x->blk, pc->stream) != JL_OK)
                return xs_fail(x, JL_ERR_COMM, "all-gather of the group tables: " + x->comm->tp_error);
            if ((rc = xs_fetch_and_wait(x, x->d_blk_recv, x->h_blk, x->blk * (size_t)world))) return rc;
        }
        lap(4);
        uint32_t most = 0;
        bool ovf = false;
        for (int r = 0; r < world; ++r) {
            const jl_exp_head *h = reinterpret_cast<const jl_exp_head *>(x->h_blk + x->blk * (size_t)r);
            most = std::max(most, h->n_groups);
            ovf = ovf || h->overflow || h->n_groups > x->gcap;
        }
        if (!ovf) break;
        if (attempt) return xs_fail(x, JL_ERR_OVERFLOW, std::to_string(most) + " groups of reads on one rank");
        // a slice produced more groups than a block holds: every rank sees the heads, grows alike and runs the step again
        uint32_t g = x->gcap;
        while (g < most) g *= 2u;
        if ((rc = xs_reserve_blocks(x, g, x->pstride))) return rc;
    }

    // ---- 6. merge + selection on the merged counts (host; a few hundred rows)
    std::vector<const uint8_t *> pats((size_t)world);
    std::vector<const uint32_t *> cnts((size_t)world);
    std::vector<uint32_t> strides((size_t)world, x->pstride), ng((size_t)world);
    std::vector<jl_phase_summary> partials((size_t)world);
    uint64_t total_groups = 0;
    for (int r = 0; r < world; ++r) {
        const uint8_t *b = x->h_blk + x->blk * (size_t)r;
        const jl_exp_head *h = reinterpret_cast<const jl_exp_head *>(b);
        ng[(size_t)r] = h->n_groups;
        cnts[(size_t)r] = reinterpret_cast<const uint32_t *>(b + sizeof(jl_exp_head));
        pats[(size_t)r] = b + sizeof(jl_exp_head) + (size_t)x->gcap * 4u;
        jl_phase_summary p;
        memset(&p, 0, sizeof p);
        p.damaged_reads = h->damaged; p.marginal_gap = h->gap; p.marginal_heteroduplex = h->heteroduplex; p.marginal_partial = h->partial;
        partials[(size_t)r] = p;
        total_groups += h->n_groups;
    }
    x->my_groups = ng[(size_t)me];
    x->mpat.resize((size_t)std::max<uint64_t>(1, total_groups) * vp);
    x->mcount.resize((size_t)std::max<uint64_t>(1, total_groups));
    x->index_me.resize(std::max<uint32_t>(1, x->my_groups));
    std::vector<uint32_t *> index((size_t)world, nullptr);
    index[(size_t)me] = x->index_me.data();
    uint32_t m = 0;
    rc = jl_merge_groups(pats.data(), strides.data(), cnts.data(), ng.data(), (uint32_t)world, vp, x->mpat.data(), x->mcount.data(),
                         (uint32_t)x->mcount.size(), &m, index.data());
    if (rc) return xs_fail(x, rc, "merge of the group tables");
    x->n_merged_groups = m;
    x->hap_of_merged.resize(std::max<uint32_t>(1, m));
    x->hap_count.assign(JL_MAX_HAPLOTYPES, 0);
    x->hap_pattern.assign((size_t)JL_MAX_HAPLOTYPES * vp, 0);
    const uint32_t hcap = std::min<uint32_t>(JL_MAX_HAPLOTYPES, std::max<uint32_t>(1, m));
    x->hit.assign((size_t)std::max<uint32_t>(1, n_var) * hcap, 0);
    x->cooc.assign((size_t)std::max<uint32_t>(1, n_var) * std::max<uint32_t>(1, n_var), 0);
    rc = jl_select_haplotypes(x->mpat.data(), x->mcount.data(), m, vp, x->remapped.data(), n_var, x->pos_cols.data(), min_reads,
                              partials.data(), (uint32_t)world, &x->summary, x->hap_count.data(), x->hap_pattern.data(), x->hit.data(),
                              hcap, x->cooc.data(), x->hap_of_merged.data());
    if (rc) return xs_fail(x, rc, "selection of the haplotypes");
    const uint32_t H = x->summary.n_haplotypes;
    x->n_haplotypes = H;
    if (H != hcap && n_var) {   // hit rows closed up to [n_var][H]
        for (uint32_t v = 1; v < n_var; ++v) memmove(x->hit.data() + (size_t)v * H, x->hit.data() + (size_t)v * hcap, H);
    }
    x->bits = H <= JL_ID4_MAX_H ? 4u : (H <= JL_ID8_MAX_H ? 8u : 16u);

    lap(5);
    // ---- 7. the merge's answer back to the device: per-read ids of this rank's slice (they stay in HBM)
    if (x->n_mine) {
        x->tab.resize(std::max<uint32_t>(1, x->my_groups));
        for (uint32_t q = 0; q < x->my_groups; ++q) x->tab[q] = x->hap_of_merged[x->index_me[q]];
        jl_xw_assign_args a;
        memset(&a, 0, sizeof a);
        a.n_dwords = pc->col_stride / 4u;
        a.flagw = pc->d_flagw; a.read_slot = pc->d_read_slot; a.slot_hap = pc->d_slot_hap; a.read_hap = pc->d_read_hap;
        a.n_groups = x->my_groups; a.bits = x->bits; a.phased = 1u;
        // no completion word: nothing of this launch is read by the host before jl_xwin_read_hap_fetch, which waits for
        // the stream, and the next step's launches on this stream come behind it — the ids of step k are written while
        // the windows' next call stage is already running
        a.arrive = pc->d_sync + 6; a.seq_dev = pc->d_sync; a.seq_host = nullptr;
        const uint16_t *d_tab = nullptr;
        if (x->my_groups > JL_XW_TAB_MAX) {
            if (x->tab_cap < x->my_groups) {
                if (x->h_tab) hipHostFree(x->h_tab);
                if (x->d_tab) hipFree(x->d_tab);
                x->h_tab = x->d_tab = nullptr;
                x->tab_cap = 0;
                if (hipHostMalloc(&x->h_tab, (size_t)x->my_groups * 2u, hipHostMallocDefault) != hipSuccess ||
                    hipMalloc(&x->d_tab, (size_t)x->my_groups * 2u) != hipSuccess)
                    return xs_fail(x, JL_ERR_MEMORY, "haplotype table of the groups");
                x->tab_cap = x->my_groups;
            }
            memcpy(x->h_tab, x->tab.data(), (size_t)x->my_groups * 2u);
            if (hipMemcpyAsync(x->d_tab, x->h_tab, (size_t)x->my_groups * 2u, hipMemcpyHostToDevice, pc->stream) != hipSuccess)
                return xs_fail(x, JL_ERR_DEVICE, "haplotype table upload");
            d_tab = x->d_tab;
        }
        jl_launch_xw_assign(&a, x->tab.data(), d_tab, pc->stream);
        if (hipGetLastError() != hipSuccess) return xs_fail(x, JL_ERR_DEVICE, "id launch failed");
        x->ids_on_device = true;
    }
    lap(6);

    out->n_haplotypes = H;
    out->n_groups = m;
    out->summary = x->summary;
    out->hap_count = x->hap_count.data();
    out->hap_pattern = x->hap_pattern.data();
    out->hit = x->hit.data();
    out->cooc = x->cooc.data();
    out->read_hap_bits = x->bits;
    return JL_OK;
}

int jl_xwin_stage_us(const jl_xwin *x, float *out)
{
    if (!x || !out) return JL_ERR_ARG;
    memcpy(out, x->stage_us, sizeof x->stage_us);
    return JL_OK;
}

int jl_xwin_read_hap_fetch(jl_xwin *x, uint16_t *read_hap)
{
    if (!x || (!read_hap && x->n_mine)) return JL_ERR_ARG;
    if (!x->n_mine) return JL_OK;
    if (!x->ids_on_device) {   // nothing was phased: every read is "damaged", as in the one-window run
        for (uint64_t i = 0; i < x->n_mine; ++i) read_hap[i] = (uint16_t)JL_HAP_DAMAGED;
        return JL_OK;
    }
    if (hipSetDevice(x->device) != hipSuccess) return xs_fail(x, JL_ERR_DEVICE, "hipSetDevice failed");
    jl_ctx *pc = x->pc;
    const size_t bytes = x->bits == 16 ? (size_t)x->n_mine * 2u : (x->bits == 8 ? (size_t)x->n_mine : (size_t)(x->n_mine + 1) / 2u);
    // through the context's pinned block, a megabyte at a time (a runtime copy into pageable memory takes milliseconds the
    // first time a process makes one: 7 of the 30 ms `juliet --windows 8` spent behind the upload)
    std::vector<uint8_t> tmp(bytes);
    const size_t readable = (size_t)pc->col_stride * 2u * 2u;   // d_read_hap holds 16 bits for every read of the padded stride
    for (size_t o = 0; o < bytes; o += pc->h_scratch_cap) {
        const size_t n = std::min(bytes - o, pc->h_scratch_cap);
        if (int rc = jl_fetch_to_host(pc, reinterpret_cast<const uint8_t *>(pc->d_read_hap) + o, n, tmp.data() + o, readable - o))
            return xs_fail_ctx(x, rc, pc);
    }
    jl_expand_ids(tmp.data(), x->bits, x->n_mine, read_hap);
    return JL_OK;
}

// The group-table collective on its own (hosts that drive the stages themselves).  Not a hot path: the block is staged
// through the host and the device buffers live for the call.
int jl_allgather_groups(jl_ctx *ctx, jl_comm *c, uint32_t cap_groups, uint32_t pattern_stride, uint8_t *patterns, uint32_t *counts,
                        uint32_t *n_groups, jl_phase_summary *partials, uint32_t *n_positions)
{
    if (!ctx || !c || !cap_groups || !pattern_stride || !patterns || !counts || !n_groups) return JL_ERR_ARG;
    if (ctx->device != c->device) return jl_fail(ctx, JL_ERR_ARG, "context and communicator are on different devices");
    const int world = c->world;
    const size_t blk = (sizeof(jl_exp_head) + (size_t)cap_groups * 4u + (size_t)cap_groups * pattern_stride + 15u) / 16u * 16u;
    std::vector<uint8_t> mine(blk, 0), all(blk * (size_t)world);
    jl_exp_head *h = reinterpret_cast<jl_exp_head *>(mine.data());
    uint32_t ng = 0, vp = 0;
    jl_phase_summary part;
    memset(&part, 0, sizeof part);
    int rc = jl_phase_groups_fetch(ctx, nullptr, 0, nullptr, 0, &ng, &vp, nullptr, 0, &part);
    if (rc == JL_OK && ng <= cap_groups && vp <= pattern_stride)
        rc = jl_phase_groups_fetch(ctx, mine.data() + sizeof(jl_exp_head) + (size_t)cap_groups * 4u, pattern_stride,
                                   reinterpret_cast<uint32_t *>(mine.data() + sizeof(jl_exp_head)), cap_groups, &ng, &vp, nullptr, 0, &part);
    // whatever happened locally the collective is issued: the head tells the peers
    h->n_groups = ng; h->vp = vp;
    h->overflow = (rc != JL_OK || ng > cap_groups || vp > pattern_stride) ? 1u : 0u;
    h->damaged = part.damaged_reads; h->gap = part.marginal_gap; h->heteroduplex = part.marginal_heteroduplex;
    h->partial = part.marginal_partial; h->clean = part.insufficient_reads;
    const int local_rc = rc;
    JL_HIP(ctx, hipSetDevice(ctx->device));
    uint8_t *d_send = nullptr, *d_recv = nullptr;
    if (hipMalloc(&d_send, blk) != hipSuccess || hipMalloc(&d_recv, blk * (size_t)world) != hipSuccess) {
        if (d_send) hipFree(d_send);
        return jl_fail(ctx, JL_ERR_MEMORY, "buffers of the group-table exchange");
    }
    if (jl_comm_direct_begin(c) != JL_OK) {
        hipFree(d_send);
        hipFree(d_recv);
        return jl_fail(ctx, JL_ERR_STATE, "the communicator has asynchronous exchanges queued or uncollected: collect them first");
    }
    hipError_t e = hipMemcpyAsync(d_send, mine.data(), blk, hipMemcpyHostToDevice, ctx->stream);
    int r = JL_OK;
    if (e == hipSuccess) r = jl_tp_allgather(c, d_send, d_recv, blk, ctx->stream);
    jl_comm_direct_end(c);
    if (e == hipSuccess && r == JL_OK) e = hipMemcpyAsync(all.data(), d_recv, blk * (size_t)world, hipMemcpyDeviceToHost, ctx->stream);
    const hipError_t e2 = hipStreamSynchronize(ctx->stream);
    hipFree(d_send);
    hipFree(d_recv);
    if (r != JL_OK) return jl_fail(ctx, JL_ERR_COMM, "all-gather of the group tables: %s", c->tp_error.c_str());
    if (e != hipSuccess || e2 != hipSuccess) return jl_fail(ctx, JL_ERR_DEVICE, "group-table exchange: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    if (local_rc != JL_OK && local_rc != JL_ERR_OVERFLOW) return local_rc;
    bool ovf = false;
    for (int k = 0; k < world; ++k) {
        const uint8_t *b = all.data() + blk * (size_t)k;
        const jl_exp_head *hk = reinterpret_cast<const jl_exp_head *>(b);
        n_groups[k] = hk->n_groups;
        if (hk->overflow) { ovf = true; continue; }
        memcpy(counts + (size_t)k * cap_groups, b + sizeof(jl_exp_head), (size_t)hk->n_groups * 4u);
        memcpy(patterns + (size_t)k * cap_groups * pattern_stride, b + sizeof(jl_exp_head) + (size_t)cap_groups * 4u,
               (size_t)hk->n_groups * pattern_stride);
        if (partials) {
            jl_phase_summary p;
            memset(&p, 0, sizeof p);
            p.damaged_reads = hk->damaged; p.marginal_gap = hk->gap; p.marginal_heteroduplex = hk->heteroduplex;
            p.marginal_partial = hk->partial; p.insufficient_reads = hk->clean; p.n_positions = hk->vp;
            partials[k] = p;
        }
        if (n_positions && hk->vp) *n_positions = hk->vp;
    }
    if (ovf) return jl_fail(ctx, JL_ERR_OVERFLOW, "a rank exported more than %u groups (or patterns wider than %u)", cap_groups, pattern_stride);
    return JL_OK;
}

}  // extern "C"
