// xwin_schedule.h — host-only plan of cross-window phasing: the distinct variant positions of the merged
// table, the window and rank that hold each position's three columns, and the column-slice exchange that follows from
// them.  Shared by the plan entry points (jl_xwin_plan, jl_xwin_slice_plan) and the code that issues the exchange
// (capi_comm.hip, capi_xwin.hip), so that what is tested as data is what runs.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/juliet_hip.h"

// Distinct variant positions of the merged (global-column) table, ascending, and the remapped table whose columns
// index the compact matrix: position k lives in compact columns 3k..3k+2.
static inline uint32_t xwin_remap(const jl_variant *merged, uint32_t n_var, jl_variant *remapped, uint32_t *pos_global)
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

// the first window that holds columns col .. col + 2 entirely (-1: none)
static inline int xwin_owner(const uint32_t *win_begin, const uint32_t *win_ncols, uint32_t n_windows, uint32_t col)
{
    for (uint32_t w = 0; w < n_windows; ++w)
        if (col >= win_begin[w] && (uint64_t)col + 3 <= (uint64_t)win_begin[w] + win_ncols[w]) return (int)w;
    return -1;
}

struct xwin_schedule {
    std::vector<uint32_t> pos;         // [vp] global column of each position, ascending
    std::vector<int32_t> owner_win;    // [vp] window that holds it
    std::vector<uint32_t> k_begin, k_count;   // [world] the run of positions each rank owns
};

// `win_rank` non-decreasing (a rank holds consecutive windows), windows in ascending column order: the positions a rank
// owns are then one run of k.  `remapped` (optional) receives the table with col = 3k.
static inline int xwin_make_schedule(const uint32_t *win_begin, const uint32_t *win_ncols, const int32_t *win_rank, uint32_t n_windows,
                                     const jl_variant *merged, uint32_t n_var, const uint64_t *slice_begin, int32_t world,
                                     jl_variant *remapped, xwin_schedule *out, std::string *err)
{
    for (uint32_t w = 0; w < n_windows; ++w) {
        if (win_rank[w] < 0 || win_rank[w] >= world || (w && win_rank[w] < win_rank[w - 1])) {
            if (err) *err = "win_rank must be non-decreasing and inside the world";
            return JL_ERR_ARG;
        }
        if (w && win_begin[w] < win_begin[w - 1]) {
            if (err) *err = "windows must come in ascending column order";
            return JL_ERR_ARG;
        }
    }
    for (int32_t s = 0; s < world; ++s)
        if (slice_begin[s + 1] < slice_begin[s] || (slice_begin[s + 1] > slice_begin[s] && (slice_begin[s] & 255u))) {
            if (err) *err = "read slices must be ascending and start on multiples of 256 reads";
            return JL_ERR_ARG;
        }
    out->pos.assign(n_var ? n_var : 1, 0);
    const uint32_t vp = xwin_remap(merged, n_var, remapped, out->pos.data());
    out->pos.resize(vp);
    out->owner_win.assign(vp, -1);
    out->k_begin.assign((size_t)world, 0);
    out->k_count.assign((size_t)world, 0);
    int32_t prev_rank = -1;
    for (uint32_t k = 0; k < vp; ++k) {
        const int w = xwin_owner(win_begin, win_ncols, n_windows, out->pos[k]);
        if (w < 0) {
            if (err) *err = "variant column " + std::to_string(out->pos[k]) + " is not fully inside any window";
            return JL_ERR_ARG;
        }
        const int32_t r = win_rank[w];
        if (r < prev_rank) {   // cannot happen with ascending windows; a guard for layouts that overlap oddly
            if (err) *err = "the windows' order does not follow the columns";
            return JL_ERR_ARG;
        }
        out->owner_win[k] = w;
        if (out->k_count[(size_t)r] == 0) out->k_begin[(size_t)r] = k;
        out->k_count[(size_t)r]++;
        prev_rank = r;
    }
    return JL_OK;
}

// plane stride of a compact matrix of n_reads reads (= jl_plane_stride: whole 128-byte lines, 1024 reads each)
static inline uint64_t xwin_stride(uint64_t n_reads) { return (n_reads + 1023) / 1024 * 128; }

// ops of one rank in issue order: its own slice, then per peer (ascending) the send and the receive
static inline void xwin_ops_of_rank(const xwin_schedule &sch, const uint64_t *slice_begin, int32_t world, int32_t rank,
                                    std::vector<jl_xwin_op> *ops)
{
    ops->clear();
    auto make = [&](int32_t op, int32_t peer, int32_t owner, int32_t receiver) {
        jl_xwin_op o;
        o.op = op;
        o.peer = peer;
        o.k_begin = sch.k_begin[(size_t)owner];
        o.k_count = sch.k_count[(size_t)owner];
        o.read_begin = slice_begin[receiver];
        o.n_reads = slice_begin[receiver + 1] - slice_begin[receiver];
        o.dst_stride = xwin_stride(o.n_reads);
        o.bytes = 9ull * o.k_count * o.dst_stride;   // nine plane rows per position
        o.dst_offset = 9ull * o.k_begin * o.dst_stride;
        return o;
    };
    const uint64_t n_mine = slice_begin[rank + 1] - slice_begin[rank];
    if (sch.k_count[(size_t)rank] && n_mine) ops->push_back(make(JL_XWIN_OP_LOCAL, rank, rank, rank));
    for (int32_t s = 0; s < world; ++s) {
        if (s == rank) continue;
        const uint64_t n_s = slice_begin[s + 1] - slice_begin[s];
        if (sch.k_count[(size_t)rank] && n_s) ops->push_back(make(JL_XWIN_OP_SEND, s, rank, s));
        if (sch.k_count[(size_t)s] && n_mine) ops->push_back(make(JL_XWIN_OP_RECV, s, s, rank));
    }
}
