// phase_plan.h — distinct variant columns of the resident variant table, ascending (SPEC §8), as a block-level
// device routine so that both the stand-alone plan kernel and the tail of compact_kernel can run it.
#pragma once
#include "jl_internal.h"

// Call with all 1024 threads of a block.  Writes vpcols / col2pos / meta; `varcol` is scratch [n_cols].
// `fast_only`: the caller will only run the single-word (Vp <= 10) kernels; more positions set overflow bit 3
// and leave vp = 0 so that the following kernels do nothing and the host re-runs the generic pipeline.
__device__ __forceinline__ void jl_phase_plan_block(const jl_variant *__restrict__ variants, uint32_t nv,
                                                    uint32_t n_cols, uint8_t *__restrict__ varcol,
                                                    uint32_t *__restrict__ vpcols, uint32_t *__restrict__ col2pos,
                                                    uint32_t kwords_cap, uint32_t fast_only,
                                                    jl_phase_meta *__restrict__ meta)
{
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_running;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    for (uint32_t c = tid; c < n_cols; c += 1024u) varcol[c] = 0;
    if (tid == 0) s_running = 0;
    __syncthreads();
    for (uint32_t v = tid; v < nv; v += 1024u) {
        const uint32_t c = variants[v].col;
        if (c + 2u < n_cols) varcol[c] = 1;
    }
    __syncthreads();
    for (uint32_t base = 0; base < n_cols; base += 1024u) {
        const uint32_t c = base + tid;
        const uint32_t f = c < n_cols ? varcol[c] : 0u;
        uint32_t inc = f;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t u = __shfl_up(inc, o, 64);
            if ((int)lane >= o) inc += u;
        }
        if (lane == 63) s_wave[wid] = inc;
        __syncthreads();
        uint32_t off = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const uint32_t x = s_wave[w];
            if (w < (int)wid) off += x;
            total += x;
        }
        if (f) {
            const uint32_t p = s_running + off + inc - 1u;
            vpcols[p] = c;
            col2pos[c] = p;
        }
        __syncthreads();
        if (tid == 0) s_running += total;
        __syncthreads();
    }
    if (tid == 0) {
        uint32_t vp = s_running;
        uint32_t kw = (vp + JL_POS_PER_WORD - 1u) / JL_POS_PER_WORD;
        meta->n_var = nv;
        meta->vp_true = vp;
        meta->overflow = 0;
        if (kw > kwords_cap) {  // key buffer too small: skip, the host re-runs with the exact size
            meta->overflow = 4u;
            vp = 0;
            kw = 0;
        } else if (fast_only && kw > 1u) {
            meta->overflow = 8u;
            vp = 0;
            kw = 0;
        }
        meta->vp = vp;
        meta->kwords = kw;
        meta->n_occupied = 0;
        jl_phase_summary z = {0, 0, 0, 0, 0, 0, vp, 0};
        meta->summary = z;
    }
}
