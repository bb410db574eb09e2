// phase_plan.h — distinct variant columns of the resident variant table, ascending (SPEC §8), as a block-level
// device routine so that both the stand-alone plan kernel and the last block of call_kernel can run it.
#pragma once
#include "jl_internal.h"

#define JL_PLAN_LDS_WORDS 4096u   // column bitset kept in LDS: windows of up to 131072 columns

// Call with every thread of a block (any multiple of 64 up to 1024).  Writes vpcols / col2pos / meta.
// `fast_only` (1 or 2): the caller will only run the fused launch of that many key words (Vp <= 10 / 20); more positions set overflow bit 3
// and leave vp = 0 so that the following kernels do nothing and the host re-runs the generic pipeline.
// Windows of up to 131072 columns mark the variant columns in an LDS bitset and rank them by prefix popcounts
// (no global round trips besides the variant rows themselves, and not even those when the caller passes their
// columns in `lds_cols`); wider ones use `varcol` [n_cols] in HBM as scratch.
__device__ __forceinline__ void jl_phase_plan_block(const jl_variant *variants, uint32_t nv, uint32_t n_cols,
                                                    uint8_t *varcol, uint32_t *vpcols, uint32_t *col2pos,
                                                    uint32_t kwords_cap, uint32_t fast_only, jl_phase_meta *meta,
                                                    const uint32_t *lds_cols = nullptr)
{
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_running;
    __shared__ uint32_t s_bits[JL_PLAN_LDS_WORDS];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6, nt = blockDim.x, nw = nt >> 6;
    const bool in_lds = n_cols <= JL_PLAN_LDS_WORDS * 32u;
    const uint32_t n_words = (n_cols + 31u) / 32u;
    if (in_lds) {
        for (uint32_t w = tid; w < n_words; w += nt) s_bits[w] = 0;
    } else {
        for (uint32_t c = tid; c < n_cols; c += nt) varcol[c] = 0;
    }
    if (tid == 0) s_running = 0;
    __syncthreads();
    for (uint32_t v = tid; v < nv; v += nt) {
        const uint32_t c = lds_cols ? lds_cols[v] : variants[v].col;   // the caller may already hold the columns in LDS
        if (c + 2u < n_cols) {
            if (in_lds) atomicOr(&s_bits[c >> 5], 1u << (c & 31u));
            else varcol[c] = 1;
        }
    }
    __syncthreads();
    // rank of every marked column = exclusive prefix count, one item (a 32-column word / a column) per thread and round
    const uint32_t n_items = in_lds ? n_words : n_cols;
    for (uint32_t base = 0; base < n_items; base += nt) {
        const uint32_t i = base + tid;
        uint32_t bits = 0;
        if (i < n_items) bits = in_lds ? s_bits[i] : (uint32_t)varcol[i];
        const uint32_t f = in_lds ? (uint32_t)__popc(bits) : bits;
        uint32_t inc = f;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t u = __shfl_up(inc, o, 64);
            if ((int)lane >= o) inc += u;
        }
        if (lane == 63) s_wave[wid] = inc;
        __syncthreads();
        uint32_t off = 0, total = 0;
        for (uint32_t w = 0; w < nw; ++w) {
            const uint32_t x = s_wave[w];
            if (w < wid) off += x;
            total += x;
        }
        uint32_t p = s_running + off + inc - f;
        if (in_lds) {
            while (bits) {
                const uint32_t c = i * 32u + (uint32_t)__ffs((int)bits) - 1u;
                bits &= bits - 1u;
                vpcols[p] = c;
                col2pos[c] = p;
                ++p;
            }
        } else if (f) {
            vpcols[p] = i;
            col2pos[i] = p;
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
        } else if (fast_only && kw > fast_only) {   // fast_only = key words the fused launch in use covers (1 or 2)
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
