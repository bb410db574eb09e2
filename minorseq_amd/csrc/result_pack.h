// result_pack.h — block-level device routines shared by the kernels that END a run: the fixed-size result block
// (device copy = all-gather source, pinned host mirror) and the completion word (see jl_run_wait).
#pragma once
#include "jl_internal.h"

// Words that other workgroups of the SAME launch update (group counts and keys, the occupied list, the read-category
// counters) are read past this CU's L1 with agent-scope loads: memory-side atomics do not refresh a copy another
// XCD's L2 may still hold.
__device__ __forceinline__ uint32_t jl_ld_coherent(const uint32_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long jl_ld_coherent64(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Completion word of a run: stored by ONE thread after everything the run wrote for the host has been drained by
// its writers and a block barrier; `seq_host` is pinned host memory.
__device__ __forceinline__ void jl_signal_done(uint32_t *seq_dev, volatile uint32_t *seq_host)
{
    const uint32_t v = __hip_atomic_load(seq_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    __hip_atomic_store(seq_dev, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ONE system-scope release: the store below carries it (write-back of this die's L2, wait, store).  A fence of its
    // own in front of a release store paid for the same write-back twice (about 2 us each in every run's tail).
    __hip_atomic_store(const_cast<uint32_t *>(seq_host), v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The same for a caller that read the run counter earlier (`seq_before`, loaded beside other loads it had to wait for
// anyway: one dependent round trip less at the very end of the run).
__device__ __forceinline__ void jl_signal_done_from(uint32_t seq_before, uint32_t *seq_dev, volatile uint32_t *seq_host)
{
    const uint32_t v = seq_before + 1u;
    __hip_atomic_store(seq_dev, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(const_cast<uint32_t *>(seq_host), v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Gathers the small results into one fixed-size block.  pk: device copy; pk2: pinned host mirror (may be null) —
// both written directly.  `variants_coherent`: the rows were written by another workgroup of this launch.
__device__ __forceinline__ void jl_result_pack_block(const jl_variant *__restrict__ variants, uint32_t n,
                                                     const jl_phase_meta *meta, uint32_t phasing,
                                                     const uint32_t *__restrict__ vpcols,
                                                     const uint32_t *__restrict__ hap_count,
                                                     const uint8_t *__restrict__ hap_pattern,
                                                     const uint8_t *__restrict__ hit, const uint32_t *__restrict__ cooc,
                                                     uint32_t cooc_cap, uint32_t cooc_ready, jl_pack *__restrict__ pk,
                                                     jl_pack *__restrict__ pk2, bool variants_coherent = false)
{
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    const uint32_t fits_call = n <= JL_PACK_MAX_VAR;
    uint32_t vp = 0, H = 0, nv = 0, ovf = 0, fits_phase = 0, cooc_fits = 0, id_bits = 16;
    if (phasing) {
        vp = jl_ld_coherent(&meta->vp); H = jl_ld_coherent(&meta->summary.n_haplotypes); nv = jl_ld_coherent(&meta->n_var);
        ovf = jl_ld_coherent(&meta->overflow);
        id_bits = jl_ld_coherent(&meta->id_bits);
        fits_phase = ovf == 0 && fits_call && vp <= JL_PACK_MAX_VP && H <= JL_PACK_MAX_HAP &&
                     H * vp <= JL_PACK_PATTERN_BYTES && nv * H <= JL_PACK_HIT_BYTES;
        cooc_fits = cooc_ready && nv <= JL_PACK_COOC_N;
    }
    jl_pack *dsts[2] = {pk, pk2};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        jl_pack *o = dsts[t];
        if (!o) continue;
        if (tid == 0) {
            o->magic = JL_PACK_MAGIC; o->nvar_total = n; o->fits_call = fits_call; o->fits_phase = fits_phase;
            o->phase_ran = phasing; o->overflow = ovf; o->vp = vp; o->H = H;
            o->nv_phase = nv; o->cooc_fits = cooc_fits; o->id_bits = id_bits;
            if (phasing) {   // the category counters were added to by other workgroups of this launch
                jl_phase_summary sm;
                sm.reported_reads = jl_ld_coherent(&meta->summary.reported_reads);
                sm.insufficient_reads = jl_ld_coherent(&meta->summary.insufficient_reads);
                sm.damaged_reads = jl_ld_coherent(&meta->summary.damaged_reads);
                sm.marginal_gap = jl_ld_coherent(&meta->summary.marginal_gap);
                sm.marginal_heteroduplex = jl_ld_coherent(&meta->summary.marginal_heteroduplex);
                sm.marginal_partial = jl_ld_coherent(&meta->summary.marginal_partial);
                sm.n_positions = jl_ld_coherent(&meta->summary.n_positions);
                sm.n_haplotypes = jl_ld_coherent(&meta->summary.n_haplotypes);
                o->summary = sm;
            }
        }
        if (fits_call)
            for (uint32_t i = tid; i < n * (uint32_t)(sizeof(jl_variant) / 8); i += nt) {
                const unsigned long long *src = reinterpret_cast<const unsigned long long *>(variants) + i;
                reinterpret_cast<unsigned long long *>(o->variants)[i] = variants_coherent ? jl_ld_coherent64(src) : *src;
            }
        if (fits_phase) {
            for (uint32_t i = tid; i < vp; i += nt) o->pos_cols[i] = vpcols[i];
            for (uint32_t i = tid; i < H; i += nt) o->hap_count[i] = hap_count[i];
            for (uint32_t i = tid; i < H * vp; i += nt)
                o->hap_pattern[i] = hap_pattern[(uint64_t)(i / vp) * JL_VARIANT_CAP + (i % vp)];
            for (uint32_t i = tid; i < nv * H; i += nt) o->hit[i] = hit[(uint64_t)(i / H) * JL_MAX_HAPLOTYPES + (i % H)];
            if (cooc_fits)
                for (uint32_t i = tid; i < nv * nv; i += nt) o->cooc[i] = cooc[(uint64_t)(i / nv) * cooc_cap + (i % nv)];
        }
    }
}

// The head of the result block this workgroup has just written (header + the rows in use) once more, to `xh`: the send part
// of a bound exchange.  Read back past the L1 (the stores went through to this die's L2); the caller fences.
__device__ __forceinline__ void jl_result_head_copy(const jl_pack *pk, uint8_t *xh)
{
    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(pk);
    unsigned long long *dst = reinterpret_cast<unsigned long long *>(xh);
    const uint32_t n = jl_ld_coherent(&pk->nvar_total);
    const uint32_t rows = n <= JL_PACK_MAX_VAR ? n : 0u;
    const uint32_t words = (uint32_t)(offsetof(jl_pack, variants) / 8u) + rows * (uint32_t)(sizeof(jl_variant) / 8u);
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) dst[i] = jl_ld_coherent64(src + i);
}

// Per-read ids in the narrowest code that holds the run's haplotype count (jl_internal.h: JL_ID4_MAX_H / JL_ID8_MAX_H):
// eight ids (16-bit codes: haplotype, JL_HAP_INSUFFICIENT, JL_HAP_DAMAGED) of reads 8t .. 8t+7 into the packed buffer
__device__ __forceinline__ void jl_store_ids(uint16_t *base, uint64_t t, const uint16_t (&h)[8], uint32_t bits)
{
    if (bits == 4u) {
        uint32_t v = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const uint32_t c = h[r] == JL_HAP_DAMAGED ? 15u : (h[r] == JL_HAP_INSUFFICIENT ? 14u : (uint32_t)h[r]);
            v |= c << (4 * r);
        }
        reinterpret_cast<uint32_t *>(base)[t] = v;
    } else if (bits == 8u) {
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const uint32_t c = h[r] == JL_HAP_DAMAGED ? 255u : (h[r] == JL_HAP_INSUFFICIENT ? 254u : (uint32_t)h[r]);
            if (r < 4) lo |= c << (8 * r);
            else hi |= c << (8 * (r - 4));
        }
        uint2 v;
        v.x = lo; v.y = hi;
        reinterpret_cast<uint2 *>(base)[t] = v;
    } else {
        uint4 v;
        v.x = h[0] | ((uint32_t)h[1] << 16); v.y = h[2] | ((uint32_t)h[3] << 16);
        v.z = h[4] | ((uint32_t)h[5] << 16); v.w = h[6] | ((uint32_t)h[7] << 16);
        // reads_pad = 2 * col_stride entries: the 16-byte store of a live lane is always inside the buffer
        reinterpret_cast<uint4 *>(base)[t] = v;
    }
}
