// kernels_phase.hip — read x variant co-occurrence phasing.
// Behaviour: doc/JULIET.md:192-211 (--mode-phasing, haplotype_hit, haplotype block), :253-254 (>= 10 reads),
// :372-381 (reported / insufficient / damaged, overlapping marginals); docs/SPEC.md §8.
//
// Stages (no host round trip):
//   plan    distinct variant columns, ascending, from the resident variant table (done by call_kernel's last block
//           in a whole-path run; phase_plan_kernel for a host-supplied table)
//   keys    per read: flags (gap / heteroduplex / partial) and the pattern of codon indices at the Vp
//           positions, 6 bits each, 10 positions per 64-bit word, first position most significant so that
//           word-wise unsigned comparison is the lexicographic order of patterns
//   group   exact grouping of clean reads into an open-addressing table
//   select  groups with >= min_reads ranked by (count desc, pattern asc), haplotype ids, patterns, hit matrix
//           + variant x variant co-occurrence, the result block, and emptying of the table slots used
//   assign  per-read haplotype id
// Vp <= 10 (one key word): ONE launch — phase_fused1_kernel (a window) / phase_group_run_kernel (several windows,
// blockIdx.z): keys + grouping in every block, the selection by the block that arrives last, and — when all
// workgroups of the launch are resident together (<= JL_FOLD_MAX_BLOCKS) — the per-read ids by every block from the slots still
// in its registers (the others wait on a flag); larger launches leave the ids to phase_assign(_group)_kernel.
// Vp > 10: phase_keys_kernel, phase_group_kernel, phase_select_kernel, phase_assign_kernel.
#include <string.h>

#include <algorithm>
#include <type_traits>

#include "call_eval.h"
#include "jl_internal.h"
#include "phase_plan.h"
#include "planes.h"
#include "result_pack.h"

namespace {

constexpr uint32_t kM1 = 0x11111111u;
constexpr uint32_t kEmpty = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t wave_sum_all(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// ---------------------------------------------------------------------------------------- plan
__global__ __launch_bounds__(1024) void phase_plan_kernel(const jl_variant *__restrict__ variants,
                                                           const uint32_t *__restrict__ n_rows, uint32_t cap,
                                                           uint32_t n_cols, uint8_t *__restrict__ varcol,
                                                           uint32_t *__restrict__ vpcols,
                                                           uint32_t *__restrict__ col2pos, uint32_t kwords_cap,
                                                           uint32_t fast_only, jl_phase_meta *__restrict__ meta)
{
    uint32_t nv = n_rows[0];
    if (nv > cap) nv = cap;
    jl_phase_plan_block(variants, nv, n_cols, varcol, vpcols, col2pos, kwords_cap, fast_only, meta);
}

// ---------------------------------------------------------------------------------------- keys
// One lane = 8 consecutive reads = one byte of each plane of every variant column, put back together as a dword of eight codes.
__global__ __launch_bounds__(256) void phase_keys_kernel(const uint8_t *__restrict__ msa, uint64_t plane_stride,
                                                          uint64_t n_reads, uint64_t reads_pad,
                                                          const uint32_t *__restrict__ vpcols,
                                                          jl_phase_meta *__restrict__ meta,
                                                          uint64_t *__restrict__ keys, uint32_t *__restrict__ flagw)
{
    const uint32_t vp = meta->vp;
    if (vp == 0) return;   // (a context on the multi-word pipeline takes it for every run, whatever the run's key width)
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;  // byte index within a plane
    const bool live = t < plane_stride;
    uint32_t gap = 0, het = 0, par = 0;
    const uint32_t kwords = meta->kwords;
    for (uint32_t gw = 0; gw < kwords; ++gw) {
        uint64_t key[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const uint32_t p0 = gw * JL_POS_PER_WORD;
        const uint32_t p1 = min(vp, p0 + JL_POS_PER_WORD);
        for (uint32_t p = p0; p < p1; ++p) {
            const uint32_t c = vpcols[p];
            uint32_t w[3];
#pragma unroll
            for (int k = 0; k < 3; ++k)
                w[k] = live ? jl_load_codes8(msa + (uint64_t)(c + k) * 3u * plane_stride + t, plane_stride) : 0x66666666u;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint32_t b0 = w[k] & kM1, b1 = (w[k] >> 1) & kM1, b2 = (w[k] >> 2) & kM1;
                gap |= b2 & ~b1 & ~b0;  // 4 = '-'
                het |= b2 & b0;         // 5 = 'N'
                par |= b2 & b1;         // 6 = not covered
            }
            const uint32_t hi2 = w[0] & 0x33333333u;
            const uint32_t lo4 = ((w[1] & 0x33333333u) << 2) | (w[2] & 0x33333333u);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const uint32_t code = (((hi2 >> (4 * r)) & 3u) << 4) | ((lo4 >> (4 * r)) & 15u);
                key[r] = (key[r] << 6) | code;
            }
        }
        if (live) {
            uint64_t *dst = keys + (uint64_t)gw * reads_pad + t * 8u;
#pragma unroll
            for (int r = 0; r < 8; r += 2) {
                ulonglong2 v;
                v.x = key[r];
                v.y = key[r + 1];
                *reinterpret_cast<ulonglong2 *>(dst + r) = v;
            }
        }
    }
    // reads beyond n_reads are padding, not damaged reads
    uint32_t valid = 0;
    if (live) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
            if (t * 8u + r < n_reads) valid |= 1u << (4 * r);
    }
    gap &= valid; het &= valid; par &= valid;
    if (live) flagw[t] = gap | (het << 1) | (par << 2) | ((valid ^ kM1) << 3);  // bit3: padding
    const uint32_t n_gap = wave_sum_all(__popc(gap));
    const uint32_t n_het = wave_sum_all(__popc(het));
    const uint32_t n_par = wave_sum_all(__popc(par));
    const uint32_t n_dam = wave_sum_all(__popc(gap | het | par));
    if ((threadIdx.x & 63u) == 0) {
        if (n_dam) atomicAdd(&meta->summary.damaged_reads, n_dam);
        if (n_gap) atomicAdd(&meta->summary.marginal_gap, n_gap);
        if (n_het) atomicAdd(&meta->summary.marginal_heteroduplex, n_het);
        if (n_par) atomicAdd(&meta->summary.marginal_partial, n_par);
    }
}

// ---------------------------------------------------------------------------------------- group
__device__ __forceinline__ bool keys_equal(const uint64_t *__restrict__ keys, uint64_t reads_pad, uint32_t kwords,
                                           uint64_t i, uint64_t j)
{
    for (uint32_t g = 0; g < kwords; ++g)
        if (keys[(uint64_t)g * reads_pad + i] != keys[(uint64_t)g * reads_pad + j]) return false;
    return true;
}

__global__ __launch_bounds__(256) void phase_group_kernel(uint64_t n_reads, uint64_t reads_pad,
                                                           const uint64_t *__restrict__ keys,
                                                           const uint32_t *__restrict__ flagw,
                                                           jl_phase_meta *__restrict__ meta, uint64_t slots_mask,
                                                           uint32_t *__restrict__ slot_rep,
                                                           uint32_t *__restrict__ slot_count,
                                                           uint32_t *__restrict__ occupied,
                                                           uint32_t *__restrict__ read_slot)
{
    const uint32_t kwords = meta->kwords;
    if (kwords == 0) return;  // nothing to phase
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    bool active = false;
    uint64_t k0 = 0;
    if (i < n_reads) {
        const uint32_t f = (flagw[i >> 3] >> (4u * (uint32_t)(i & 7u))) & 15u;
        active = f == 0;
        if (active) k0 = keys[i];
    }
    uint64_t todo = __ballot(active);
    while (todo) {  // wave-uniform: one iteration per distinct key in the wave
        const int leader = __ffsll((unsigned long long)todo) - 1;
        const uint64_t lk0 = __shfl(k0, leader, 64);
        const uint64_t li = __shfl(i, leader, 64);
        bool same = active && k0 == lk0;
        if (same && kwords > 1) {
            for (uint32_t g = 1; g < kwords; ++g)
                if (keys[(uint64_t)g * reads_pad + i] != keys[(uint64_t)g * reads_pad + li]) { same = false; break; }
        }
        const uint64_t grp = __ballot(same);
        uint32_t slot = 0;
        if ((int)lane == leader) {
            uint64_t h = mix64(k0 + 0x9E3779B97F4A7C15ull);
            for (uint32_t g = 1; g < kwords; ++g) h = mix64(h ^ keys[(uint64_t)g * reads_pad + i]);
            uint64_t s = h & slots_mask;
            const uint32_t cnt = (uint32_t)__popcll(grp);
            for (;;) {
                uint32_t rep = __hip_atomic_load(&slot_rep[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (rep == kEmpty) {
                    rep = atomicCAS(&slot_rep[s], kEmpty, (uint32_t)i);
                    if (rep == kEmpty) {
                        occupied[atomicAdd(&meta->n_occupied, 1u)] = (uint32_t)s;
                        rep = (uint32_t)i;
                    }
                }
                if (rep == (uint32_t)i || keys_equal(keys, reads_pad, kwords, rep, i)) {
                    atomicAdd(&slot_count[s], cnt);
                    break;
                }
                s = (s + 1u) & slots_mask;
            }
            slot = (uint32_t)s;
        }
        slot = __shfl(slot, leader, 64);
        if (same) {
            read_slot[i] = slot;
            active = false;
        }
        todo &= ~grp;
    }
}

__device__ __forceinline__ uint32_t ld_coherent(const uint32_t *p) { return jl_ld_coherent(p); }
__device__ __forceinline__ unsigned long long ld_coherent64(const unsigned long long *p) { return jl_ld_coherent64(p); }
__device__ __forceinline__ void signal_done(uint32_t *seq_dev, volatile uint32_t *seq_host) { jl_signal_done(seq_dev, seq_host); }

// width of the per-read ids for H reported haplotypes (jl_internal.h: JL_ID4_MAX_H / JL_ID8_MAX_H)
__device__ __forceinline__ uint32_t id_bits_for(uint32_t H) { return H <= JL_ID4_MAX_H ? 4u : (H <= JL_ID8_MAX_H ? 8u : 16u); }

__device__ __forceinline__ void store_ids(uint16_t *base, uint64_t t, const uint16_t (&h)[8], uint32_t bits) { jl_store_ids(base, t, h, bits); }

// ---------------------------------------------------------------------------------------- select
template <bool BYKEY>
__device__ __forceinline__ uint32_t pattern_code(const uint64_t *keys, unsigned long long *slot_key, uint64_t reads_pad,
                                                 uint32_t vp, uint32_t id, uint32_t p)
{
    const uint32_t g = p / JL_POS_PER_WORD;
    const uint32_t in_word = min(JL_POS_PER_WORD, vp - g * JL_POS_PER_WORD);
    const uint32_t sh = 6u * (in_word - 1u - (p - g * JL_POS_PER_WORD));
    const uint64_t k = BYKEY ? (uint64_t)ld_coherent64(&slot_key[id])
                             : keys[(uint64_t)g * reads_pad + id];
    return (uint32_t)(k >> sh) & 63u;
}

// Block-level (any block size that is a multiple of 64): groups with >= min_reads ranked by (count desc, pattern asc),
// haplotype ids, patterns, hit matrix, co-occurrence, the result block, and emptying of the table slots the run used.
// Every table / key / meta word it reads was written by other workgroups: the caller has acquired them.
// BYKEY: single-word pipeline — a group's pattern is its 64-bit table key (coherent: written by CAS), so no
// representative read has to be looked up in the key buffer another workgroup may have written in this launch.
#define JL_SELECT_LDS_WORDS (JL_CAND_CAP + 704u)

template <bool BYKEY>
__device__ __forceinline__ void phase_select_block(uint32_t min_reads, uint64_t reads_pad, const uint64_t *keys,
                                                   jl_phase_meta *meta, uint32_t *slot_rep, uint32_t *slot_count,
                                                   const uint32_t *occupied, uint32_t *slot_hap,
                                                   const jl_variant *variants, const uint32_t *col2pos, uint32_t n_cols,
                                                   uint32_t *hap_count, uint8_t *hap_pattern, uint8_t *hit,
                                                   const uint32_t *n_rows, const uint32_t *vpcols, uint32_t *cooc,
                                                   uint32_t cooc_cap, jl_pack *pk, jl_pack *mirror,
                                                   unsigned long long *slot_key, uint32_t *seq_dev, uint32_t *lds,
                                                   uint32_t *exp_count = nullptr, uint8_t *exp_pattern = nullptr,
                                                   uint32_t exp_cap = 0, uint32_t exp_stride = 0, uint32_t *cache = nullptr,
                                                   unsigned long long *key_cache = nullptr, uint32_t key_cache_words = 0,
                                                   uint32_t *exp_head = nullptr)
{
    // `lds`: JL_SELECT_LDS_WORDS words of LDS of the caller (the fused launch lends the tables its grouping is done with)
    uint32_t *s_cand = lds;                               // [JL_CAND_CAP] slot of each candidate
    uint32_t *s_hrep = lds + JL_CAND_CAP;                 // [JL_MAX_HAPLOTYPES]
    __shared__ uint32_t s_ncand, s_insufficient, s_reported, s_nhap;
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    const uint32_t vp = ld_coherent(&meta->vp);
    const uint32_t kwords = ld_coherent(&meta->kwords);
    const uint32_t nv = ld_coherent(&meta->n_var);
    const uint32_t n_occ = ld_coherent(&meta->n_occupied);
    if (vp != 0 && exp_count) {  // block-uniform: the groups go out as they are (see jl_select_args)
        if (tid == 0) s_insufficient = 0;
        __syncthreads();
        for (uint32_t q = tid; q < n_occ; q += nt) {
            const uint32_t s = ld_coherent(&occupied[q]);
            const uint32_t c = ld_coherent(&slot_count[s]);
            atomicAdd(&s_insufficient, c);
            // the slot remembers WHICH exported group it is: the merge answers per group (jl_phase_regroup, jl_xwin_phase_sharded)
            __hip_atomic_store(&slot_hap[s], q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (q < exp_cap) {
                exp_count[q] = c;
                // the group's key words once each, its row of codon codes stored 8 bytes at a time (exp_stride is a multiple
                // of 8: the rows may lie in pinned host memory, where every store is a transaction of its own)
                const uint32_t rep = BYKEY ? s : ld_coherent(&slot_rep[s]);
                uint64_t word = 0;
                uint32_t have = 0xFFFFFFFFu;
                for (uint32_t p8 = 0; p8 < vp; p8 += 8u) {
                    unsigned long long out = 0;
                    for (uint32_t j = 0; j < 8u && p8 + j < vp; ++j) {
                        const uint32_t p = p8 + j, g = p / JL_POS_PER_WORD;
                        if (g != have) {
                            word = BYKEY ? (uint64_t)ld_coherent64(&slot_key[s]) : keys[(uint64_t)g * reads_pad + rep];
                            have = g;
                        }
                        const uint32_t in_word = min(JL_POS_PER_WORD, vp - g * JL_POS_PER_WORD);
                        const uint32_t sh = 6u * (in_word - 1u - (p - g * JL_POS_PER_WORD));
                        out |= (unsigned long long)((word >> sh) & 63u) << (8u * j);
                    }
                    *reinterpret_cast<unsigned long long *>(exp_pattern + (uint64_t)q * exp_stride + p8) = out;
                }
            }
        }
        __syncthreads();
        if (tid == 0) {
            if (n_occ > exp_cap) meta->overflow |= 16u;
            meta->summary.reported_reads = 0;
            meta->summary.insufficient_reads = s_insufficient;   // clean reads: the merge decides which are reported
            meta->summary.n_haplotypes = 0;
            __hip_atomic_store(&meta->id_bits, 16u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (exp_head) {   // the scalars a merge needs, next to the groups (pinned host memory or an all-gather's send buffer)
                exp_head[0] = n_occ; exp_head[1] = vp; exp_head[2] = n_occ > exp_cap ? 1u : 0u;
                exp_head[3] = ld_coherent(&meta->summary.damaged_reads); exp_head[4] = ld_coherent(&meta->summary.marginal_gap);
                exp_head[5] = ld_coherent(&meta->summary.marginal_heteroduplex); exp_head[6] = ld_coherent(&meta->summary.marginal_partial);
                exp_head[7] = s_insufficient;
            }
        }
    } else if (vp != 0) {  // block-uniform
    if (tid == 0) { s_ncand = 0; s_insufficient = 0; s_reported = 0; s_nhap = 0; }
    __syncthreads();
    for (uint32_t q = tid; q < n_occ; q += nt) {
        const uint32_t s = ld_coherent(&occupied[q]);
        const uint32_t c = ld_coherent(&slot_count[s]);
        if (c >= min_reads) {
            const uint32_t k = atomicAdd(&s_ncand, 1u);
            if (k < JL_CAND_CAP) s_cand[k] = s;
            else { atomicAdd(&s_insufficient, c); __hip_atomic_store(&slot_hap[s], (uint32_t)JL_HAP_INSUFFICIENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        } else {
            atomicAdd(&s_insufficient, c);
            __hip_atomic_store(&slot_hap[s], (uint32_t)JL_HAP_INSUFFICIENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    uint32_t ncand = s_ncand;
    if (ncand > JL_CAND_CAP) {
        if (tid == 0) meta->overflow |= 1u;
        ncand = JL_CAND_CAP;
    }
    // `cache` (2 * JL_CAND_CAP words of LDS, the general launch has them): every candidate's count and representative
    // once, instead of a trip past the L1 per comparison (300 candidates: 104 us of a 234 us run)
    uint32_t *s_ccnt = cache, *s_crep = cache ? cache + JL_CAND_CAP : nullptr;
    if (cache) {
        for (uint32_t a = tid; a < ncand; a += nt) {
            const uint32_t sa = s_cand[a];
            s_ccnt[a] = ld_coherent(&slot_count[sa]);
            s_crep[a] = BYKEY ? sa : ld_coherent(&slot_rep[sa]);
        }
        __syncthreads();
    }
    // ... and, when they fit, the candidates' keys: equal counts are common among the small groups, and every tie was two
    // more trips to the key buffer
    const bool keys_lds = !BYKEY && cache && key_cache && (uint64_t)ncand * kwords <= key_cache_words;
    if (keys_lds) {
        for (uint32_t q = tid; q < ncand * kwords; q += nt) {
            const uint32_t a = q / kwords, g = q - a * kwords;
            key_cache[q] = keys[(uint64_t)g * reads_pad + s_crep[a]];
        }
        __syncthreads();
    }
    // rank sort: (count desc, pattern asc); patterns are unique so ranks are a permutation
    for (uint32_t a = tid; a < ncand; a += nt) {
        const uint32_t sa = s_cand[a], ca = cache ? s_ccnt[a] : ld_coherent(&slot_count[sa]);
        const uint32_t ra = cache ? s_crep[a] : (BYKEY ? sa : ld_coherent(&slot_rep[sa]));
        uint32_t rank = 0;
        if (cache && (BYKEY || keys_lds)) {
            // everything the comparison needs sits in LDS: one pass without a trip to memory
            const uint64_t ka0 = BYKEY ? ld_coherent64(&slot_key[sa]) : key_cache[a * kwords];
            // eight candidates per trip, count and first key word of each read unconditionally (independent LDS reads in
            // flight: one at a time behind a data-dependent branch was 380 cycles per candidate, and equal counts are the
            // rule among small groups); only equal counts AND equal first words go on to the remaining words
            // (the caches arrive as generic pointers: typed as LDS here, or every read is a flat load)
            typedef __attribute__((address_space(3))) const uint32_t lds_u32;
            typedef __attribute__((address_space(3))) const unsigned long long lds_u64;
            lds_u32 *l_ccnt = (lds_u32 *)s_ccnt;
            lds_u32 *l_cand = (lds_u32 *)s_cand;
            lds_u64 *l_key = (lds_u64 *)key_cache;
            for (uint32_t b0 = 0; b0 < ncand; b0 += 8u) {
                uint32_t cb[8];
                uint64_t kb[8];
#pragma unroll
                for (uint32_t j = 0; j < 8u; ++j) {
                    const uint32_t b = b0 + j < ncand ? b0 + j : a;
                    cb[j] = l_ccnt[b];
                    kb[j] = BYKEY ? ld_coherent64(&slot_key[l_cand[b]]) : l_key[b * kwords];
                }
                uint32_t ties = 0;
#pragma unroll
                for (uint32_t j = 0; j < 8u; ++j) {
                    const bool live = b0 + j < ncand && b0 + j != a;
                    rank += (live && (cb[j] > ca || (cb[j] == ca && kb[j] < ka0))) ? 1u : 0u;
                    ties |= (live && cb[j] == ca && kb[j] == ka0) ? 1u << j : 0u;
                }
                while (!BYKEY && ties) {   // same count, same first ten positions: the later words decide
                    const uint32_t b = b0 + (uint32_t)__ffs((int)ties) - 1u;
                    ties &= ties - 1u;
                    for (uint32_t g = 1; g < kwords; ++g) {
                        const uint64_t ka = key_cache[a * kwords + g], kbg = key_cache[b * kwords + g];
                        if (kbg != ka) { rank += kbg < ka ? 1u : 0u; break; }
                    }
                }
            }
        } else
        for (uint32_t b = 0; b < ncand; ++b) {
            if (b == a) continue;
            const uint32_t sb = s_cand[b], cb = cache ? s_ccnt[b] : ld_coherent(&slot_count[sb]);
            if (cb > ca) { ++rank; continue; }
            if (cb < ca) continue;
            if (BYKEY) {
                const uint64_t ka = ld_coherent64(&slot_key[sa]);
                const uint64_t kb = ld_coherent64(&slot_key[sb]);
                if (kb < ka) ++rank;
                continue;
            }
            const uint32_t rb = cache ? s_crep[b] : ld_coherent(&slot_rep[sb]);
            for (uint32_t g = 0; g < kwords; ++g) {
                const uint64_t ka = keys[(uint64_t)g * reads_pad + ra], kb = keys[(uint64_t)g * reads_pad + rb];
                if (kb != ka) { if (kb < ka) ++rank; break; }
            }
        }
        if (rank < JL_MAX_HAPLOTYPES) {
            hap_count[rank] = ca;
            s_hrep[rank] = ra;  // BYKEY: the slot, else a read carrying the pattern
            __hip_atomic_store(&slot_hap[sa], rank, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // write-through: read by other workgroups
            atomicAdd(&s_reported, ca);
            atomicAdd(&s_nhap, 1u);
        } else {
            __hip_atomic_store(&slot_hap[sa], (uint32_t)JL_HAP_INSUFFICIENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicAdd(&s_insufficient, ca);
        }
    }
    __syncthreads();
    const uint32_t H = s_nhap;
    if (tid == 0) {
        if (ncand > JL_MAX_HAPLOTYPES) meta->overflow |= 2u;
        meta->summary.reported_reads = s_reported;
        meta->summary.insufficient_reads = s_insufficient;
        meta->summary.n_haplotypes = H;
        __hip_atomic_store(&meta->id_bits, id_bits_for(H), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (uint32_t q = tid; q < H * vp; q += nt) {
        const uint32_t h = q / vp, p = q - h * vp;
        hap_pattern[(uint64_t)h * JL_VARIANT_CAP + p] = (uint8_t)pattern_code<BYKEY>(keys, slot_key, reads_pad, vp, s_hrep[h], p);
    }
    // With the cache (free again after the ranking) the co-occurrence sums run out of LDS: the haplotype counts and, per
    // variant, the set of haplotypes that carry it as a bit mask — filled by the hit loop below.  (Summing over two
    // global hit rows per pair was most of this launch at 49 variants x 125 haplotypes: 98 us.)
    const uint32_t nvc_l = nv < cooc_cap ? nv : cooc_cap;
    constexpr uint32_t kHapWords = (JL_MAX_HAPLOTYPES + 31u) / 32u;
    const bool cooc_lds = cache && (uint64_t)nvc_l * kHapWords + 1024u <= 2u * JL_CAND_CAP;
    uint32_t *s_hc = cache, *s_bits = cache ? cache + 1024 : nullptr;   // [JL_MAX_HAPLOTYPES <= 1024], [nvc][kHapWords]
    if (cooc_lds) {
        __syncthreads();   // every thread is done with the ranking's use of the cache
        for (uint32_t q = tid; q < nvc_l * kHapWords; q += nt) s_bits[q] = 0;
        __syncthreads();
    }
    for (uint32_t q = tid; q < nv * H; q += nt) {
        const uint32_t v = q / H, h = q - v * H;
        // BYKEY: the table (and col2pos) may come from another workgroup of this launch
        const uint32_t c = BYKEY ? ld_coherent(&variants[v].col) : variants[v].col;
        uint8_t x = 0;
        if (c + 2u < n_cols) {  // rows outside this window never hit
            const uint32_t p = BYKEY ? ld_coherent(&col2pos[c]) : col2pos[c];
            // ref_codon / codon / flags share one dword
            const uint32_t cw = BYKEY ? ld_coherent(reinterpret_cast<const uint32_t *>(&variants[v]) + 3) : 0u;
            const uint32_t codon = BYKEY ? ((cw >> 8) & 0xFFu) : (uint32_t)variants[v].codon;
            x = pattern_code<BYKEY>(keys, slot_key, reads_pad, vp, s_hrep[h], p) == codon;
        }
        hit[(uint64_t)v * JL_MAX_HAPLOTYPES + h] = x;
        if (cooc_lds && x && v < nvc_l) atomicOr(&s_bits[v * kHapWords + (h >> 5)], 1u << (h & 31u));
    }
    __syncthreads();
    // co-occurrence over the reported haplotypes, for the first cooc_cap variants
    if (cooc_lds) {
        for (uint32_t h = tid; h < H; h += nt) s_hc[h] = hap_count[h];
        __syncthreads();
        for (uint32_t q = tid; q < nvc_l * nvc_l; q += nt) {
            const uint32_t v = q / nvc_l, w = q - v * nvc_l;
            uint32_t sum = 0;
            for (uint32_t j = 0; j < (H + 31u) / 32u; ++j) {
                uint32_t m = s_bits[v * kHapWords + j] & s_bits[w * kHapWords + j];
                while (m) {
                    sum += s_hc[32u * j + (uint32_t)__ffs((int)m) - 1u];
                    m &= m - 1u;
                }
            }
            cooc[(uint64_t)v * cooc_cap + w] = sum;
        }
    } else {
        const uint32_t nvc = nvc_l;
        for (uint32_t q = tid; q < nvc * nvc; q += nt) {
            const uint32_t v = q / nvc, w = q - v * nvc;
            uint32_t sum = 0;
            for (uint32_t h = 0; h < H; ++h)
                if (hit[(uint64_t)v * JL_MAX_HAPLOTYPES + h] && hit[(uint64_t)w * JL_MAX_HAPLOTYPES + h]) sum += hap_count[h];
            cooc[(uint64_t)v * cooc_cap + w] = sum;
        }
    }
    } else if (tid == 0) {
        __hip_atomic_store(&meta->id_bits, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // nothing phased: every read is "damaged"
        if (exp_head) {
            exp_head[0] = 0; exp_head[1] = 0; exp_head[2] = 0;
            exp_head[3] = 0; exp_head[4] = 0; exp_head[5] = 0; exp_head[6] = 0; exp_head[7] = 0;
        }
    }
    __syncthreads();
    // device copy double-buffered by the parity of the run index (an exchange may still read run n's block while
    // run n+1 writes its own); an exporting run has no result block: its groups ARE the result
    if (!exp_count) {
        pk += __hip_atomic_load(seq_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u;
        jl_result_pack_block(variants, ld_coherent(&n_rows[0]), meta, 1u, vpcols, hap_count, hap_pattern, hit, cooc, cooc_cap, 1u, pk,
                             mirror, BYKEY);
    }
    // leave the table empty for the next run: only the slots this run touched
    for (uint32_t q = tid; q < n_occ; q += nt) {
        const uint32_t s = ld_coherent(&occupied[q]);
        slot_key[s] = ~0ull;
        slot_rep[s] = 0xFFFFFFFFu;
        slot_count[s] = 0;
    }
}

__global__ __launch_bounds__(1024) void phase_select_kernel(uint32_t min_reads, uint64_t reads_pad, const uint64_t *keys,
                                                             jl_phase_meta *meta, uint32_t *slot_rep,
                                                             uint32_t *slot_count, const uint32_t *occupied,
                                                             uint32_t *slot_hap, const jl_variant *variants,
                                                             const uint32_t *col2pos, uint32_t n_cols,
                                                             uint32_t *hap_count, uint8_t *hap_pattern, uint8_t *hit,
                                                             const uint32_t *n_rows, const uint32_t *vpcols,
                                                             uint32_t *cooc, uint32_t cooc_cap, jl_pack *pk,
                                                             jl_pack *mirror, unsigned long long *slot_key,
                                                             uint32_t *seq_dev, volatile uint32_t *seq_host,
                                                             uint32_t *exp_count, uint8_t *exp_pattern, uint32_t exp_cap,
                                                             uint32_t exp_stride, uint32_t *exp_head)
{
    __shared__ uint32_t s_select[JL_SELECT_LDS_WORDS];
    __shared__ uint32_t s_cache[2u * JL_CAND_CAP];
    __shared__ unsigned long long s_keys[2048];
    phase_select_block<false>(min_reads, reads_pad, keys, meta, slot_rep, slot_count, occupied, slot_hap, variants, col2pos,
                              n_cols, hap_count, hap_pattern, hit, n_rows, vpcols, cooc, cooc_cap, pk, mirror, slot_key, seq_dev,
                              s_select, exp_count, exp_pattern, exp_cap, exp_stride, s_cache, s_keys, 2048u, exp_head);
    if (seq_host) {  // last kernel of the run: the result block is on its way to the host
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) signal_done(seq_dev, seq_host);
    }
}

typedef jl_select_args select_args;

// ---------------------------------------------------------------------------------------- fused keys + group
// Vp <= 10: the whole pattern is ONE 64-bit word, so the table can be keyed by value (64-bit CAS, no
// representative lookup) and a block can aggregate before it touches HBM:
//   0. (whole-path runs) every workgroup derives the variant columns from the call masks the Fisher stage left
//      (a few KB, L2-resident) — no launch or workgroup hand-off between calling and phasing; one EXTRA workgroup
//      meanwhile compacts the called rows into the ordered table,
//   1. every lane builds the keys and flags of its 8 reads (30 independent dword loads in flight),
//   2. the block's dominant key (that of its first clean read — the wild type for all but pathological
//      inputs) is counted with popcount-style compares and costs ONE global insert per block,
//   3. the remaining clean reads go through an LDS table (CAS on the 64-bit key), which is then flushed
//      with one global insert per distinct key per block.
// Global atomics on the hot slot drop from one per wave to one per 2048 reads.
#ifdef JL_EXP_STAMPS   // experiment builds only: device-clock stamps (100 MHz) of one workgroup's way through the fused launch
__device__ unsigned long long g_stamps[64];
#define JL_STAMP(k) do { if (threadIdx.x == 0 && (blockIdx.x == 0 || (k) >= 8)) g_stamps[(k)] = wall_clock64(); } while (0)
#define JL_STAMP_ON 1
#else
#define JL_STAMP(k) do { } while (0)
#define JL_STAMP_ON 0
__device__ unsigned long long g_stamps[32];   // (never written in this build: the stores sit behind JL_STAMP_ON)
#endif
constexpr uint32_t kLdsSlots = 1024;
constexpr uint64_t kNoKey = ~0ull;
constexpr uint32_t kPlanList = 256;   // called positions a workgroup can rank in LDS; more take the multi-word pipeline

struct plan_state {   // LDS: what the prologue derives — identical in every workgroup of a window
    uint32_t n_list, n_rows, vp, ovf;
    uint32_t cols[2u * JL_POS_PER_WORD];   // the variant columns, ascending (first 20: two key words)
    uint32_t list[kPlanList];
    uint8_t first[kPlanList];
};

// Distinct start columns of the called positions, ascending, from the call masks (SPEC §8).  All 256 threads.
__device__ __forceinline__ void plan_prologue(const select_args &S, uint32_t n_cols, plan_state &L)
{
    const uint32_t tid = threadIdx.x;
    if (tid == 0) { L.n_list = 0; L.n_rows = 0; L.vp = 0; L.ovf = 0; }
    if (tid < 2u * JL_POS_PER_WORD) L.cols[tid] = 0;
    __syncthreads();
    const uint32_t P = S.P;
    for (uint32_t base = 0; base < P; base += 1024u) {
        unsigned long long m[4];
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {   // four independent loads in flight per lane
            const uint32_t q = base + k * 256u + tid;
            m[k] = q < P ? S.called[q] : 0ull;
        }
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            if (m[k]) {
                const uint32_t c = S.pos_col[base + k * 256u + tid];
                atomicAdd(&L.n_rows, (uint32_t)__popcll(m[k]));
                if (c + 2u < n_cols) {
                    const uint32_t i = atomicAdd(&L.n_list, 1u);
                    if (i < kPlanList) L.list[i] = c;
                }
            }
        }
    }
    __syncthreads();
    const uint32_t n = min(L.n_list, kPlanList);
    if (tid < n) {   // first occurrence of its column? (overlapping genes in one frame call the same column twice)
        const uint32_t c = L.list[tid];
        bool f = true;
        for (uint32_t j = 0; j < tid; ++j) f = f && L.list[j] != c;
        L.first[tid] = f;
    }
    __syncthreads();
    if (tid < n && L.first[tid]) {
        const uint32_t c = L.list[tid];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; ++j) rank += (L.first[j] && L.list[j] < c) ? 1u : 0u;
        if (rank < 2u * JL_POS_PER_WORD) L.cols[rank] = c;
        atomicAdd(&L.vp, 1u);
    }
    if (tid == 0 && L.n_list > kPlanList) L.ovf = 1u;
    __syncthreads();
}

__device__ __forceinline__ uint32_t key_code(uint64_t key, uint32_t vp, uint32_t p) { return (uint32_t)(key >> (6u * (vp - 1u - p))) & 63u; }
// the same for a pattern of up to 20 positions in two words: positions 0..9 in w0, the rest in w1
__device__ __forceinline__ uint32_t key_code2(uint64_t w0, uint64_t w1, uint32_t vp, uint32_t p)
{
    const uint32_t n0 = vp < JL_POS_PER_WORD ? vp : JL_POS_PER_WORD;
    return p < n0 ? key_code(w0, n0, p) : key_code(w1, vp - n0, p - n0);
}

// Selection of the single-word pipeline out of LDS: every table word it needs is fetched ONCE (occupied list -> counts
// and keys: two dependent round trips), ranking, patterns, hit matrix and co-occurrence then run on LDS copies, and
// the result block is stored from there.  The general routine above chases each of them through memory again (a
// dozen dependent round trips on one CU: 10 us and more).  Handles what fits the result block (<= 128 candidates,
// <= 128 variant rows, hit matrix <= 4 KB); returns false — with nothing but idempotent side effects — otherwise.
// the half-key tables of a two-word launch, emptied of what the run put there (and their counters)
__device__ __forceinline__ void two_word_cleanup(const jl_two_word &tw)
{
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    const uint32_t na = ld_coherent(&tw.n_occ[0]), nb = ld_coherent(&tw.n_occ[1]);
    for (uint32_t q = tid; q < na; q += nt) tw.key_a[ld_coherent(&tw.occ_a[q])] = ~0ull;
    for (uint32_t q = tid; q < nb; q += nt) tw.key_b[ld_coherent(&tw.occ_b[q])] = ~0ull;
    __syncthreads();
    if (tid == 0) { tw.n_occ[0] = 0; tw.n_occ[1] = 0; }
}

constexpr uint32_t kSelCand = JL_PACK_MAX_HAP;
struct sel_lds {
    uint32_t slot[kSelCand], cnt[kSelCand];
    unsigned long long key[kSelCand], key1[kSelCand];     // key1 / hkey1: the second word of a two-word pattern
    uint32_t hcnt[kSelCand];
    unsigned long long hkey[kSelCand], hkey1[kSelCand];
    uint32_t vpos[JL_PACK_MAX_VAR];   // per variant row: how its position reads out of a pattern (see the selection)
    uint8_t hit[JL_SEL_HIT_BYTES];
    uint32_t hmask[JL_PACK_MAX_VAR][kSelCand / 32u];   // per variant: the haplotypes that carry it (co-occurrence sums)
    uint32_t ncand, insufficient, reported, bail;
};
static_assert(sizeof(sel_lds) <= kLdsSlots * 5u * 4u, "the selection's scratch must fit the grouping tables");
static_assert(offsetof(sel_lds, hmask) % 8 == 0, "a variant's haplotype set is read as two 64-bit words");

// KW = 2: the main table's key of a group is the pair (slot of its first word in table A, slot of its second word in
// table B); the words themselves are one more round trip away (jl_two_word).
// `ranked(bits)`: called by every thread once the slot -> haplotype table is complete and performed (the ids depend on nothing
// else): the caller releases the workgroups that wait for it and writes its own reads' ids THERE, so that the rest of the
// selection — hit matrix, co-occurrence, result block: 3.5 us at five positions, 11 at sixteen — runs beside the other
// workgroups' ids instead of in front of them.
template <int KW, typename Ranked>
__device__ __forceinline__ bool phase_select_lds(const jl_win_phase &w, const plan_state &L, uint32_t vp, uint32_t nv,
                                                 uint32_t n_rows, sel_lds &T, const jl_two_word *tw, uint32_t n_occ, uint32_t occ0,
                                                 const uint32_t *cat, uint32_t *id_bits_out, Ranked &&ranked)
{
    // n_occ, occ0: the group count and this thread's first entry of the group list, loaded by the caller beside the read
    // categories (occ0 is only meaningful below n_occ); cat: the read categories of the whole window (LDS)
    const select_args &S = w.S;
    jl_phase_meta *meta = w.meta;
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    if (tid == 0) { T.ncand = 0; T.insufficient = 0; T.reported = 0; T.bail = (nv > JL_PACK_MAX_VAR || n_rows > JL_PACK_MAX_VAR) ? 1u : 0u; }
    __syncthreads();
    if (T.bail) return false;
    // The groups, four per thread and trip: all of a thread's list entries are loaded before any is used, then all their
    // counts and keys (one dependent round trip per 1024 groups each way instead of one per 256: at 725 groups the scan was
    // a quarter of the selection).
    // Same-address LDS atomics serialise: one per non-candidate group (616 of 725 at sixteen positions) was most of the
    // scan.  The read counts of those groups add up in registers (one atomic per wave at the end); candidates are numbered
    // a wave at a time (ballot + one atomic per wave and trip).
    uint32_t insufficient = 0;
    const uint32_t lane = tid & 63u;
    uint32_t sl_keep[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};   // the first 1024 groups' slots: emptied from registers below
    // the two-word launch's half-key lists, for the same purpose: counts and each thread's first entries, requested now
    // ... and what the selection needs further down and does not depend on the groups: this thread's variant row and the
    // run counter (the parity of the result block) — requested here, beside the groups, instead of one round trip each later
    uint32_t row_c = 0, row_cw = 0;
    if (tid < nv) {
        const uint32_t *row = reinterpret_cast<const uint32_t *>(S.variants + tid);
        row_c = ld_coherent(row + 2);
        row_cw = ld_coherent(row + 3);
    }
    const uint32_t seq_pre = __hip_atomic_load(S.seq_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t tw_n[2] = {0, 0}, tw_first[2] = {0, 0};
    if (KW == 2) {
        tw_n[0] = ld_coherent(&tw->n_occ[0]);
        tw_n[1] = ld_coherent(&tw->n_occ[1]);
        tw_first[0] = ld_coherent(&tw->occ_a[tid]);   // (the lists hold at least 256 entries; used below the counts only)
        tw_first[1] = ld_coherent(&tw->occ_b[tid]);
    }
    for (uint32_t q0 = 0; q0 < n_occ; q0 += 4u * nt) {
        uint32_t sl[4], cn[4];
        unsigned long long ky[4];
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            const uint32_t q = q0 + j * nt + tid;
            sl[j] = q < n_occ ? ((q0 == 0u && j == 0u) ? occ0 : ld_coherent(&w.occupied[q])) : 0xFFFFFFFFu;
        }
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            cn[j] = 0;
            ky[j] = 0;
            if (sl[j] != 0xFFFFFFFFu) {
                cn[j] = ld_coherent(&w.slot_count[sl[j]]);
                ky[j] = ld_coherent64(&w.slot_key[sl[j]]);
            }
        }
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            const bool have = sl[j] != 0xFFFFFFFFu;
            const bool cand = have && cn[j] >= S.min_reads;
            const unsigned long long m = __ballot(cand);
            if (m) {   // wave-uniform
                uint32_t base = 0;
                const uint32_t leader = (uint32_t)__ffsll((long long)m) - 1u;
                if (lane == leader) base = atomicAdd(&T.ncand, (uint32_t)__popcll(m));
                base = (uint32_t)__shfl((int)base, (int)leader, 64);
                if (cand) {
                    const uint32_t i = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    if (i < kSelCand) { T.slot[i] = sl[j]; T.cnt[i] = cn[j]; T.key[i] = ky[j]; }
                    else T.bail = 1u;
                }
            }
            if (have && !cand) {
                insufficient += cn[j];
                __hip_atomic_store(&S.slot_hap[sl[j]], (uint32_t)JL_HAP_INSUFFICIENT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (q0 == 0u) {
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j) sl_keep[j] = sl[j];
        }
    }
    insufficient = wave_sum_all(insufficient);
    if (lane == 0 && insufficient) atomicAdd(&T.insufficient, insufficient);
    // the variant rows: column -> position index, codon (the table may come from another workgroup of this launch)
    for (uint32_t v = tid; v < nv; v += nt) {
        const uint32_t *row = reinterpret_cast<const uint32_t *>(S.variants + v);
        const uint32_t c = v == tid ? row_c : ld_coherent(row + 2), cw = v == tid ? row_cw : ld_coherent(row + 3);
        uint32_t pos = 0xFFu;
        if (c + 2u < S.n_cols)
            for (uint32_t p = 0; p < vp; ++p) pos = L.cols[p] == c ? p : pos;
        // how the hit loop reads this variant's position out of a pattern: bits 0-5 the shift, bit 6 the word, bits 8-15
        // the codon; bit 31: no such position
        const uint32_t n0 = vp < JL_POS_PER_WORD ? vp : JL_POS_PER_WORD;
        uint32_t e = 0x80000000u;
        if (pos != 0xFFu)
            e = (pos < n0 ? 6u * (n0 - 1u - pos) : (64u | (6u * (vp - n0 - 1u - (pos - n0))))) | (((cw >> 8) & 0xFFu) << 8);
        T.vpos[v] = e;
    }
    __syncthreads();
    JL_STAMP(13);
    const uint32_t H = T.ncand;
    if (T.bail || nv * H > JL_SEL_HIT_BYTES) return false;
    if (KW == 2) {   // the candidates' two words, from the half-key tables
        for (uint32_t a = tid; a < H; a += nt) {
            const unsigned long long pair = T.key[a];
            T.key[a] = ld_coherent64(&tw->key_a[(uint32_t)(pair >> 32)]);
            T.key1[a] = ld_coherent64(&tw->key_b[(uint32_t)pair]);
        }
        __syncthreads();
    }
    JL_STAMP(23);
    // Nothing below gives the run back to the general routine, and nothing reads the tables again: they are left empty for
    // the next run now (only the slots this run touched) — stores that go out beside the ranking instead of behind it.
    // (the first 1024 groups' slots are still in registers from the scan: stores only)
#pragma unroll
    for (uint32_t j = 0; j < 4u; ++j)
        if (sl_keep[j] != 0xFFFFFFFFu) {
            w.slot_key[sl_keep[j]] = ~0ull;
            w.slot_rep[sl_keep[j]] = 0xFFFFFFFFu;
            w.slot_count[sl_keep[j]] = 0;
        }
    for (uint32_t q = 4u * nt + tid; q < n_occ; q += nt) {
        const uint32_t sl = ld_coherent(&w.occupied[q]);
        w.slot_key[sl] = ~0ull;
        w.slot_rep[sl] = 0xFFFFFFFFu;
        w.slot_count[sl] = 0;
    }
    if (KW == 2) {
        if (tid < tw_n[0]) tw->key_a[tw_first[0]] = ~0ull;
        if (tid < tw_n[1]) tw->key_b[tw_first[1]] = ~0ull;
        for (uint32_t q = nt + tid; q < tw_n[0]; q += nt) tw->key_a[ld_coherent(&tw->occ_a[q])] = ~0ull;
        for (uint32_t q = nt + tid; q < tw_n[1]; q += nt) tw->key_b[ld_coherent(&tw->occ_b[q])] = ~0ull;
        if (tid == 0) { tw->n_occ[0] = 0; tw->n_occ[1] = 0; }   // (every thread has its copy of the counts)
    }
    // rank: (count desc, pattern asc); patterns are unique, so the ranks are a permutation.  Two threads per candidate
    // (at most 128 of them, 256 threads): each counts the competitors of one half that come before it.
    uint32_t *half_rank = reinterpret_cast<uint32_t *>(T.hit);   // [kSelCand] scratch: the hit matrix is filled later
    uint32_t my_rank = 0;
    {
        const uint32_t a = tid & 127u;
        if (a < H) {
            const uint32_t ca = T.cnt[a];
            const unsigned long long ka = T.key[a], ka1 = KW == 2 ? T.key1[a] : 0ull;
            const uint32_t b0 = tid < 128u ? 0u : H / 2u, b1 = tid < 128u ? H / 2u : H;
            uint32_t r = 0;
            // eight competitors' words are on their way from LDS before the first is compared (one at a time the loop
            // waited out an LDS round trip per competitor: 8 us at 109 candidates)
            for (uint32_t b = b0; b < b1; b += 8u) {
                uint32_t cb[8];
                unsigned long long kb[8], kb1[8];
#pragma unroll
                for (uint32_t j = 0; j < 8u; ++j) {
                    const uint32_t i = b + j < b1 ? b + j : b1 - 1u;
                    cb[j] = T.cnt[i];
                    kb[j] = T.key[i];
                    kb1[j] = KW == 2 ? T.key1[i] : 0ull;
                }
#pragma unroll
                for (uint32_t j = 0; j < 8u; ++j) {
                    const bool before = KW == 2 ? (kb[j] < ka || (kb[j] == ka && kb1[j] < ka1)) : kb[j] < ka;
                    r += (b + j < b1 && (cb[j] > ca || (cb[j] == ca && before))) ? 1u : 0u;
                }
            }
            if (tid < 128u) my_rank = r;
            else half_rank[a] = r;
        }
    }
    __syncthreads();
    JL_STAMP(24);
    {   // (H <= 128: the threads that hold my_rank)
        uint32_t ca = 0;
        if (tid < H) {
            ca = T.cnt[tid];
            const unsigned long long ka = T.key[tid], ka1 = KW == 2 ? T.key1[tid] : 0ull;
            const uint32_t rank = my_rank + half_rank[tid];
            T.hcnt[rank] = ca;
            T.hkey[rank] = ka;
            if (KW == 2) T.hkey1[rank] = ka1;
            __hip_atomic_store(&S.slot_hap[T.slot[tid]], rank, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // read by other workgroups
        }
        const uint32_t rs = wave_sum_all(ca);
        if ((tid & 63u) == 0 && rs) atomicAdd(&T.reported, rs);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the table's stores, the scan's among them, are performed)
    __syncthreads();
    JL_STAMP(25);
    ranked(id_bits_for(H));
    // hit[v][h] and, per variant, the set of haplotypes that carry it as bits: a lane per haplotype (its pattern words in
    // registers), a wave per variant (no division per element; the set makes the co-occurrence sums a walk over a few
    // common bits instead of H terms).  Beside it the patterns as bytes, [h][p], in the scratch the ranking is done with.
    uint8_t *pat = reinterpret_cast<uint8_t *>(T.slot);
    static_assert(offsetof(sel_lds, hcnt) >= kSelCand * 2u * JL_POS_PER_WORD + 8u, "the pattern bytes reuse slot/cnt/key/key1");
    for (uint32_t h0 = 0; h0 < H; h0 += 64u) {
        const uint32_t h = h0 + (tid & 63u);
        const unsigned long long k0 = h < H ? T.hkey[h] : 0ull, k1 = (KW == 2 && h < H) ? T.hkey1[h] : 0ull;
        for (uint32_t v = tid >> 6; v < nv; v += nt >> 6) {
            const uint32_t e = T.vpos[v];   // (wave-uniform)
            const unsigned long long word = (KW == 2 && (e & 64u)) ? k1 : k0;
            const bool x = h < H && !(e & 0x80000000u) && ((uint32_t)(word >> (e & 63u)) & 63u) == ((e >> 8) & 0xFFu);
            if (h < H) T.hit[v * H + h] = x ? 1 : 0;
            const unsigned long long m = __ballot(x);
            if ((tid & 63u) == 0) { T.hmask[v][h0 / 32u] = (uint32_t)m; T.hmask[v][h0 / 32u + 1u] = (uint32_t)(m >> 32); }
        }
    }
    for (uint32_t h = tid & 127u; h < H; h += 128u) {   // (slot / cnt / key / key1 are dead: the ranks are out)
        const unsigned long long k0 = T.hkey[h], k1 = KW == 2 ? T.hkey1[h] : 0ull;
        for (uint32_t p = tid >> 7; p < vp; p += 2u) pat[h * vp + p] = (uint8_t)(KW == 2 ? key_code2(k0, k1, vp, p) : key_code(k0, vp, p));
    }
    if (tid < 8u) {   // whole 8-byte words leave below: their tails are zeros
        pat[H * vp + tid] = 0;
        if (nv * H + tid < JL_SEL_HIT_BYTES) T.hit[nv * H + tid] = 0;
    }
    __syncthreads();
    const uint32_t bits = id_bits_for(H);
    *id_bits_out = bits;   // (every thread: a register of the caller)
    JL_STAMP(14);
    const uint32_t nvc = nv < S.cooc_cap ? nv : S.cooc_cap;
    const bool cooc_fits = nv <= JL_PACK_COOC_N;
    // ---- outputs.  The resident arrays are what the stage-API fetches read; a whole-path run whose results fit the result
    // block is read from that block alone (jl_phase_fetch, jl_run_view_get), so they are not written then (byte stores
    // with a division each: 6 us of the 36 the selection took at sixteen positions) ...
    const bool resident = S.called == nullptr || !cooc_fits;
    if (resident) {
        for (uint32_t h = tid; h < H; h += nt) S.hap_count[h] = T.hcnt[h];
        for (uint32_t h = tid & 127u; h < H; h += 128u)
            for (uint32_t p = tid >> 7; p < vp; p += 2u) S.hap_pattern[(uint64_t)h * JL_VARIANT_CAP + p] = pat[h * vp + p];
        for (uint32_t v = tid >> 6; v < nv; v += nt >> 6)
            for (uint32_t h = tid & 63u; h < H; h += 64u) S.hit[(uint64_t)v * JL_MAX_HAPLOTYPES + h] = T.hit[v * H + h];
    }
    JL_STAMP(26);
    jl_pack *pk = S.pk + (seq_pre & 1u);
    jl_pack *dsts[2] = {pk, S.mirror};
    // Co-occurrence C[v][x] = sum over the haplotypes that carry both of their read counts.  The counts are cut into bit
    // planes (plane b = the haplotypes whose count has bit b set, as a 128-bit set): C = sum_b 2^b popcount(set_v & set_x &
    // plane_b) — a fixed two dozen words per pair, no walk over the common haplotypes (7 us at 38 variants x 109
    // haplotypes).  The matrix is symmetric: a thread per pair v <= x of the upper triangle.
    unsigned long long *plane = T.hkey;   // [32][2]: the patterns are out as bytes, their words are dead
    static_assert(sizeof(T.hkey) >= 32u * 2u * 8u && kSelCand == 128u, "the bit planes reuse hkey; a set is two 64-bit words");
    const uint32_t n_planes = 32u - (uint32_t)__clz((int)(T.hcnt[0] | 1u));   // rank 0 holds the largest count
    if (tid < 128u) {
        const uint32_t c = tid < H ? T.hcnt[tid] : 0u;
        for (uint32_t b = 0; b < n_planes; ++b) {
            const unsigned long long m = __ballot((c >> b) & 1u);
            if ((tid & 63u) == 0) plane[b * 2u + (tid >> 6)] = m;
        }
    }
    __syncthreads();
    const uint32_t n_pairs = nvc * (nvc + 1u) / 2u;
    for (uint32_t q = tid; q < n_pairs; q += nt) {
        // q -> (v, x) of the upper triangle, rows v = 0 .. nvc-1 of lengths nvc, nvc-1, ...: the row from the closed form
        // (single precision is exact enough below 2^24 pairs), then at most one step either way
        const float nn = 2.0f * (float)nvc + 1.0f;
        uint32_t v = (uint32_t)((nn - __fsqrt_rn(nn * nn - 8.0f * (float)q)) * 0.5f);
        if (v >= nvc) v = nvc - 1u;
        while (v > 0u && v * nvc - v * (v - 1u) / 2u > q) --v;
        while ((v + 1u) * nvc - (v + 1u) * v / 2u <= q) ++v;
        const uint32_t x = v + (q - (v * nvc - v * (v - 1u) / 2u));
        const unsigned long long *mv = reinterpret_cast<const unsigned long long *>(T.hmask[v]);
        const unsigned long long *mx = reinterpret_cast<const unsigned long long *>(T.hmask[x]);
        const unsigned long long a0 = mv[0] & mx[0], a1 = mv[1] & mx[1];
        uint32_t sum = 0;
        if (a0 | a1)
            for (uint32_t b = 0; b < n_planes; ++b)
                sum += (uint32_t)(__popcll(a0 & plane[b * 2u]) + __popcll(a1 & plane[b * 2u + 1u])) << b;
        if (resident) {
            S.cooc[(uint64_t)v * S.cooc_cap + x] = sum;
            S.cooc[(uint64_t)x * S.cooc_cap + v] = sum;
        }
        if (cooc_fits)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                if (dsts[t]) { dsts[t]->cooc[v * nv + x] = sum; dsts[t]->cooc[x * nv + v] = sum; }
    }
    JL_STAMP(15);
    // ... and the result block, device copy (all-gather source) and pinned host mirror, straight from LDS, eight bytes a store
    if (tid == 0) {
        jl_phase_summary sm;
        sm.reported_reads = T.reported;
        sm.insufficient_reads = T.insufficient;
        sm.damaged_reads = cat[0];
        sm.marginal_gap = cat[1];
        sm.marginal_heteroduplex = cat[2];
        sm.marginal_partial = cat[3];
        sm.n_positions = vp;
        sm.n_haplotypes = H;
        meta->summary.reported_reads = sm.reported_reads;
        meta->summary.insufficient_reads = sm.insufficient_reads;
        meta->summary.n_haplotypes = H;
        meta->summary.n_positions = vp;
        __hip_atomic_store(&meta->id_bits, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            jl_pack *o = dsts[t];
            if (!o) continue;
            o->magic = JL_PACK_MAGIC; o->nvar_total = n_rows; o->fits_call = 1u; o->fits_phase = 1u;
            o->phase_ran = 1u; o->overflow = 0u; o->vp = vp; o->H = H;
            o->nv_phase = nv; o->cooc_fits = cooc_fits ? 1u : 0u; o->id_bits = bits;
            o->summary = sm;
        }
    }
    static_assert(offsetof(jl_pack, hap_pattern) % 8 == 0 && offsetof(jl_pack, hit) % 8 == 0 && offsetof(jl_pack, hap_count) % 8 == 0 &&
                  offsetof(jl_pack, variants) % 8 == 0 && offsetof(sel_lds, hit) % 8 == 0 && offsetof(sel_lds, hcnt) % 8 == 0,
                  "the result block's arrays leave as 8-byte words");
    {
        const unsigned long long *pat8 = reinterpret_cast<const unsigned long long *>(pat);
        const unsigned long long *hit8 = reinterpret_cast<const unsigned long long *>(T.hit);
        const unsigned long long *cnt8 = reinterpret_cast<const unsigned long long *>(T.hcnt);
        const uint32_t n_pat8 = (H * vp + 7u) / 8u, n_hit8 = (nv * H + 7u) / 8u < JL_SEL_HIT_BYTES / 8u ? (nv * H + 7u) / 8u : JL_SEL_HIT_BYTES / 8u;
        const uint32_t n_cnt8 = (H + 1u) / 2u;   // (hcnt holds kSelCand words: an odd H takes one stale word along, never read)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            jl_pack *o = dsts[t];
            if (!o) continue;
            for (uint32_t i = tid; i < nv * (uint32_t)(sizeof(jl_variant) / 8); i += nt)
                reinterpret_cast<unsigned long long *>(o->variants)[i] =
                    ld_coherent64(reinterpret_cast<const unsigned long long *>(S.variants) + i);
            for (uint32_t i = tid; i < vp; i += nt) o->pos_cols[i] = L.cols[i];
            for (uint32_t i = tid; i < n_cnt8; i += nt) reinterpret_cast<unsigned long long *>(o->hap_count)[i] = cnt8[i];
            for (uint32_t i = tid; i < n_pat8; i += nt) reinterpret_cast<unsigned long long *>(o->hap_pattern)[i] = pat8[i];
            for (uint32_t i = tid; i < n_hit8; i += nt) reinterpret_cast<unsigned long long *>(o->hit)[i] = hit8[i];
        }
    }
    JL_STAMP(16);
    JL_STAMP(17);
    if (tid == 0 && JL_STAMP_ON) g_stamps[18] = n_occ;
    return true;
}

// The exporting selection of the single-word pipeline (phasing sharded by reads: the groups go to the merge as they are),
// written for its latency: it is the tail of a launch the host waits for.  Everything it needs arrives in TWO dependent
// round trips — {group count, the first 256 entries of the occupied list, the run counter} and {count, key of each group} —
// where the general routine chases a dozen; the groups' rows leave 8 bytes at a time (they may lie in pinned host memory).
// `cat`: the read categories of the whole matrix, summed by the caller (LDS).
template <int KW>
__device__ __forceinline__ void phase_export_fast(const jl_win_phase &w, uint32_t vp, const uint32_t *cat, uint32_t *s_acc,
                                                  const jl_two_word *tw = nullptr)
{
    const select_args &S = w.S;
    jl_phase_meta *meta = w.meta;
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    if (tid == 0) *s_acc = 0;
    // trip 1 (the occupied list always holds at least one entry per thread: it is sized by the reads)
    const uint32_t n_occ = ld_coherent(&meta->n_occupied);
    uint32_t s_first = ld_coherent(&w.occupied[tid]);
    __syncthreads();
    uint32_t clean = 0;
    for (uint32_t q = tid; q < n_occ; q += nt) {
        const uint32_t s = q == tid ? s_first : ld_coherent(&w.occupied[q]);
        // trip 2: both words of the group before either is used
        const uint32_t c = ld_coherent(&w.slot_count[s]);
        unsigned long long key = ld_coherent64(&w.slot_key[s]), key_b = 0;
        if (KW == 2) {   // the pair of half-key slots -> the two words (one more round trip, all groups at once)
            const unsigned long long pair = key;
            key = ld_coherent64(&tw->key_a[(uint32_t)(pair >> 32)]);
            key_b = ld_coherent64(&tw->key_b[(uint32_t)pair]);
        }
        clean += c;
        // the slot remembers WHICH exported group it is: the merge answers per group
        __hip_atomic_store(&S.slot_hap[s], q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (q < S.exp_cap) {
            S.exp_count[q] = c;
            for (uint32_t p8 = 0; p8 < vp; p8 += 8u) {
                unsigned long long out = 0;
                for (uint32_t j = 0; j < 8u && p8 + j < vp; ++j)
                    out |= (unsigned long long)(KW == 2 ? key_code2(key, key_b, vp, p8 + j) : key_code(key, vp, p8 + j)) << (8u * j);
                *reinterpret_cast<unsigned long long *>(S.exp_pattern + (uint64_t)q * S.exp_stride + p8) = out;
            }
        }
        // leave the table empty for the next run: only the slots this run touched
        w.slot_key[s] = ~0ull;
        w.slot_rep[s] = 0xFFFFFFFFu;
        w.slot_count[s] = 0;
    }
    clean = wave_sum_all(clean);
    if ((tid & 63u) == 0 && clean) atomicAdd(s_acc, clean);
    __syncthreads();
    if (tid == 0) {
        const uint32_t total_clean = *s_acc;
        if (S.exp_head) {
            S.exp_head[0] = n_occ; S.exp_head[1] = vp; S.exp_head[2] = n_occ > S.exp_cap ? 1u : 0u;
            S.exp_head[3] = cat[0]; S.exp_head[4] = cat[1]; S.exp_head[5] = cat[2]; S.exp_head[6] = cat[3];
            S.exp_head[7] = total_clean;
        }
        // the run's scalars for the stage API (jl_phase_groups_fetch reads them from here)
        if (n_occ > S.exp_cap) atomicOr(&meta->overflow, 16u);
        meta->summary.reported_reads = 0;
        meta->summary.insufficient_reads = total_clean;   // clean reads: the merge decides which are reported
        meta->summary.damaged_reads = cat[0];
        meta->summary.marginal_gap = cat[1];
        meta->summary.marginal_heteroduplex = cat[2];
        meta->summary.marginal_partial = cat[3];
        meta->summary.n_haplotypes = 0;
        meta->id_bits = 16u;
        // a session's next launch may come without a plan kernel in front (jl_direct_cols): the group list starts empty
        if (S.exp_head) { meta->n_occupied = 0; meta->overflow = 0; }
    }
    if (KW == 2) {
        __syncthreads();
        two_word_cleanup(*tw);
    }
}

__device__ __forceinline__ uint32_t global_insert64(uint64_t key, uint32_t cnt, uint32_t first, uint64_t slots_mask,
                                                    unsigned long long *__restrict__ slot_key,
                                                    uint32_t *__restrict__ slot_rep, uint32_t *__restrict__ slot_count,
                                                    uint32_t *__restrict__ occupied, uint32_t *__restrict__ n_occupied)
{
    // (slot_rep / slot_count null: a table that only numbers its keys — the half-key tables of the two-word launch)
    uint64_t s = mix64(key + 0x9E3779B97F4A7C15ull) & slots_mask;
    for (;;) {
        // Look first: a slot's key never changes within a run, so a resident key read past the L1 is final and needs no
        // atomic.  Same-address atomics serialise at the memory side (about 12 ns each): with one CAS per workgroup on
        // the wild type's slot a million reads queued 500 of them in front of every workgroup's arrival.
        unsigned long long old = ld_coherent64(&slot_key[s]);
        if (old == kNoKey) old = atomicCAS(&slot_key[s], (unsigned long long)kNoKey, (unsigned long long)key);
        if (old == kNoKey) {
            // write-through stores: the selection may run in another workgroup of this launch
            if (slot_rep) __hip_atomic_store(&slot_rep[s], first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // any read carrying the key
            __hip_atomic_store(&occupied[atomicAdd(n_occupied, 1u)], (uint32_t)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            old = key;
        }
        if (old == key) {
            if (slot_count) atomicAdd(&slot_count[s], cnt);
            return (uint32_t)s;
        }
        s = (s + 1u) & slots_mask;
    }
}

template <int KW>
__device__ __forceinline__ void phase_fused1_body(const jl_win_phase &w, const jl_direct_cols *dc = nullptr, const jl_two_word *tw = nullptr)
{
    constexpr uint32_t NP = (uint32_t)KW * JL_POS_PER_WORD;   // positions this instantiation covers
    const uint8_t *__restrict__ msa = w.msa;
    const uint64_t col_stride = w.col_stride, n_reads = w.n_reads, reads_pad = w.reads_pad;
    jl_phase_meta *meta = w.meta;
    uint64_t *keys = w.keys;
    uint32_t *flagw = w.flagw;
    const uint64_t slots_mask = w.slots_mask;
    unsigned long long *slot_key = w.slot_key;
    uint32_t *slot_rep = w.slot_rep, *slot_count = w.slot_count, *occupied = w.occupied, *read_slot = w.read_slot;
    const select_args &S = w.S;
    // one LDS block per table set: the grouping tables (8 + 3 x 4 KB), lent to the selection once the grouping is done.  The
    // two-word launch has two sets: its first two rounds — the two words of the patterns, independent of each other — run
    // together, one set each.
    constexpr uint32_t NS = KW == 2 ? 2u : 1u;
    __shared__ unsigned long long s_tables[NS][kLdsSlots * 5u / 2u];
    auto s_key = [&](uint32_t z) -> unsigned long long * { return s_tables[z]; };                                          // [kLdsSlots]
    auto s_cnt = [&](uint32_t z) -> uint32_t * { return reinterpret_cast<uint32_t *>(s_tables[z] + kLdsSlots); };          // [kLdsSlots]
    auto s_first = [&](uint32_t z) -> uint32_t * { return s_cnt(z) + kLdsSlots; };                                         // [kLdsSlots]
    auto s_gslot = [&](uint32_t z) -> uint32_t * { return s_cnt(z) + 2u * kLdsSlots; };                                    // [kLdsSlots]
    __shared__ plan_state s_plan;
    __shared__ unsigned long long s_dom[NS];
    __shared__ uint32_t s_domcnt[NS], s_domfirst, s_domslot[NS], s_nlist[NS];
    __shared__ uint32_t s_last, s_cat[4], s_idbits, s_scan[4], s_running;
    __shared__ uint16_t s_list[NS][kLdsSlots];
    const uint32_t tid = threadIdx.x;
    const bool from_called = S.called != nullptr;
    // whole-path runs launch one workgroup more than the reads need: it compacts the called rows meanwhile
    const uint32_t n_arrive = w.n_blocks + (from_called ? 1u : 0u);
    const bool plan_block = from_called && blockIdx.x == w.n_blocks;
    JL_STAMP(0);

    // ---- 0. the plan
    uint32_t vp, n_rows;
    bool work;
    if (from_called) {
        plan_prologue(S, S.n_cols, s_plan);
        const uint32_t vpt = s_plan.vp;
        work = !s_plan.ovf && vpt >= 1u && vpt <= NP;
        vp = work ? vpt : 0u;
        n_rows = s_plan.n_rows;
        if (plan_block) {
            // the ordered table (SPEC §6) + what the fetch calls and a multi-word re-run read: written through, the
            // selection runs in whichever workgroup arrives last
            const uint32_t n = jl_compact_rows_block<false>(S.P, S.called, S.staged, S.rows, S.cap, s_scan, &s_running,
                                                            nullptr, 0u, true);
            if (tid < vp) {
                __hip_atomic_store(&S.vpcols_out[tid], s_plan.cols[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&S.col2pos_out[s_plan.cols[tid]], tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (tid == 0) {
                __hip_atomic_store(S.n_rows_out, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t need_kw = (vpt + JL_POS_PER_WORD - 1u) / JL_POS_PER_WORD;
                uint32_t ovf = 0;
                if (!work && (vpt != 0u || s_plan.ovf)) ovf = (s_plan.ovf || need_kw <= S.kwords_cap) ? 8u : 4u;
                __hip_atomic_store(&meta->n_var, n < S.cap ? n : S.cap, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&meta->vp_true, s_plan.ovf ? JL_POS_PER_WORD + 1u : vpt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&meta->vp, vp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&meta->kwords, work ? (uint32_t)KW : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&meta->overflow, ovf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&meta->summary.n_positions, vp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    } else if (dc && dc->on) {   // columns by pointer, vp by value: nothing to read, no kernel in front
        work = dc->vp != 0;
        vp = dc->vp;
        n_rows = 0;
    } else {
        const uint32_t mvp = meta->vp, kw = meta->kwords;  // one scalar round trip for both; the plan kernel wrote them
        work = (mvp != 0) & (kw >= 1) & (kw <= (uint32_t)KW);   // block-uniform
        vp = work ? mvp : 0u;
        // all the column indices in one go (the array always holds JL_VARIANT_CAP words; entries past vp are never used)
        if (tid < NP) s_plan.cols[tid] = w.vpcols[tid];
        if (tid == 0) s_plan.n_rows = 0;
        n_rows = 0;
        __syncthreads();
    }
    const uint64_t t = (uint64_t)blockIdx.x * 256u + tid;  // dword index within a column = 8 reads
    const bool live = t * 4u < col_stride;                 // never for the extra workgroup
    JL_STAMP(1);
    uint32_t clean_keep = 0;   // bit 4r: read r of this lane is clean
    uint32_t gslot[8];         // global table slot of each clean read
#pragma unroll
    for (int r = 0; r < 8; ++r) gslot[r] = 0;
    if (work && !plan_block) {
    auto clear_sets = [&](uint32_t n_sets) {
        for (uint32_t z = 0; z < n_sets; ++z) {
            for (uint32_t i = tid; i < kLdsSlots; i += 256u) { s_key(z)[i] = kNoKey; s_cnt(z)[i] = 0; s_first(z)[i] = 0xFFFFFFFFu; }
            if (tid == 0) { s_dom[z] = kNoKey; s_domcnt[z] = 0; s_domslot[z] = 0; s_nlist[z] = 0; }
        }
        if (tid == 0) s_domfirst = 0xFFFFFFFFu;
    };
    clear_sets(NS);
    if (tid < 4) s_cat[tid] = 0;
    __syncthreads();

    // ---- 1. keys and flags
    // Lanes past the end of the columns load from a clamped address and drop the value, so the loads need no
    // per-lane branch.
    uint32_t cols[NP];
#pragma unroll
    for (uint32_t p = 0; p < NP; ++p) cols[p] = s_plan.cols[p];
    // (a lane whose eight reads are all past the last read loads nothing of its own: a slice read in place — `direct` —
    // ends where the window's rows may end too)
    const uint64_t t_ld = (live && t * 8u < n_reads) ? t : 0u;
    uint32_t wd[NP][3];
    const bool direct = dc && dc->on;   // block-uniform
#pragma unroll
    for (uint32_t p = 0; p < NP; ++p) {
        if (p < vp) {  // block-uniform
            // nine plane rows per position (three columns x three planes), one byte = 8 reads of each
            const uint64_t ps = direct ? dc->stride : col_stride / 4u;
            const uint8_t *c0 = (direct && p < JL_POS_PER_WORD) ? dc->col[p] : msa + (uint64_t)cols[p] * 3u * ps;
#pragma unroll
            for (int k = 0; k < 3; ++k) wd[p][k] = jl_load_codes8(c0 + (uint64_t)k * 3u * ps + t_ld, ps);
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) wd[p][k] = 0x66666666u;
        }
    }
    if (!live) {
#pragma unroll
        for (uint32_t p = 0; p < NP; ++p)
#pragma unroll
            for (int k = 0; k < 3; ++k) wd[p][k] = 0x66666666u;
    }
    uint32_t gap = 0, het = 0, par = 0;
    uint64_t key[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t key1[KW == 2 ? 8 : 1] = {0};   // positions 10 .. 19 (two-word launch)
#pragma unroll
    for (uint32_t p = 0; p < NP; ++p) {
        if (p < vp) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint32_t b0 = wd[p][k] & kM1, b1 = (wd[p][k] >> 1) & kM1, b2 = (wd[p][k] >> 2) & kM1;
                gap |= b2 & ~b1 & ~b0;
                het |= b2 & b0;
                par |= b2 & b1;
            }
            const uint32_t hi2 = wd[p][0] & 0x33333333u;
            const uint32_t lo4 = ((wd[p][1] & 0x33333333u) << 2) | (wd[p][2] & 0x33333333u);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const uint64_t code = (((hi2 >> (4 * r)) & 3u) << 4) | ((lo4 >> (4 * r)) & 15u);
                if (KW == 1 || p < JL_POS_PER_WORD) key[r] = (key[r] << 6) | code;
                else key1[KW == 2 ? r : 0] = (key1[KW == 2 ? r : 0] << 6) | code;
            }
        }
    }
    uint32_t valid = 0;
    if (live) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
            if (t * 8u + r < n_reads) valid |= 1u << (4 * r);
    }
    JL_STAMP(2);
    gap &= valid; het &= valid; par &= valid;
    const uint32_t dirty = gap | het | par;
    const uint32_t cleanm = valid & ~dirty;  // bit 4r: read r is clean
    if (live && !S.fold) {
        // the flags and slots of every read are only needed by a later launch that writes the ids
        flagw[t] = gap | (het << 1) | (par << 2) | ((valid ^ kM1) << 3);
    }
    if (KW == 1 && live && !S.run) {   // the multi-word pipeline's own group launch reads the keys
        uint64_t *dst = keys + t * 8u;
#pragma unroll
        for (int r = 0; r < 8; r += 2) {
            ulonglong2 v;
            v.x = key[r];
            v.y = key[r + 1];
            *reinterpret_cast<ulonglong2 *>(dst + r) = v;
        }
    }
    {
        // read categories: wave sums into LDS here, ONE global atomic per block and counter further down (the four
        // counters share a cache line: 1564 waves adding to it serialise at the memory side)
        const uint32_t n_gap = wave_sum_all(__popc(gap)), n_het = wave_sum_all(__popc(het));
        const uint32_t n_par = wave_sum_all(__popc(par)), n_dam = wave_sum_all(__popc(dirty));
        if ((tid & 63u) == 0) {
            if (n_dam) atomicAdd(&s_cat[0], n_dam);
            if (n_gap) atomicAdd(&s_cat[1], n_gap);
            if (n_het) atomicAdd(&s_cat[2], n_het);
            if (n_par) atomicAdd(&s_cat[3], n_par);
        }
    }
    // One ROUND of grouping: the clean reads' 64-bit keys kk[z] -> the slot of each in the global table T[z] (gs[z][r]), for N
    // independent key words at once (they share the round's barriers and their trips to the global tables overlap: a round is
    // 6.5 us whatever it carries).  The one-word launch runs one round of one, on the patterns themselves; the two-word
    // launch a round of two — the words of the patterns — and a round of one on the pairs of their slots (jl_two_word).
    struct table_t { unsigned long long *key; uint32_t *rep, *cnt, *occ, *nocc; };
    auto group_round = [&](auto n_const, const uint64_t (&kk)[decltype(n_const)::value][8], const table_t (&T)[decltype(n_const)::value],
                           uint32_t (&gs)[decltype(n_const)::value][8], bool first_round) {
        constexpr uint32_t N = decltype(n_const)::value;
        if (!first_round) {   // (the first round's tables were cleared while the columns were on their way)
            clear_sets(N);
            __syncthreads();
        }
        // ---- 2. dominant key of the block = key of its first clean read
        uint32_t myfirst = 0xFFFFFFFFu;
        if (cleanm) myfirst = (uint32_t)(t * 8u) + ((uint32_t)__ffs((int)cleanm) - 1u) / 4u;
        {
            uint32_t m = myfirst;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, o, 64));
            if ((tid & 63u) == 0 && m != 0xFFFFFFFFu) atomicMin(&s_domfirst, m);
        }
        __syncthreads();
        const uint32_t domfirst = s_domfirst;
        if (domfirst != 0xFFFFFFFFu && (uint64_t)(domfirst >> 3) == t) {
#pragma unroll
            for (int r = 0; r < 8; ++r)  // static indices keep the keys in registers
                if ((domfirst & 7u) == (uint32_t)r) {
#pragma unroll
                    for (uint32_t z = 0; z < N; ++z) s_dom[z] = kk[z][r];
                }
        }
        __syncthreads();
        unsigned long long dom[N];
        uint32_t isdom[N];  // bit 4r
#pragma unroll
        for (uint32_t z = 0; z < N; ++z) {
            dom[z] = s_dom[z];
            isdom[z] = 0;
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if (((cleanm >> (4 * r)) & 1u) && kk[z][r] == dom[z]) isdom[z] |= 1u << (4 * r);
            const uint32_t c = wave_sum_all(__popc(isdom[z]));
            if ((tid & 63u) == 0 && c) atomicAdd(&s_domcnt[z], c);
        }
        JL_STAMP(3);
        // ---- 3. everything else through the LDS table
        uint32_t myslot[N][8];
#pragma unroll
        for (uint32_t z = 0; z < N; ++z) {
            const uint32_t rest = cleanm & ~isdom[z];
            unsigned long long *tk = s_key(z);
            uint32_t *tc = s_cnt(z), *tf = s_first(z);
#pragma unroll
            for (int r = 0; r < 8; ++r) myslot[z][r] = 0xFFFFFFFFu;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if ((rest >> (4 * r)) & 1u) {
                    const unsigned long long k = kk[z][r];
                    uint32_t sl = (uint32_t)mix64(k) & (kLdsSlots - 1u);
                    for (uint32_t probe = 0; probe < kLdsSlots; ++probe) {
                        const unsigned long long old = atomicCAS(&tk[sl], (unsigned long long)kNoKey, k);
                        if (old == kNoKey || old == k) {
                            atomicAdd(&tc[sl], 1u);
                            atomicMin(&tf[sl], (uint32_t)(t * 8u + r));
                            myslot[z][r] = sl;
                            break;
                        }
                        sl = (sl + 1u) & (kLdsSlots - 1u);
                    }
                }
            }
        }
        __syncthreads();
        JL_STAMP(4);
        // One global insert per distinct key of the block, ALL AT ONCE: the keys are listed densely first, so thread i takes
        // the i-th (a sweep over the 1024 table slots, four per thread, made the thread that owned two occupied slots — and
        // thread 0, which also had the dominant key — do its inserts one after the other: 8.7 us of a 25 us launch at a
        // million reads, two to three dependent round trips each).  The lists of a round of two lie one behind the other.
        // The dominant keys go with the last threads, which have list entries of their own only in blocks with some 250
        // distinct keys or more.
        {
            // (the lists are made by a sweep over the tables: four LDS reads per thread and set, and a barrier)
#pragma unroll
            for (uint32_t z = 0; z < N; ++z)
                for (uint32_t sl = tid; sl < kLdsSlots; sl += 256u)
                    if (s_cnt(z)[sl]) s_list[z][atomicAdd(&s_nlist[z], 1u)] = (uint16_t)sl;
            __syncthreads();
            const uint32_t n0 = s_nlist[0], n_all = n0 + (N == 2u ? s_nlist[N - 1u] : 0u);
            for (uint32_t i = tid; i < n_all; i += 256u) {
                const uint32_t z = (N == 2u && i >= n0) ? 1u : 0u;
                const uint32_t sl = s_list[z][i - (z ? n0 : 0u)];
                s_gslot(z)[sl] = global_insert64(s_key(z)[sl], s_cnt(z)[sl], s_first(z)[sl], slots_mask, T[z].key, T[z].rep, T[z].cnt, T[z].occ, T[z].nocc);
            }
#pragma unroll
            for (uint32_t z = 0; z < N; ++z)
                if (tid == 255u - z && s_domcnt[z])
                    s_domslot[z] = global_insert64(dom[z], s_domcnt[z], s_domfirst, slots_mask, T[z].key, T[z].rep, T[z].cnt, T[z].occ, T[z].nocc);
        }
        __syncthreads();
#pragma unroll
        for (uint32_t z = 0; z < N; ++z)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if ((cleanm >> (4 * r)) & 1u) {
                    uint32_t g;
                    if ((isdom[z] >> (4 * r)) & 1u) g = s_domslot[z];
                    else if (myslot[z][r] != 0xFFFFFFFFu) g = s_gslot(z)[myslot[z][r]];
                    else  // LDS table full (> 1024 distinct keys in 2048 reads): straight to the global table
                        g = global_insert64(kk[z][r], 1u, (uint32_t)(t * 8u + r), slots_mask, T[z].key, T[z].rep, T[z].cnt, T[z].occ, T[z].nocc);
                    gs[z][r] = g;
                }
            }
        if (KW == 2) __syncthreads();   // the next round clears the LDS tables this one still reads
    };
    const table_t T_main = {slot_key, slot_rep, slot_count, occupied, &meta->n_occupied};
    if constexpr (KW == 1) {
        const uint64_t (&k1)[1][8] = reinterpret_cast<const uint64_t (&)[1][8]>(key);
        uint32_t (&g1)[1][8] = reinterpret_cast<uint32_t (&)[1][8]>(gslot);
        const table_t T1[1] = {T_main};
        group_round(std::integral_constant<uint32_t, 1u>{}, k1, T1, g1, true);
    } else {
        uint64_t kk2[2][8];
#pragma unroll
        for (int r = 0; r < 8; ++r) { kk2[0][r] = key[r]; kk2[1][r] = key1[r]; }
        uint32_t g01[2][8] = {{0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}};
        const table_t T2[2] = {{tw->key_a, nullptr, nullptr, tw->occ_a, tw->n_occ}, {tw->key_b, nullptr, nullptr, tw->occ_b, tw->n_occ + 1}};
        group_round(std::integral_constant<uint32_t, 2u>{}, kk2, T2, g01, true);
        uint64_t kk3[1][8];
#pragma unroll
        for (int r = 0; r < 8; ++r) kk3[0][r] = ((uint64_t)g01[0][r] << 32) | g01[1][r];   // never all ones: slots are 31-bit numbers
        uint32_t (&g1)[1][8] = reinterpret_cast<uint32_t (&)[1][8]>(gslot);
        const table_t T1[1] = {T_main};
        group_round(std::integral_constant<uint32_t, 1u>{}, kk3, T1, g1, false);
    }
    // read categories of this workgroup's reads: written through to its own four words; the selection adds the workgroups
    // up (four atomics per workgroup on ONE cache line were the longest queue of the launch at a million reads)
    if (tid >= 64u && tid < 68u)
        __hip_atomic_store(&w.blockcat[blockIdx.x * 4u + (tid - 64u)], s_cat[tid - 64u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!S.fold) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
            if ((cleanm >> (4 * r)) & 1u) read_slot[t * 8u + r] = gslot[r];
    }
    JL_STAMP(5);
    clean_keep = cleanm;
    }  // work
    if (!S.run) return;  // the generic pipeline has its own select launch
    bool ids_written = false;
    auto write_ids = [&](uint32_t bits) {
        if (live && bits) {
            uint16_t h[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                h[r] = JL_HAP_DAMAGED;
                if ((clean_keep >> (4 * r)) & 1u)
                    h[r] = (uint16_t)__hip_atomic_load(&S.slot_hap[gslot[r]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            store_ids(S.read_hap, t, h, bits);
        }
    };
    // ---- hand-off: the block that arrives last ranks the groups and writes the result block.  Everything the
    // selection reads was written by agent-scope atomics or write-through stores (slot keys and counts, the occupied
    // list, the read-category counters, the compacted rows): no release fence — an L2 write-back per block serialises
    // in the L2.  Every wave drains its stores -> block barrier -> one lane: the arrival add; the last arriver reads
    // with agent-scope loads (past its L1).  The counter is zero before the first launch and the last arriver leaves it zero.
    JL_STAMP(6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const uint32_t prev = __hip_atomic_fetch_add(S.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t last = prev == n_arrive - 1u;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(S.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_last = last;
    }
    __syncthreads();
    JL_STAMP(7);
    const bool last = s_last != 0;
    if (!last && !S.fold) return;
    if (last) {
        JL_STAMP(8);
        uint32_t nv = 0, seq_before = 0;
        const bool fast_export = work && S.exp_count != nullptr;
        // the selection's first loads go out beside those of the read categories (the list entry is used below the count only)
        uint32_t n_occ_pre = 0, occ_pre = 0;
        if (work && !S.exp_count) {
            n_occ_pre = ld_coherent(&meta->n_occupied);
            occ_pre = ld_coherent(&occupied[tid]);   // (the list holds at least 256 entries: reserve_phase)
        }
        if (!S.exp_count) {   // (an exporting selection needs neither: it ranks nothing)
            nv = from_called ? (n_rows < S.cap ? n_rows : S.cap) : ld_coherent(&meta->n_var);
            if (!from_called) n_rows = ld_coherent(&S.n_rows[0]);
        } else if (tid == 0 && S.seq_host) {
            seq_before = ld_coherent(S.seq_dev);   // beside the loads below: not a round trip of its own at the end
        }
        if (work) {   // the read categories: every workgroup's four words, summed here
            uint32_t c4[4] = {0, 0, 0, 0};
            for (uint32_t b = tid; b < w.n_blocks; b += 256u) {
#pragma unroll
                for (int k = 0; k < 4; ++k) c4[k] += ld_coherent(&w.blockcat[b * 4u + k]);   // written through by their workgroups
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) c4[k] = wave_sum_all(c4[k]);
            if (tid < 4u) s_cat[tid] = 0;
            __syncthreads();
            if ((tid & 63u) == 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) atomicAdd(&s_cat[k], c4[k]);
            }
            __syncthreads();
            if (!fast_export && tid == 0) {   // the run's scalars (the general selection and the stage-API fetches read them there)
                __hip_atomic_store(&meta->summary.damaged_reads, s_cat[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&meta->summary.marginal_gap, s_cat[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&meta->summary.marginal_heteroduplex, s_cat[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&meta->summary.marginal_partial, s_cat[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        JL_STAMP(9);
        if (fast_export) {
            phase_export_fast<KW>(w, vp, s_cat, &s_running, tw);
            JL_STAMP(10);
            // every wave's stores (the groups may lie in host memory) are performed; ONE system-scope release, carried by
            // the completion word's store, pushes them out in front of it: data and word leave the same workgroup
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            JL_STAMP(11);
            if (S.seq_host && tid == 0) jl_signal_done_from(seq_before, S.seq_dev, S.seq_host);
            JL_STAMP(12);
            return;
        }
        bool done = false;
        uint32_t sel_bits = 0;   // width of the ids, when the selection out of LDS ran
        if (work && !S.exp_count)
            done = phase_select_lds<KW>(w, s_plan, vp, nv, n_rows, *reinterpret_cast<sel_lds *>(s_tables[0]), tw, n_occ_pre, occ_pre, s_cat,
                                        &sel_bits, [&](uint32_t bits) {
                                            if (!S.fold) return;
                                            if (tid == 0) __hip_atomic_store(S.flag, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                            JL_STAMP(19);
                                            write_ids(bits);
                                            ids_written = true;
                                        });
        if (KW == 2 && !done) {
            // More candidates / rows than the selection out of LDS holds (or nothing to phase): the general routine reads
            // one-word keys, so the run is handed to the multi-word pipeline — tables emptied, the run flagged as one that
            // needs it (the fetch calls re-run it, as they do for more than 20 positions).
            __syncthreads();
            if (work) {
                const uint32_t n_occ = ld_coherent(&meta->n_occupied);
                for (uint32_t q = tid; q < n_occ; q += 256u) {
                    const uint32_t sl = ld_coherent(&occupied[q]);
                    slot_key[sl] = ~0ull;
                    slot_rep[sl] = 0xFFFFFFFFu;
                    slot_count[sl] = 0;
                }
                two_word_cleanup(*tw);
                if (tid == 0) {
                    __hip_atomic_store(&meta->overflow, 8u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&meta->vp_true, 2u * JL_POS_PER_WORD + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&meta->vp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&meta->kwords, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&meta->n_occupied, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the general routine below reads these scalars back
        }
        if (!done) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the read categories: it reads them from the run's scalars)
            __syncthreads();
            phase_select_block<true>(S.min_reads, reads_pad, keys, meta, slot_rep, slot_count, occupied, S.slot_hap, S.variants,
                                     S.col2pos, S.n_cols, S.hap_count, S.hap_pattern, S.hit, S.n_rows, w.vpcols, S.cooc, S.cooc_cap,
                                     S.pk, S.mirror, slot_key, S.seq_dev, reinterpret_cast<uint32_t *>(s_tables[0]), S.exp_count,
                                     S.exp_pattern, S.exp_cap, S.exp_stride, nullptr, nullptr, 0u, S.exp_head);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (S.xhead) {   // bound exchange: the head into the region the launch's all-gather works in
            jl_result_head_copy(S.pk + (__hip_atomic_load(S.seq_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u), S.xhead);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        // The result block went to pinned host memory from THIS compute die; the completion word will be stored by
        // whichever workgroup arrives last, possibly on another die.  A system-scope release here pushes the block
        // out of this die's L2 first — without it the host now and then saw the word before the block's header
        // (new per-read ids and variant rows beside the previous run's read categories: one group run in ten).
        if (tid == 0) __threadfence_system();
        if (!S.fold) {
            if (S.seq_host && tid == 0) signal_done(S.seq_dev, S.seq_host);  // no per-read ids wanted: the run ends here
            return;
        }
        // the slot -> haplotype table is complete (write-through stores, drained above): release the waiting
        // workgroups; the flag carries the width of the ids
        if (tid == 0) {
            const uint32_t bits = done ? sel_bits : ld_coherent(&meta->id_bits);
            s_idbits = bits;
            if (!ids_written) {     // (the selection out of LDS released them as soon as it had ranked)
                __hip_atomic_store(S.flag, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                JL_STAMP(19);
            }
        }
        __syncthreads();
    } else {
        // wait for the selection.  Every workgroup of this launch is resident (the host folds only small grids), so
        // the flag does arrive; the bound turns a broken invariant into a loud failure, not a hang — and not a fault
        // either (a trap can take the device down for every tenant): the run is marked failed where both ways of reading
        // it look (the phase scalars for jl_phase_fetch, the pinned block's magic for jl_run_view_get), this workgroup
        // writes no ids and goes on to the second arrival so that the launch still ends.
        if (tid == 0) {
            uint32_t spins = 0, bits;
            while ((bits = __hip_atomic_load(S.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > (1u << 23)) {
                    atomicOr(&meta->overflow, 32u);
                    if (S.mirror) __hip_atomic_store(&S.mirror->magic, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
            }
#ifdef JL_TUNING
            if (S.fold == 2u && bits != 0u) {   // forced: behave as if the wait had run out just before the selection arrived
                atomicOr(&meta->overflow, 32u);
                if (S.mirror) __hip_atomic_store(&S.mirror->magic, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                bits = 0u;
            }
#endif
            s_idbits = bits;   // 0: timed out
        }
        JL_STAMP(20);
        __syncthreads();
    }
    // ---- per-read haplotype ids of this workgroup's own reads, straight from the slots still in registers
    if (!ids_written) write_ids(s_idbits);
    // Second arrival: every workgroup is past the flag by now, so the one that arrives last resets it (and the
    // counter) for the next launch and, when this launch ends a run, stores the completion word behind all the ids.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    JL_STAMP(21);
    if (tid == 0) {
        if (S.seq_host) __threadfence_system();   // this workgroup's ids leave its die's L2 before it arrives (see above)
        const uint32_t prev = __hip_atomic_fetch_add(S.arrive2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == n_arrive - 1u) {
            __hip_atomic_store(S.arrive2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(S.flag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // A workgroup that gave up waiting marked the run failed; a selection that was only LATE has since written a
            // valid result block over that mark.  The mark in the run's scalars is the lasting one: whoever arrives last
            // makes the blocks say so again, just before the completion word.
            if (ld_coherent(&meta->overflow) & 32u) {
                if (S.mirror) __hip_atomic_store(&S.mirror->magic, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                jl_pack *pk = S.pk + (__hip_atomic_load(S.seq_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u);
                __hip_atomic_store(&pk->magic, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (S.seq_host) signal_done(S.seq_dev, S.seq_host);
            JL_STAMP(22);
        }
    }
}

__global__ __launch_bounds__(256) void phase_fused1_kernel(jl_win_phase w) { phase_fused1_body<1>(w); }
__global__ __launch_bounds__(256) void phase_fused1_direct_kernel(jl_win_phase w, jl_direct_cols dc) { phase_fused1_body<1>(w, &dc); }
// 11 .. 20 variant positions: the same launch with two key words (jl_two_word)
__global__ __launch_bounds__(256) void phase_fused2_kernel(jl_win_phase w, jl_two_word tw) { phase_fused1_body<2>(w, nullptr, &tw); }

// one launch for several windows: blockIdx.z = window
__global__ __launch_bounds__(256) void phase_group_run_kernel(jl_phase_group_args args)
{
    const jl_win_phase &w = args.w[blockIdx.z];
    if (blockIdx.x >= w.n_blocks + (w.S.called ? 1u : 0u)) return;
    phase_fused1_body<1>(w);
}

// ---------------------------------------------------------------------------------------- assign
// Per-read ids by a launch of their own (launches too large for their workgroups to wait for each other, and the
// multi-word pipeline).  Eight reads per lane: one flag word, two 16-byte loads of slots, one store of packed ids.
__device__ __forceinline__ void assign_body(uint64_t n_dwords, const uint32_t *__restrict__ flagw,
                                            const jl_phase_meta *__restrict__ meta, const uint32_t *__restrict__ read_slot,
                                            const uint32_t *__restrict__ slot_hap, uint16_t *__restrict__ read_hap)
{
    const bool phased = meta->vp != 0;
    const uint32_t bits = meta->id_bits;
    // t = dword index within a column; the grid may be smaller than the window (see jl_launch_assign_group)
    for (uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x; t < n_dwords; t += (uint64_t)gridDim.x * 256u) {
        uint16_t h[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) h[r] = JL_HAP_DAMAGED;
        if (phased) {
            const uint32_t f = flagw[t];
            const uint4 s0 = *reinterpret_cast<const uint4 *>(read_slot + t * 8u);
            const uint4 s1 = *reinterpret_cast<const uint4 *>(read_slot + t * 8u + 4u);
            const uint32_t slot[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if (((f >> (4 * r)) & 15u) == 0) h[r] = (uint16_t)slot_hap[slot[r]];   // clean reads only: their slot is valid
        }
        store_ids(read_hap, t, h, bits);
    }
}

__global__ __launch_bounds__(256) void phase_assign_kernel(uint64_t n_dwords, const uint32_t *__restrict__ flagw,
                                                            const jl_phase_meta *__restrict__ meta,
                                                            const uint32_t *__restrict__ read_slot,
                                                            const uint32_t *__restrict__ slot_hap,
                                                            uint16_t *__restrict__ read_hap)
{
    assign_body(n_dwords, flagw, meta, read_slot, slot_hap, read_hap);
}

// the same for the windows of a group launch (blockIdx.z = window)
__global__ __launch_bounds__(256) void phase_assign_group_kernel(jl_phase_group_args args)
{
    const jl_win_phase &w = args.w[blockIdx.z];
    assign_body(w.col_stride / 4u, w.flagw, w.meta, w.read_slot, w.S.slot_hap, w.S.read_hap);
}

}  // namespace

#ifdef JL_EXP_STAMPS
extern "C" __attribute__((visibility("default"))) int jl_debug_stamps(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(g_stamps)) == hipSuccess ? 0 : -2;
}
#endif

#ifndef JL_ASSIGN_HOST_BLOCKS
#define JL_ASSIGN_HOST_BLOCKS 128u
#endif
// workgroups per window of a launch that writes per-read ids (see jl_launch_assign_group)
static uint32_t jl_assign_blocks(uint32_t n_win, uint32_t max_read_blocks, bool to_host)
{
    if (!to_host) return max_read_blocks;   // ids that stay in HBM: one workgroup per 2048 reads
    // to pinned host memory: JL_ASSIGN_HOST_BLOCKS workgroups in all (see jl_launch_assign_group)
    const uint32_t cap = std::max(1u, JL_ASSIGN_HOST_BLOCKS / std::max(1u, n_win));
    return std::min<uint32_t>(cap, max_read_blocks);
}

// Argument block of the fused phase launch for one window.  `from_called`: the plan comes out of the call masks of
// the Fisher stage (whole-path runs); otherwise meta / vpcols hold it already.  `fold_budget`: workgroups that may
// still be added to a launch whose members all wait for each other (the per-read ids are then written by the same
// launch); returns whether this window folds.
bool jl_fill_win_phase(jl_ctx *ctx, uint32_t min_reads, bool signal, uint32_t fold_budget, bool from_called, jl_win_phase *w)
{
    memset(w, 0, sizeof *w);
    const bool generic = ctx->phase_generic;
    const bool ids_to_host = ctx->read_hap_out != nullptr;
    const bool signal_select = signal && !ids_to_host;   // no ids wanted on the host: the selection ends the run
    const uint32_t n_dwords = (uint32_t)(ctx->col_stride / 4u);
    const uint32_t fblocks = (n_dwords + 255u) / 256u;
    w->msa = ctx->d_msa; w->col_stride = ctx->col_stride; w->n_reads = ctx->n_reads; w->reads_pad = ctx->col_stride * 2u;
    w->vpcols = ctx->d_vpcols; w->meta = ctx->d_meta; w->keys = ctx->d_keys; w->flagw = ctx->d_flagw;
    w->slots_mask = ctx->table_slots - 1u; w->slot_key = (unsigned long long *)ctx->d_slot_key;
    w->slot_rep = ctx->d_slot_rep; w->slot_count = ctx->d_slot_count; w->occupied = ctx->d_occupied;
    w->read_slot = ctx->d_read_slot;
    w->blockcat = ctx->d_blockcat;
    w->n_blocks = fblocks;
    select_args &S = w->S;
    S.run = generic ? 0u : 1u;
    S.min_reads = min_reads; S.n_cols = ctx->n_cols; S.cooc_cap = ctx->cooc_cap;
    S.slot_hap = ctx->d_slot_hap; S.variants = ctx->d_variants; S.col2pos = ctx->d_col2pos;
    S.hap_count = ctx->d_hap_count; S.hap_pattern = ctx->d_hap_pattern; S.hit = ctx->d_hit; S.n_rows = ctx->d_nvar;
    S.cooc = ctx->d_cooc; S.pk = ctx->d_pack; S.mirror = ctx->pack_mirror;
    S.arrive = ctx->d_sync + 2; S.seq_dev = ctx->d_sync;
    // a launch of at most JL_FOLD_MAX_BLOCKS workgroups in all also writes the per-read ids: one launch less
    if (ctx->phase_export) {
        if (ctx->exp_ext_count) {   // a session's block (pinned host memory, or the send buffer of the all-gather)
            S.exp_count = ctx->exp_ext_count; S.exp_pattern = ctx->exp_ext_pattern; S.exp_head = ctx->exp_ext_head;
            S.exp_cap = ctx->exp_ext_cap; S.exp_stride = ctx->exp_ext_stride;
        } else {
            S.exp_count = ctx->d_exp_count; S.exp_pattern = ctx->d_exp_pattern;
            S.exp_cap = ctx->exp_cap; S.exp_stride = ctx->exp_stride;
        }
    }
    // (an exporting run keeps every read's flags and slot: jl_phase_regroup maps them once the merge is known)
    const bool fold = !generic && !ctx->phase_export && !ctx->no_fold && fblocks + 1u <= fold_budget;
    S.fold = fold ? 1u : 0u;
#ifdef JL_TUNING
    // tests of the re-run: the waiting workgroups of a folded launch give up at once (as if they had not been resident together)
    if (fold && getenv("JL_FORCE_FOLD_TIMEOUT")) S.fold = 2u;
#endif
    S.flag = ctx->d_sync + 4; S.arrive2 = ctx->d_sync + 3;
    S.read_hap = ctx->read_hap_out ? ctx->read_hap_out : ctx->d_read_hap;
    S.seq_host = (fold ? signal : signal_select) ? ctx->h_seq : nullptr;
    if (from_called && !generic) {
        S.called = ctx->d_called; S.staged = ctx->d_staged; S.pos_col = ctx->d_pos_col; S.rows = ctx->d_variants;
        S.n_rows_out = ctx->d_nvar; S.vpcols_out = ctx->d_vpcols; S.col2pos_out = ctx->d_col2pos;
        S.P = ctx->P; S.cap = JL_VARIANT_CAP; S.kwords_cap = ctx->keys_words;
    }
    return fold;
}

// The phasing launches behind a plan.  `from_called`: whole-path run, the Fisher stage left call masks (ignored by the
// multi-word pipeline, whose plan a compact launch made); otherwise the stand-alone plan kernel runs here when
// `planned` is false.  ctx->phase_generic selects the multi-word pipeline; the default runs only the single-word
// (Vp <= 10) kernel, whose last block also does the selection, and flags inputs that need more (jl_phase_fetch then
// switches and re-runs).  `signal`: this launch ends a jl_run_async — its last kernel stores the completion word.
// Returns whether the launches store the run's completion word themselves (otherwise the caller adds done_kernel).
bool jl_launch_phase(jl_ctx *ctx, hipStream_t st, uint32_t min_reads, bool planned, bool from_called, bool signal)
{
    const uint64_t reads_pad = ctx->col_stride * 2u;
    const bool generic = ctx->phase_generic;
    const bool ids_to_host = ctx->read_hap_out != nullptr;
    const bool signal_select = signal && !ids_to_host;
    const bool two = ctx->phase_two && !generic;
    if (!planned && !(from_called && !generic))
        hipLaunchKernelGGL(phase_plan_kernel, dim3(1), dim3(1024), 0, st, ctx->d_variants, ctx->d_nvar, JL_VARIANT_CAP,
                           ctx->n_cols, ctx->d_varcol, ctx->d_vpcols, ctx->d_col2pos, ctx->keys_words,
                           generic ? 0u : (two ? 2u : 1u), ctx->d_meta);
    const uint32_t n_dwords = (uint32_t)(ctx->col_stride / 4u);
    const uint32_t rblocks = (uint32_t)((ctx->n_reads + 255u) / 256u);
    if (generic)
        hipLaunchKernelGGL(phase_keys_kernel, dim3((n_dwords + 255u) / 256u), dim3(256), 0, st, ctx->d_msa,
                           ctx->plane_stride, ctx->n_reads, reads_pad, ctx->d_vpcols, ctx->d_meta, ctx->d_keys,
                           ctx->d_flagw);
    jl_win_phase w;
    const bool fold = jl_fill_win_phase(ctx, min_reads, signal, JL_FOLD_MAX_BLOCKS, from_called, &w);
    // (the multi-word pipeline has its own keys / grouping / selection launches: the fused launch would find nothing to do)
    if (!generic && ctx->direct.on && !w.S.called)
        hipLaunchKernelGGL(phase_fused1_direct_kernel, dim3(w.n_blocks), dim3(256), 0, st, w, ctx->direct);
    else if (two) {
        jl_two_word tw;
        tw.key_a = (unsigned long long *)ctx->d_slot_key_a; tw.key_b = (unsigned long long *)ctx->d_slot_key_b;
        tw.occ_a = ctx->d_occ_a; tw.occ_b = ctx->d_occ_b; tw.n_occ = ctx->d_sync + 10;
        hipLaunchKernelGGL(phase_fused2_kernel, dim3(w.n_blocks + (w.S.called ? 1u : 0u)), dim3(256), 0, st, w, tw);
    } else if (!generic) hipLaunchKernelGGL(phase_fused1_kernel, dim3(w.n_blocks + (w.S.called ? 1u : 0u)), dim3(256), 0, st, w);
    if (generic) {
        hipLaunchKernelGGL(phase_group_kernel, dim3(rblocks), dim3(256), 0, st, ctx->n_reads, reads_pad, ctx->d_keys,
                           ctx->d_flagw, ctx->d_meta, ctx->table_slots - 1u, ctx->d_slot_rep, ctx->d_slot_count,
                           ctx->d_occupied, ctx->d_read_slot);
        hipLaunchKernelGGL(phase_select_kernel, dim3(1), dim3(1024), 0, st, min_reads, reads_pad, ctx->d_keys,
                           ctx->d_meta, ctx->d_slot_rep, ctx->d_slot_count, ctx->d_occupied, ctx->d_slot_hap,
                           ctx->d_variants, ctx->d_col2pos, ctx->n_cols, ctx->d_hap_count, ctx->d_hap_pattern, ctx->d_hit,
                           ctx->d_nvar, ctx->d_vpcols, ctx->d_cooc, ctx->cooc_cap, ctx->d_pack, ctx->pack_mirror,
                           (unsigned long long *)ctx->d_slot_key, ctx->d_sync, signal_select ? ctx->h_seq : nullptr,
                           w.S.exp_count, w.S.exp_pattern, w.S.exp_cap, w.S.exp_stride, w.S.exp_head);
    }
    if (fold) return signal;
    if (ctx->phase_export) return false;   // the ids wait for the merge (jl_phase_regroup)
    hipLaunchKernelGGL(phase_assign_kernel, dim3(jl_assign_blocks(1, w.n_blocks, ids_to_host)), dim3(256), 0, st,
                       (uint64_t)n_dwords, ctx->d_flagw, ctx->d_meta, ctx->d_read_slot, ctx->d_slot_hap,
                       ctx->read_hap_out ? ctx->read_hap_out : ctx->d_read_hap);
    return false;
}

static void fill_phase_group_args(jl_phase_group_args *args, const jl_win_phase *h_wins, uint32_t n_win)
{
    memset(args, 0, sizeof *args);
    memcpy(args->w, h_wins, sizeof(jl_win_phase) * (n_win < JL_GROUP_MAX ? n_win : JL_GROUP_MAX));
}

void jl_launch_phase_group(const jl_win_phase *h_wins, uint32_t n_win, uint32_t max_blocks, hipStream_t st)
{
    jl_phase_group_args args;
    fill_phase_group_args(&args, h_wins, n_win);
    hipLaunchKernelGGL(phase_group_run_kernel, dim3(max_blocks + 1u, 1, n_win), dim3(256), 0, st, args);
}

void jl_launch_assign_group(const jl_win_phase *h_wins, uint32_t n_win, uint32_t max_read_blocks, bool to_host, hipStream_t st)
{
    jl_phase_group_args args;
    fill_phase_group_args(&args, h_wins, n_win);
    // The ids usually go to pinned host memory, i.e. over PCIe at a fiftieth of the HBM rate.  A launch that has all
    // of them in flight at once (49 workgroups per 100k reads) fills the L2's write queues with host-bound lines and the
    // pileups of the other launches on the chip wait behind them; about 32 workgroups looping over the reads keep PCIe
    // saturated all the same.
    const uint32_t bx = jl_assign_blocks(n_win, max_read_blocks, to_host);
    hipLaunchKernelGGL(phase_assign_group_kernel, dim3(bx, 1, n_win), dim3(256), 0, st, args);
}


// ---------------------------------------------------------------------------------------- sharded by reads: the merge's answer
// hap_of_group[q] = haplotype id (or JL_HAP_INSUFFICIENT) of the q-th group this matrix exported; every table slot holds
// its group's index since the exporting selection (kernels_xwin.hip maps the reads)
void jl_launch_regroup(jl_ctx *ctx, const uint16_t *d_hap_of_group, uint32_t n_groups, uint32_t n_haplotypes, bool phased)
{
    jl_xw_assign_args a;
    memset(&a, 0, sizeof a);
    a.n_dwords = ctx->col_stride / 4u;
    a.flagw = ctx->d_flagw; a.read_slot = ctx->d_read_slot; a.slot_hap = ctx->d_slot_hap; a.read_hap = ctx->d_read_hap;
    a.n_groups = n_groups;
    a.bits = n_haplotypes <= JL_ID4_MAX_H ? 4u : (n_haplotypes <= JL_ID8_MAX_H ? 8u : 16u);
    a.phased = phased ? 1u : 0u;   // no variant position: every read is "damaged", flags and slots were never written
    jl_launch_xw_assign(&a, nullptr, d_hap_of_group, ctx->stream);
}
