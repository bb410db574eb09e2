// jl_synth.h — deterministic synthetic aligned-CCS reads, one pure function per MSA cell, shared by the
// device fill kernel (kernels_synth.hip), the C++ front end's BAM writer and mirrored in numpy
// (minorseq_amd/synth.py) for the tests.  Mixture semantics follow the reference's mixdata tool
// (doc/MIXDATA.md:9-22: one major clone + minors at a given percentage) and the fixture name on
// doc/img/juliet_input.png ("..._3000_96_1": 96 % major, 1 % minors); rates are benchmark parameters
// (SURVEY.md §8d), not claims about juliet.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define JL_HD __host__ __device__ inline
#else
#define JL_HD inline
#endif

#define JL_SYNTH_N_EDITS 5

typedef struct {
    uint64_t seed;
    uint32_t n_cols;
    uint32_t t_mask, t_del, t_sub;  // cumulative 24-bit thresholds
    uint32_t t_partial;             // 24-bit threshold
    uint32_t cum_permille[4];       // cumulative minor haplotype frequencies in 1/1000
    uint32_t edit_col[JL_SYNTH_N_EDITS];   // column of each edit
    uint8_t edit_base[JL_SYNTH_N_EDITS];   // base code written there
    uint8_t edit_hap[JL_SYNTH_N_EDITS];    // haplotype (1..4) carrying the edit
} jl_synth_plan;

JL_HD uint64_t jl_splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// per-read draw: haplotype (0 = major) and covered column range [start, end)
JL_HD void jl_synth_read(const jl_synth_plan *pl, uint64_t read, uint32_t *hap, uint32_t *start, uint32_t *end)
{
    uint64_t uh = jl_splitmix64(pl->seed * 0x100000001B3ull + 2 * read);
    uint32_t v = (uint32_t)(uh % 1000u);
    uint32_t h = 0;
    if (v < pl->cum_permille[0]) h = 1;
    else if (v < pl->cum_permille[1]) h = 2;
    else if (v < pl->cum_permille[2]) h = 3;
    else if (v < pl->cum_permille[3]) h = 4;
    *hap = h;
    uint64_t up = jl_splitmix64(pl->seed * 0x100000001B3ull + 2 * read + 1);
    uint32_t s = 0, e = pl->n_cols;
    if ((uint32_t)(up >> 40) < pl->t_partial) {
        uint32_t q = pl->n_cols / 4 + 1;
        s = (uint32_t)(up & 0xFFFFu) % q;
        e = pl->n_cols - (uint32_t)((up >> 16) & 0xFFFFu) % q;
    }
    *start = s;
    *end = e;
}

// symbol code (0..6) of cell (read, col) given the read's draw and the reference base there
JL_HD uint32_t jl_synth_cell(const jl_synth_plan *pl, uint64_t read, uint32_t col, uint32_t hap, uint32_t start,
                             uint32_t end, uint32_t ref_base)
{
    if (col < start || col >= end) return 6u;
    uint32_t b = ref_base;
    for (int k = 0; k < JL_SYNTH_N_EDITS; ++k)
        if (pl->edit_col[k] == col && pl->edit_hap[k] == hap) b = pl->edit_base[k];
    uint64_t u = jl_splitmix64(pl->seed + 0x632BE59BD9B4E019ull * (read * (uint64_t)pl->n_cols + col + 1));
    uint32_t v = (uint32_t)(u >> 40);
    if (v < pl->t_mask) return 5u;
    if (v < pl->t_del) return 4u;
    if (v < pl->t_sub) return (b + 1u + (uint32_t)((u >> 8) % 3u)) & 3u;
    return b;
}

// host-side plan construction (same in C++ and numpy): thresholds, edit sites
#include <math.h>
static inline void jl_synth_make_plan(jl_synth_plan *pl, uint64_t seed, uint32_t n_cols, double sub_rate,
                                      double del_rate, double mask_rate, double partial_rate,
                                      const uint32_t minor_permille[4], const uint8_t *ref)
{
    static const uint32_t aa[JL_SYNTH_N_EDITS] = {41, 65, 181, 190, 215};  // M41L K65R Y181C G190A T215Y
    static const uint8_t hap_of[JL_SYNTH_N_EDITS] = {1, 2, 3, 3, 4};       // A.3: Y181C+G190A co-occur
    pl->seed = seed;
    pl->n_cols = n_cols;
    const double two24 = 16777216.0;
    pl->t_mask = (uint32_t)floor(mask_rate * two24);
    pl->t_del = pl->t_mask + (uint32_t)floor(del_rate * two24);
    pl->t_sub = pl->t_del + (uint32_t)floor(sub_rate * two24);
    pl->t_partial = (uint32_t)floor(partial_rate * two24);
    uint32_t c = 0;
    for (int k = 0; k < 4; ++k) { c += minor_permille[k]; pl->cum_permille[k] = c; }
    uint32_t P = n_cols / 3;
    for (int k = 0; k < JL_SYNTH_N_EDITS; ++k) {
        uint32_t kk = (uint32_t)(((uint64_t)aa[k] * P) / 1000u);
        uint32_t col = 3 * kk + (uint32_t)(k % 3);
        if (col >= n_cols) col = n_cols - 1;
        pl->edit_col[k] = col;
        pl->edit_base[k] = (uint8_t)((ref[col] + 1 + (k & 1)) & 3);
        pl->edit_hap[k] = hap_of[k];
    }
}

// random ACGT reference without a stop codon in frame 0
static inline void jl_synth_reference(uint64_t seed, uint32_t n_cols, uint8_t *ref)
{
    for (uint32_t c = 0; c < n_cols; ++c) ref[c] = (uint8_t)(jl_splitmix64((seed ^ 0x6A756C696574ull) + c) & 3u);
    for (uint32_t c = 0; c + 2 < n_cols; c += 3) {
        uint32_t cod = 16u * ref[c] + 4u * ref[c + 1] + ref[c + 2];
        if (cod == 48u || cod == 50u || cod == 56u) ref[c] = 1;  // TAA TAG TGA -> CAA CAG CGA
    }
}
