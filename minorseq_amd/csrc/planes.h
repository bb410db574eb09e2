// planes.h — bit-level helpers of the resident format (jl_internal.h: three bit planes per column).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define JL_PL_FN __host__ __device__ __forceinline__
#else
#define JL_PL_FN static inline
#endif

// bit k of the eight 4-bit codes in w -> eight bits (code 0 in bit 0)
JL_PL_FN uint32_t jl_plane_bits8(uint32_t w, uint32_t k)
{
    uint32_t x = (w >> k) & 0x11111111u;     // bit k of the eight codes, one per nibble
    x = (x | (x >> 3)) & 0x03030303u;        // two per byte
    x = (x | (x >> 6)) & 0x000F000Fu;        // four per half
    return (x | (x >> 12)) & 0xFFu;          // eight: code 0 in bit 0
}

// eight bits -> bit 0 of eight nibbles (bit 0 in nibble 0): the inverse
JL_PL_FN uint32_t jl_spread8(uint32_t b)
{
    uint32_t x = (b | (b << 12)) & 0x000F000Fu;
    x = (x | (x << 6)) & 0x03030303u;
    return (x | (x << 3)) & 0x11111111u;
}

// the three plane bytes of 8 reads of one column -> their codes as eight nibbles
JL_PL_FN uint32_t jl_planes_to_nibbles8(uint32_t b0, uint32_t b1, uint32_t b2)
{
    return jl_spread8(b0) | (jl_spread8(b1) << 1) | (jl_spread8(b2) << 2);
}

#if defined(__HIPCC__)
// eight reads of one column as a dword of codes: byte `p` of plane 0 and the same byte of planes 1 and 2
__device__ __forceinline__ uint32_t jl_load_codes8(const uint8_t *p, uint64_t plane_stride)
{
    const uint32_t b0 = p[0], b1 = p[plane_stride], b2 = p[2u * plane_stride];
    return jl_planes_to_nibbles8(b0, b1, b2);
}
#endif
