// stream_probe.hip — how fast can gfx950 READ a 150 MB column-packed window, as a function of launch shape?
// Tuning aid only (not product, not test).  Build: hipcc -O3 --offload-arch=gfx950 stream_probe.hip -o stream_probe
// Each variant xor-reduces the bytes it loads; the result is stored only if it matches an impossible value, so the
// loads cannot be removed.  Timing: hipEvents around ONE launch (what rocprof's duration sees, plus launch latency)
// and around R back-to-back launches (throughput form).  Buffers: `nbuf` different windows in rotation (nbuf*150 MB
// beyond the 256 MiB Infinity Cache) or the same window every time (nbuf = 1: cache-resident).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

// flat grid-stride stream: U independent 16-B loads per lane per iteration
template <int U, bool NT>
__global__ __launch_bounds__(256) void flat_kernel(const u32x4 *__restrict__ p, uint64_t n16, uint32_t *sink)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256u * U;
    u32x4 acc = {0, 0, 0, 0};
    for (uint64_t i = (uint64_t)blockIdx.x * 256u * U + threadIdx.x; i < n16; i += stride) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t k = i + (uint64_t)u * 256u;
            if (k < n16) v[u] = NT ? __builtin_nontemporal_load(p + k) : p[k];
            else v[u] = (u32x4){0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9E3779B9u) sink[threadIdx.x] = acc.x;
}

// contiguous slab per block (block b reads [b*slab, (b+1)*slab) ), U loads in flight
template <int U, bool NT>
__global__ __launch_bounds__(256) void slab_kernel(const u32x4 *__restrict__ p, uint64_t n16, uint32_t *sink)
{
    const uint64_t slab = (n16 + gridDim.x - 1) / gridDim.x;
    const uint64_t b0 = (uint64_t)blockIdx.x * slab;
    const uint64_t b1 = b0 + slab < n16 ? b0 + slab : n16;
    u32x4 acc = {0, 0, 0, 0};
    for (uint64_t i = b0 + threadIdx.x; i < b1; i += 256u * U) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t k = i + (uint64_t)u * 256u;
            if (k < b1) v[u] = NT ? __builtin_nontemporal_load(p + k) : p[k];
            else v[u] = (u32x4){0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9E3779B9u) sink[threadIdx.x] = acc.x;
}

// the pileup's shape: block = W columns x strided 4 KiB tiles, W loads in flight per lane
template <int W, bool NT>
__global__ __launch_bounds__(256) void col_kernel(const uint8_t *__restrict__ msa, uint64_t col_stride, uint32_t n_cols,
                                                  uint32_t n_tiles, uint32_t *sink)
{
    const uint32_t c0 = blockIdx.x * W;
    u32x4 acc = {0, 0, 0, 0};
    for (uint32_t t = blockIdx.y; t < n_tiles; t += gridDim.y) {
        const uint64_t off = (uint64_t)t * 4096u + threadIdx.x * 16u;
        if (off >= col_stride) continue;
        u32x4 v[W];
#pragma unroll
        for (int j = 0; j < W; ++j) {
            if (c0 + j < n_cols) {
                const u32x4 *src = reinterpret_cast<const u32x4 *>(msa + (uint64_t)(c0 + j) * col_stride + off);
                v[j] = NT ? __builtin_nontemporal_load(src) : *src;
            } else v[j] = (u32x4){0, 0, 0, 0};
        }
#pragma unroll
        for (int j = 0; j < W; ++j) acc ^= v[j];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9E3779B9u) sink[threadIdx.x] = acc.x;
}

struct launcher { const char *name; void (*fn)(const uint8_t *, uint32_t *, hipStream_t); };

static uint64_t g_bytes, g_stride;
static uint32_t g_cols, g_lds;   // g_lds: dynamic LDS per block of the col kernels (caps the blocks per CU: 160 KB / g_lds)

template <int U, bool NT, int BLOCKS> static void l_flat(const uint8_t *p, uint32_t *s, hipStream_t st)
{ hipLaunchKernelGGL((flat_kernel<U, NT>), dim3(BLOCKS), dim3(256), 0, st, (const u32x4 *)p, g_bytes / 16, s); }
template <int U, bool NT, int BLOCKS> static void l_slab(const uint8_t *p, uint32_t *s, hipStream_t st)
{ hipLaunchKernelGGL((slab_kernel<U, NT>), dim3(BLOCKS), dim3(256), 0, st, (const u32x4 *)p, g_bytes / 16, s); }
template <int W, bool NT, int RS> static void l_col(const uint8_t *p, uint32_t *s, hipStream_t st)
{
    const uint32_t n_tiles = (uint32_t)((g_stride + 4095) / 4096);
    hipLaunchKernelGGL((col_kernel<W, NT>), dim3((g_cols + W - 1) / W, RS), dim3(256), g_lds, st, p, g_stride, g_cols, n_tiles, s);
}

int main(int argc, char **argv)
{
    const uint64_t n_reads = argc > 1 ? strtoull(argv[1], 0, 10) : 100000;
    g_cols = argc > 2 ? atoi(argv[2]) : 3000;
    g_lds = argc > 3 ? (uint32_t)atoi(argv[3]) : 0;
    g_stride = ((n_reads + 1) / 2 + 127) / 128 * 128;
    g_bytes = g_stride * g_cols;
    const int NB = 4, R = 20;
    uint8_t *buf[NB];
    uint32_t *sink;
    for (int i = 0; i < NB; ++i) { CHECK(hipMalloc(&buf[i], g_bytes)); CHECK(hipMemset(buf[i], 0x11 * (i + 1), g_bytes)); }
    CHECK(hipMalloc(&sink, 4096));
    hipStream_t st; CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));

    const launcher ls[] = {
#define FL(U, NT, B) {"flat U" #U " nt" #NT " blocks" #B, l_flat<U, NT, B>}
#define SL(U, NT, B) {"slab U" #U " nt" #NT " blocks" #B, l_slab<U, NT, B>}
#define CL(W, NT, RS) {"col  W" #W " nt" #NT " rsplit" #RS, l_col<W, NT, RS>}
        FL(4, true, 1024), FL(4, true, 2048), FL(4, true, 4096), FL(4, true, 8192),
        FL(2, true, 2048), FL(2, true, 4096), FL(8, true, 1024), FL(8, true, 2048),
        FL(4, false, 2048), FL(4, false, 4096), FL(1, true, 4096), FL(1, true, 8192), FL(1, true, 16384),
        SL(4, true, 1024), SL(4, true, 2048), SL(4, true, 4096), SL(8, true, 1024), SL(8, true, 512),
        CL(3, true, 1), CL(3, true, 2), CL(3, true, 4), CL(6, true, 1), CL(6, true, 2), CL(6, true, 4), CL(12, true, 4), CL(12, true, 8),
        CL(3, false, 1), CL(6, false, 2),
    };
    printf("window %llu reads x %u cols = %.1f MB (stride %llu)\n", (unsigned long long)n_reads, g_cols, g_bytes / 1e6, (unsigned long long)g_stride);
    printf("%-34s %10s %10s %10s %10s\n", "variant", "1x rot us", "b2b rot us", "1x same us", "b2b same us");
    for (const launcher &l : ls) {
        if (g_lds && strncmp(l.name, "col", 3) != 0) continue;
        float res[4];
        for (int mode = 0; mode < 2; ++mode) {       // 0: rotate 4 buffers, 1: same buffer
            for (int w = 0; w < 8; ++w) l.fn(buf[mode ? 0 : w % NB], sink, st);
            CHECK(hipStreamSynchronize(st));
            std::vector<float> one;
            for (int r = 0; r < R; ++r) {
                CHECK(hipEventRecord(e0, st));
                l.fn(buf[mode ? 0 : r % NB], sink, st);
                CHECK(hipEventRecord(e1, st));
                CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                one.push_back(ms);
            }
            std::sort(one.begin(), one.end());
            res[mode * 2] = one[R / 2] * 1000.f;
            CHECK(hipEventRecord(e0, st));
            for (int r = 0; r < R; ++r) l.fn(buf[mode ? 0 : r % NB], sink, st);
            CHECK(hipEventRecord(e1, st));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            res[mode * 2 + 1] = ms * 1000.f / R;
        }
        printf("%-34s %10.1f %10.1f %10.1f %10.1f   (%.2f TB/s b2b rot)\n", l.name, res[0], res[1], res[2], res[3], g_bytes / (res[1] * 1e-6) / 1e12);
        fflush(stdout);
    }
    return 0;
}
