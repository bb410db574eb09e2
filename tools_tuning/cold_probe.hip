// cold_probe.hip — what does a dependent chain of small loads cost right after a 150 MB stream, as a function of how
// the small buffers were allocated?  (a) 32 separate hipMallocs of 4 KB, (b) 32 regions 64 KB apart in ONE allocation.
// Chain kernel: one thread walks p = buf[k][p] 32 times (each load depends on the last); timed by the device clock.
// Tuning aid.  Build: hipcc -O3 --offload-arch=gfx950 cold_probe.hip -o cold_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void stream_kernel(const u32x4 *__restrict__ p, uint64_t n16, uint32_t *sink)
{
    u32x4 acc = {0, 0, 0, 0};
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256u)
        acc ^= __builtin_nontemporal_load(p + i);
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9E3779B9u) sink[threadIdx.x] = acc.x;
}

struct ptrs { uint32_t *b[32]; };
__global__ void chain_kernel(ptrs P, uint64_t *out)
{
    const uint64_t t0 = wall_clock64();
    uint32_t p = 0;
    for (int k = 0; k < 32; ++k) p = __hip_atomic_load(P.b[k] + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t t1 = wall_clock64();
    out[0] = t1 - t0 + p;
}

int main()
{
    const uint64_t bytes = 150ull << 20;
    uint8_t *big; uint32_t *sink; uint64_t *d_out;
    CHECK(hipMalloc(&big, bytes)); CHECK(hipMemset(big, 1, bytes));
    CHECK(hipMalloc(&sink, 4096)); CHECK(hipMalloc(&d_out, 64));
    ptrs A, B;
    for (int k = 0; k < 32; ++k) { CHECK(hipMalloc(&A.b[k], 4096)); CHECK(hipMemset(A.b[k], 0, 4096)); }
    uint8_t *arena; CHECK(hipMalloc(&arena, 4u << 20)); CHECK(hipMemset(arena, 0, 4u << 20));
    for (int k = 0; k < 32; ++k) B.b[k] = (uint32_t *)(arena + (size_t)k * 65536);
    hipStream_t st; CHECK(hipStreamCreate(&st));
    for (int mode = 0; mode < 4; ++mode) {   // 0/1: separate, cold/warm; 2/3: arena, cold/warm
        const ptrs &P = mode < 2 ? A : B;
        const bool cold = (mode & 1) == 0;
        double sum = 0;
        for (int r = 0; r < 20; ++r) {
            if (cold) hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, st, (const u32x4 *)big, bytes / 16, sink);
            hipLaunchKernelGGL(chain_kernel, dim3(1), dim3(1), 0, st, P, d_out);
            uint64_t h; CHECK(hipMemcpyAsync(&h, d_out, 8, hipMemcpyDeviceToHost, st)); CHECK(hipStreamSynchronize(st));
            if (r >= 4) sum += (double)h / 100.0;
        }
        printf("%s buffers, %s: 32 dependent loads take %.2f us (%.0f ns each)\n", mode < 2 ? "32 separate 4 KB" : "one 4 MB arena  ",
               cold ? "after a 150 MB stream" : "back to back         ", sum / 16, sum / 16 * 1000 / 32);
    }
    return 0;
}
