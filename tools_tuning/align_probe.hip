// How much slower are 16-byte loads at byte addresses that are not multiples of 16 (the ingest's quality loads)?  A wave's lanes
// read 32 consecutive bytes each as two dwordx4 loads (lane stride 32 B), from an offset of `mis` bytes; 300 MB per launch.
// hipcc --offload-arch=gfx950 -O3 -o align_probe align_probe.hip && ./align_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4a1 __attribute__((ext_vector_type(4), aligned(1)));
template <int MODE>   // 0: two x4 loads at p + mis; 1: three aligned x4 loads around it; 2: eight dword loads at p + mis (4-aligned when mis % 4 == 0)
__global__ __launch_bounds__(256) void probe(const uint8_t *__restrict__ src, size_t n32, int mis, uint32_t *__restrict__ out)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n32; i += (size_t)gridDim.x * 256) {
        const uint8_t *p = src + 32 * i + mis;
        if (MODE == 0) {
            const u32x4 a = *reinterpret_cast<const u32x4a1 *>(p), b = *reinterpret_cast<const u32x4a1 *>(p + 16);
            acc += a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w;
        } else if (MODE == 1) {
            const u32x4 *q = reinterpret_cast<const u32x4 *>(src + 32 * i);
            const u32x4 a = q[0], b = q[1], c = q[2];
            acc += a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.w;
        } else {
            typedef uint32_t u32a1 __attribute__((aligned(1)));
            const u32a1 *q = reinterpret_cast<const u32a1 *>(p);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += q[k];
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main()
{
    const size_t bytes = (size_t)300 << 20, n32 = bytes / 32 - 4;
    uint8_t *d[4];
    uint32_t *out;
    for (auto &p : d) { hipMalloc(&p, bytes + 256); hipMemset(p, 1, bytes + 256); }
    hipMalloc(&out, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode)
        for (int mis : {0, 16, 8, 4, 2, 1, 7}) {
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(256 * 16), dim3(256), 0, 0, d[rep & 3], n32, mis, out);
                else if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(256 * 16), dim3(256), 0, 0, d[rep & 3], n32, mis, out);
                else hipLaunchKernelGGL(probe<2>, dim3(256 * 16), dim3(256), 0, 0, d[rep & 3], n32, mis, out);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep && ms < best) best = ms;
            }
            printf("mode %d (%s) misalignment %2d: %7.1f us  %6.0f GB/s\n", mode, mode == 0 ? "2 x dwordx4 at the byte address" : mode == 1 ? "3 aligned dwordx4" : "8 dword loads", mis, 1e3 * best, bytes / best / 1e6);
        }
    return 0;
}
