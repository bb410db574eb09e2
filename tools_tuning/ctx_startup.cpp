// ctx_startup.cpp — where the time goes between process start and a usable juliet context (the CLI's "context ready"):
//   hipcc -O2 -Iinclude tools_tuning/ctx_startup.cpp -o /tmp/ctx_startup -Lminorseq_amd -ljuliet_hip -Wl,-rpath,$PWD/minorseq_amd
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

#include "juliet_hip.h"

static std::chrono::steady_clock::time_point t0;
static double lap(const char *what)
{
    static double last = 0;
    const double now = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    printf("%-44s %8.2f ms  (at %8.2f)\n", what, now - last, now);
    last = now;
    return now;
}

__global__ void nop_kernel(int *p) { if (p) *p = 1; }

int main()
{
    t0 = std::chrono::steady_clock::now();
    int n = 0;
    hipGetDeviceCount(&n);
    lap("hipGetDeviceCount (runtime init)");
    hipSetDevice(0);
    lap("hipSetDevice");
    hipFree(nullptr);
    lap("hipFree(0)");
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    lap("hipStreamCreate");
    void *d = nullptr, *h = nullptr;
    hipMalloc(&d, 1 << 20);
    lap("first hipMalloc 1 MB");
    hipMalloc(&d, 1 << 20);
    lap("second hipMalloc 1 MB");
    hipHostMalloc(&h, 64, hipHostMallocDefault);
    lap("first hipHostMalloc 64 B");
    hipHostMalloc(&h, 1 << 20, hipHostMallocDefault);
    lap("hipHostMalloc 1 MB");
    hipHostMalloc(&h, 64 << 20, hipHostMallocDefault);
    lap("hipHostMalloc 64 MB");
    nop_kernel<<<1, 64, 0, s>>>(nullptr);
    hipStreamSynchronize(s);
    lap("first launch of this program's kernel + sync");
    hipStream_t s2;
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    lap("second hipStreamCreate");
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    lap("two hipEventCreate");
    for (int i = 0; i < 14; ++i) hipMalloc(&d, 4096 << (i % 5));
    lap("14 small hipMalloc");
    hipMalloc(&d, 300 << 20);
    lap("hipMalloc 300 MB");
    hipHostMalloc(&h, 64 << 10, hipHostMallocDefault);
    lap("hipHostMalloc 64 KB");
    hipMemsetAsync(d, 0, 1 << 20, s2);
    hipStreamSynchronize(s2);
    lap("memset 1 MB + sync on the new stream");
    jl_ctx *c = nullptr;
    jl_ctx_create(0, nullptr, &c);
    lap("jl_ctx_create #1");
    jl_ctx *c2 = nullptr;
    jl_ctx_create(0, nullptr, &c2);
    lap("jl_ctx_create #2");
    return 0;
}
