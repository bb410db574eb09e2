// Pageable host-to-device copies: one thread against several threads copying slices at once (is the runtime's staging copy
// the bound, and does it run in parallel?).  hipcc -O2 h2d_threads.cpp -o h2d_threads -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <algorithm>
#include <sys/mman.h>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t N = (size_t)192 << 20;
    char *h = (char *)malloc(N);
    memset(h, 1, N);
    char *d = nullptr;
    double tA = now();
    hipMalloc(&d, N);
    printf("first hipMalloc (runtime start-up) %.1f ms\n", (now() - tA) * 1e3);
    tA = now();
    hipMemcpy(d, h, 1 << 20, hipMemcpyHostToDevice);
    printf("first pageable copy, 1 MB: %.2f ms\n", (now() - tA) * 1e3);
    tA = now();
    hipMemcpy(d, h + (8 << 20), 1 << 20, hipMemcpyHostToDevice);
    printf("second pageable copy, 1 MB elsewhere: %.2f ms\n", (now() - tA) * 1e3);
    tA = now();
    hipMemcpy(d, h + (16 << 20), 32 << 20, hipMemcpyHostToDevice);
    printf("third, 32 MB elsewhere: %.2f ms\n", (now() - tA) * 1e3);
    tA = now();
    hipMemcpy(d, h + (64 << 20), 32 << 20, hipMemcpyHostToDevice);
    printf("fourth, 32 MB elsewhere: %.2f ms\n", (now() - tA) * 1e3);
    for (int T : {1, 2, 4, 8}) {
        for (int rep = 0; rep < 2; ++rep) {
            const double t0 = now();
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t] {
                    hipSetDevice(0);
                    const size_t a = N / T * t, b = (t + 1 == T) ? N : N / T * (t + 1);
                    hipMemcpy(d + a, h + a, b - a, hipMemcpyHostToDevice);
                });
            for (auto &x : th) x.join();
            const double dt = now() - t0;
            if (rep) printf("%d thread(s): %.1f ms = %.1f GB/s\n", T, dt * 1e3, N / dt / 1e9);
        }
    }
    {   // what the front end does: many chunks of a few MB, each from its own buffer, first time and again
        const int K = 64; const size_t C = (size_t)5 << 19;   // 2.5 MB
        std::vector<char *> bufs;
        for (int k = 0; k < K; ++k) { char *b = (char *)malloc(C); memset(b, k, C); bufs.push_back(b); }
        hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        for (int pass = 0; pass < 3; ++pass) {
            const double t0 = now();
            for (int k = 0; k < K; ++k) { hipMemcpyAsync(d + (size_t)k * C, bufs[k], C, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); }
            const double dt = now() - t0;
            printf("64 chunks of 2.5 MB, own buffers, pass %d: %.1f ms = %.1f GB/s\n", pass, dt * 1e3, K * C / dt / 1e9);
        }
        // the same bytes as ONE copy out of a contiguous pageable buffer that was filled by memcpy first
        char *big = (char *)malloc(K * C);
        double t0 = now();
        for (int k = 0; k < K; ++k) memcpy(big + (size_t)k * C, bufs[k], C);
        const double tm = now() - t0;
        t0 = now();
        hipMemcpyAsync(d, big, K * C, hipMemcpyHostToDevice, st); hipStreamSynchronize(st);
        const double dt = now() - t0;
        printf("gathered by memcpy %.1f ms, then one copy %.1f ms = %.1f GB/s\n", tm * 1e3, dt * 1e3, K * C / dt / 1e9);
    }
    {   // the front end's gathered arrays: 2 MB-aligned, MADV_HUGEPAGE, filled by memcpy, two arrays in flight, one wait
        const size_t A = (size_t)150 << 20, B = (size_t)52 << 20;
        for (int huge = 0; huge < 2; ++huge) {
            char *a = (char *)aligned_alloc((size_t)2 << 20, A), *b = (char *)aligned_alloc((size_t)2 << 20, B);
            if (huge) { madvise(a, A, MADV_HUGEPAGE); madvise(b, B, MADV_HUGEPAGE); }
            double t0 = now();
            for (size_t o = 0; o < A; o += (size_t)5 << 19) memcpy(a + o, h + o % ((size_t)64 << 20), std::min((size_t)5 << 19, A - o));
            for (size_t o = 0; o < B; o += (size_t)5 << 19) memcpy(b + o, h + o, std::min((size_t)5 << 19, B - o));
            const double tf = now() - t0;
            hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
            t0 = now();
            hipMemcpyAsync(d, a, A, hipMemcpyHostToDevice, st);
            const double t1 = now() - t0;
            hipMemcpyAsync(d + A, b, B > N - A ? N - A : B, hipMemcpyHostToDevice, st);
            const double t2 = now() - t0;
            hipStreamSynchronize(st);
            const double dt = now() - t0;
            printf("gathered arrays 150 + 42 MB, madvise(HUGEPAGE) %d: filled in %.1f ms; async calls return after %.1f / %.1f ms, done after %.1f ms = %.1f GB/s\n",
                   huge, tf * 1e3, t1 * 1e3, t2 * 1e3, dt * 1e3, (A + (B > N - A ? N - A : B)) / dt / 1e9);
            t0 = now();
            hipMemcpyAsync(d, a, A, hipMemcpyHostToDevice, st);
            hipStreamSynchronize(st);
            printf("   the 150 MB again: %.1f ms\n", (now() - t0) * 1e3);
            free(a); free(b);
        }
    }
    // registered (pinned in place) for comparison
    double t0 = now();
    hipHostRegister(h, N, hipHostRegisterDefault);
    const double treg = now() - t0;
    t0 = now();
    hipMemcpy(d, h, N, hipMemcpyHostToDevice);
    const double dt = now() - t0;
    printf("hipHostRegister %.1f ms; registered copy %.1f ms = %.1f GB/s\n", treg * 1e3, dt * 1e3, N / dt / 1e9);
    return 0;
}
