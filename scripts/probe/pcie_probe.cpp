// pcie_probe.cpp -- measuring aid (not product): what the host-pointer layer can expect from this box.
//   pageable hipMemcpy, hipHostRegister cost, registered and pinned-staging transfer rates, both directions at once.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main()
{
    const size_t n = 100u * 1000 * 1000;
    unsigned char *h = (unsigned char *)aligned_alloc(4096, n + 4096), *h2 = (unsigned char *)aligned_alloc(4096, n + 4096);
    memset(h, 1, n); memset(h2, 2, n);
    unsigned char *d, *d2; CK(hipMalloc(&d, n)); CK(hipMalloc(&d2, n));
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    for (int r = 0; r < 3; r++) {
        double t0 = now(); CK(hipMemcpy(d, h, n, hipMemcpyHostToDevice)); double t1 = now();
        CK(hipMemcpy(h2, d, n, hipMemcpyDeviceToHost)); double t2 = now();
        printf("pageable        H2D %.1f GB/s  D2H %.1f GB/s\n", n / (t1 - t0) / 1e9, n / (t2 - t1) / 1e9);
    }
    for (int r = 0; r < 3; r++) {
        double t0 = now(); CK(hipHostRegister(h, n, hipHostRegisterDefault)); double t1 = now();
        CK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); double t2 = now();
        CK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1)); double t3 = now();
        CK(hipHostUnregister(h)); double t4 = now();
        printf("register %.2f ms  unregister %.2f ms | registered H2D %.1f GB/s  D2H %.1f GB/s\n", (t1 - t0) * 1e3, (t4 - t3) * 1e3, n / (t2 - t1) / 1e9, n / (t3 - t2) / 1e9);
    }
    unsigned char *p, *p2; CK(hipHostMalloc(&p, n)); CK(hipHostMalloc(&p2, n));
    for (int r = 0; r < 3; r++) {
        double t0 = now(); CK(hipMemcpyAsync(d, p, n, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); double t1 = now();
        CK(hipMemcpyAsync(p2, d2, n, hipMemcpyDeviceToHost, s2)); CK(hipStreamSynchronize(s2)); double t2 = now();
        CK(hipMemcpyAsync(d, p, n, hipMemcpyHostToDevice, s1)); CK(hipMemcpyAsync(p2, d2, n, hipMemcpyDeviceToHost, s2));
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2)); double t3 = now();
        printf("pinned          H2D %.1f GB/s  D2H %.1f GB/s  both at once %.1f GB/s each\n", n / (t1 - t0) / 1e9, n / (t2 - t1) / 1e9, n / (t3 - t2) / 1e9);
    }
    for (int r = 0; r < 3; r++) {          // pageable, both directions "at once" from ONE host thread, whole buffers and 8 MB slices
        double t0 = now();
        CK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s1)); double ta = now(); CK(hipMemcpyAsync(h2, d2, n, hipMemcpyDeviceToHost, s2)); double tb = now();
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2)); double t1 = now();
        const size_t sl = 8u << 20;
        for (size_t o = 0; o < n; o += sl) {
            const size_t l = n - o < sl ? n - o : sl;
            CK(hipMemcpyAsync(d + o, h + o, l, hipMemcpyHostToDevice, s1)); CK(hipMemcpyAsync(h2 + o, d2 + o, l, hipMemcpyDeviceToHost, s2));
        }
        double tc = now();
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2)); double t2 = now();
        printf("pageable both, one thread: whole %.1f GB/s each (calls returned after %.2f / %.2f ms of %.2f) | 8 MB slices %.1f GB/s each (issue %.2f of %.2f ms)\n",
               n / (t1 - t0) / 1e9, (ta - t0) * 1e3, (tb - t0) * 1e3, (t1 - t0) * 1e3, n / (t2 - t1) / 1e9, (tc - t1) * 1e3, (t2 - t1) * 1e3);
    }
    for (int r = 0; r < 3; r++) {          // ... from TWO host threads
        double t0 = now();
        std::thread a([&] { const size_t sl = 8u << 20; for (size_t o = 0; o < n; o += sl) { const size_t l = n - o < sl ? n - o : sl; CK(hipMemcpyAsync(d + o, h + o, l, hipMemcpyHostToDevice, s1)); } CK(hipStreamSynchronize(s1)); });
        std::thread b([&] { const size_t sl = 8u << 20; for (size_t o = 0; o < n; o += sl) { const size_t l = n - o < sl ? n - o : sl; CK(hipMemcpyAsync(h2 + o, d2 + o, l, hipMemcpyDeviceToHost, s2)); } CK(hipStreamSynchronize(s2)); });
        a.join(); b.join();
        double t1 = now();
        printf("pageable both, two threads, 8 MB slices: %.1f GB/s each\n", n / (t1 - t0) / 1e9);
    }
    for (int nt : {1, 2, 4, 8, 16}) {
        double best = 1e9;
        for (int r = 0; r < 3; r++) {
            double t0 = now();
            std::vector<std::thread> th;
            for (int t = 0; t < nt; t++) th.emplace_back([&, t] { size_t a = n / nt * t, b = t == nt - 1 ? n : n / nt * (t + 1); memcpy(p + a, h + a, b - a); });
            for (auto &x : th) x.join();
            double t1 = now(); if (t1 - t0 < best) best = t1 - t0;
        }
        printf("host memcpy pageable -> pinned, %2d threads: %.1f GB/s\n", nt, n / best / 1e9);
    }
    return 0;
}
