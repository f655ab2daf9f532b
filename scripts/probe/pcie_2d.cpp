// pcie_2d.cpp -- measuring aid (not product): how fast does hipMemcpy2DAsync move the k-th 1/K of every chunk (rows of chunk / K
// bytes, pitch = chunk) between page-locked host memory and the device?  The question behind it: could a host-pointer call deliver the
// first part of EVERY chunk first, so that the coder waves start before the whole input is there?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main()
{
    const size_t n = 100u * 1000 * 1000;
    unsigned char *p, *d; CK(hipHostMalloc(&p, n + 65536)); CK(hipMalloc(&d, n + 65536));
    memset(p, 3, n);
    hipStream_t s; CK(hipStreamCreate(&s));
    for (size_t chunk : { (size_t)4096, (size_t)16384 })
        for (size_t K : { (size_t)1, (size_t)2, (size_t)4, (size_t)8, (size_t)16 }) {
            const size_t w = chunk / K, rows = n / chunk;
            double best[2] = { 1e9, 1e9 };
            for (int dir = 0; dir < 2; dir++)
                for (int r = 0; r < 3; r++) {
                    CK(hipStreamSynchronize(s));
                    const double t0 = now();
                    for (size_t k = 0; k < K; k++) {
                        if (dir == 0) CK(hipMemcpy2DAsync(d + k * w, chunk, p + k * w, chunk, w, rows, hipMemcpyHostToDevice, s));
                        else CK(hipMemcpy2DAsync(p + k * w, chunk, d + k * w, chunk, w, rows, hipMemcpyDeviceToHost, s));
                    }
                    CK(hipStreamSynchronize(s));
                    const double t = now() - t0;
                    if (t < best[dir]) best[dir] = t;
                }
            printf("chunk %5zu  K %2zu  row %5zu B x %zu rows x %zu passes:  H2D %6.1f GB/s   D2H %6.1f GB/s\n", chunk, K, w, rows, K,
                   rows * chunk / best[0] / 1e9, rows * chunk / best[1] / 1e9);
        }
    return 0;
}
