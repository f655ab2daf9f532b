// measuring aid: are 2-byte-aligned ds_read_b32 legal on gfx950 under ROCm (unaligned LDS mode), and what do they cost?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32; typedef unsigned short u16;
__global__ void k(u32 *out, int iters, int mode)
{
    extern __shared__ unsigned char smem[];
    const u32 lane = threadIdx.x & 63;
    for (u32 i = threadIdx.x; i < 16384; i += blockDim.x) ((u16 *)smem)[i] = (u16)(i * 7 + 1);
    __syncthreads();
    u32 acc = 0, pos = lane * 132 + ((lane * 6) & 126);       // per-lane ring rows, 2-byte-aligned offsets
    for (int it = 0; it < iters; it++) {
        u32 a = pos & 0x3ffe, v;
        if (mode == 0) { asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a)); u32 v2; asm volatile("ds_read_u16 %0, %1 offset:2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v2) : "v"(a)); v |= v2 << 16; }
        else { asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a)); }
        acc += v; pos += 2 + (v & 2);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main()
{
    u32 *d; hipMalloc(&d, 4 << 20);
    u32 *h = (u32 *)malloc(4 << 20);
    for (int mode = 0; mode < 2; mode++) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(k, dim3(1024), dim3(256), 32768 + 1024, 0, d, 2000, mode);
        hipEventRecord(a); hipLaunchKernelGGL(k, dim3(1024), dim3(256), 32768 + 1024, 0, d, 2000, mode); hipEventRecord(b);
        hipError_t e = hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, a, b);
        hipMemcpy(h, d, 4 << 20, hipMemcpyDeviceToHost);
        unsigned long long s = 0; for (int i = 0; i < 1024 * 256; i++) s += h[i];
        printf("mode %d (%s): %s, %.3f ms, checksum %llu\n", mode, mode ? "one 2-byte-aligned ds_read_b32" : "two ds_read_u16", hipGetErrorString(e), ms, s);
    }
    return 0;
}
