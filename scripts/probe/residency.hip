// residency.hip -- how many workgroups of a given footprint (threads, dynamic LDS) does an MI355X hold AT ONCE, and where?
// The model-per-lane coders live on "one residency round": their launch time is (rounds) x (one wave's time), so whether 1018
// workgroups of 36 / 39.5 / 40 KiB are co-resident on 256 CUs decides a factor two.  Every workgroup notes its start time and its
// hardware place, then spins for `hold` microseconds; the host counts how many started before the first one finished (= resident
// together), how many had to wait for a slot, and the spread over XCDs / CUs.  A second launch right behind a first one of a
// DIFFERENT LDS size shows whether the LDS allocator of a CU fragments.
// build: hipcc -O2 --offload-arch=gfx950 residency.hip -o residency
// usage: residency <workgroups> <threads> <lds bytes> [hold_us] [prior_lds bytes] [prior_threads]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <vector>

struct Rec { unsigned long long t0, t1; unsigned hw, xcc; };

__global__ void hold_kernel(Rec *out, unsigned long long hold_ticks)
{
    extern __shared__ unsigned char smem[];
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) smem[0] = 1;                          // touch the allocation
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    while (wall_clock64() - t0 < hold_ticks) __builtin_amdgcn_s_sleep(8);
    if ((threadIdx.x & 63u) == 0) { Rec r; r.t0 = t0; r.t1 = wall_clock64(); r.hw = hw; r.xcc = xcc; out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = r; }
}

static void run(unsigned wgs, unsigned threads, unsigned lds, double hold_us, Rec *d_out, std::vector<Rec> &h)
{
    (void)hipFuncSetAttribute((const void *)hold_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const unsigned long long ticks = (unsigned long long)(hold_us * 100.0);      // wall_clock64: 100 MHz
    hipLaunchKernelGGL(hold_kernel, dim3(wgs), dim3(threads), lds, 0, d_out, ticks);
    h.resize(wgs * (threads / 64));
}

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: residency <workgroups> <threads> <lds> [hold_us] [prior_lds]\n"); return 2; }
    const unsigned wgs = (unsigned)atoi(argv[1]), threads = (unsigned)atoi(argv[2]), lds = (unsigned)atoi(argv[3]);
    const double hold_us = argc > 4 ? atof(argv[4]) : 200.0;
    const unsigned prior = argc > 5 ? (unsigned)atoi(argv[5]) : 0;
    const unsigned prior_threads = argc > 6 ? (unsigned)atoi(argv[6]) : 64;
    const unsigned wpg = threads / 64, nw = wgs * wpg;          // every WAVE notes its place
    Rec *d_out; (void)hipMalloc(&d_out, sizeof(Rec) * (nw + 4096));
    std::vector<Rec> h;
    int occ = -1;
    (void)hipFuncSetAttribute((const void *)hold_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)hold_kernel, (int)threads, lds);
    if (prior) {                                                // a short launch with another LDS size right in front (same stream)
        std::vector<Rec> tmp;
        run(prior_threads == 64 ? 1024 : 1018, prior_threads, prior, 20.0, d_out + nw, tmp);
    }
    run(wgs, threads, lds, hold_us, d_out, h);
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    (void)hipMemcpy(h.data(), d_out, sizeof(Rec) * nw, hipMemcpyDeviceToHost);
    unsigned long long first_end = ~0ull, t_min = ~0ull, t_max = 0;
    for (auto &r : h) { first_end = std::min(first_end, r.t1); t_min = std::min(t_min, r.t0); t_max = std::max(t_max, r.t1); }
    unsigned together = 0;
    std::map<unsigned, unsigned> per_cu, per_xcc, per_simd;
    for (auto &r : h) if (r.t0 < first_end) {
        together++;
        const unsigned simd = (r.hw >> 4) & 3, cu = (r.hw >> 8) & 0xf, sh = (r.hw >> 12) & 1, se = (r.hw >> 13) & 7, xcc = r.xcc & 0xf;
        per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
        per_simd[(xcc << 14) | (se << 10) | (sh << 6) | (cu << 2) | simd]++;
        per_xcc[xcc]++;
    }
    std::map<unsigned, unsigned> shist;
    for (auto &kv : per_simd) shist[kv.second]++;
    // which waves of a workgroup share a SIMD?  pairs[d] = workgroups' wave pairs (w, w + d) found on the same SIMD
    unsigned pairs[16] = {0}, wgs_seen = 0;
    if (wpg > 1) for (unsigned g = 0; g < wgs; g++) {
        wgs_seen++;
        for (unsigned a = 0; a < wpg; a++) for (unsigned b = a + 1; b < wpg; b++)
            if (((h[g * wpg + a].hw >> 4) & 3) == ((h[g * wpg + b].hw >> 4) & 3)) pairs[b - a]++;
    }
    std::map<unsigned, unsigned> hist;
    for (auto &kv : per_cu) hist[kv.second]++;
    printf("wgs %u x %u threads, lds %u B (prior launch: lds %u, %u threads): occupancy API says %d per CU; waves resident together %u, late %u; "
           "kernel %.1f us for hold %.0f us -> %.2f rounds\n", wgs, threads, lds, prior, prior ? prior_threads : 0, occ, together, nw - together,
           (t_max - t_min) / 100.0, hold_us, (t_max - t_min) / 100.0 / hold_us);
    printf("  CUs seen %zu; waves per CU (first round): ", per_cu.size());
    for (auto &kv : hist) printf("%u CUs x %u, ", kv.second, kv.first);
    printf("\n  SIMDs seen %zu of 1024; waves per SIMD (first round): ", per_simd.size());
    for (auto &kv : shist) printf("%u SIMDs x %u, ", kv.second, kv.first);
    if (wpg > 1) {
        printf("\n  waves of a workgroup on the same SIMD, by distance in the workgroup: ");
        for (unsigned d = 1; d < wpg && d < 16; d++) printf("+%u: %u  ", d, pairs[d]);
        printf("(of %u workgroups)", wgs_seen);
    }
    printf("\n  per XCD: ");
    for (auto &kv : per_xcc) printf("%u:%u ", kv.first, kv.second);
    printf("\n");
    return 0;
}
