// vcc_hazard.hip -- does a VALU read of VCC / an SGPR pair straight behind the VALU write of it ever see a stale value on gfx950?
//
// hipcc for gfx950 keeps two wait states between a VALU write of an SGPR / VCC and a VALU read of it (LLVM GCNHazardRecognizer,
// hasVDecCoExecHazard(): gfx940+; `v_sub_co / s_nop 1 / v_subb_co` in its own 64-bit subtracts).  Rounds 3-4 of this tree shipped
// hand-written carry chains with ZERO wait states; the parity suite passed, which proves nothing about conditions it never ran
// under.  This probe runs the same chains back to back, lane by lane against 64-bit arithmetic the compiler generates (and pads),
// at 1 / 2 / 4 / 8 waves per SIMD, next to co-resident waves that hammer the scalar unit, the LDS, the vector memory path or the
// VALU.  It can only ever show that the hazard EXISTS (one mismatch is enough); zero mismatches is "not observed", not a proof:
// the product's kernels carry the wait states either way (round 5).
//
//   workgroup = 4 x K waves; layer 0 (waves 0-3, one per SIMD) always tests; layers 1.. run `--mix`:
//   0 testers too | 1 SALU | 2 LDS | 3 VMEM | 4 VALU (mul / cmp / cndmask) | 5 one of each, by layer
// build: hipcc -O2 --offload-arch=gfx950 vcc_hazard.hip -o vcc_hazard
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef unsigned u32;
typedef unsigned long long u64;
#define NT 7

__device__ __forceinline__ u32 xs(u32 &s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

__device__ void tester(u32 *err, int trips, u32 seed)
{
    u32 s = seed ^ (blockIdx.x * 1024u + threadIdx.x) * 2654435761u; if (!s) s = 1;
    u32 bad[NT] = { 0, 0, 0, 0, 0, 0, 0 };
    for (int t = 0; t < trips; t++) {
        u32 a = xs(s), b = xs(s), c = xs(s), d = xs(s);
        if (t & 1) { b = ~a + (c & 3u); }                               // sums right at the carry boundary
        if ((t & 6) == 2) { c = d = 0xffffffffu; }
        // T0: add_co -> addc (64-bit add)
        { u32 lo, hi;
          asm volatile("v_add_co_u32_e32 %0, vcc, %2, %3\n\tv_addc_co_u32_e32 %1, vcc, %4, %5, vcc" : "=&v"(lo), "=&v"(hi) : "v"(a), "v"(b), "v"(c), "v"(d) : "vcc");
          const u64 r = (((u64)c << 32) | a) + (((u64)d << 32) | b);
          bad[0] += (lo != (u32)r) | (hi != (u32)(r >> 32)); }
        // T1: sub_co -> subb -> subb_e64 (the bit step of the bitwise decoder: difference and borrow mask)
        { u32 lo, hi, m;
          asm volatile("v_sub_co_u32_e32 %0, vcc, %3, %4\n\tv_subb_co_u32_e32 %1, vcc, %5, %6, vcc\n\tv_subb_co_u32_e64 %2, vcc, 0, 0, vcc"
                       : "=&v"(lo), "=&v"(hi), "=&v"(m) : "v"(a), "v"(b), "v"(c), "v"(d) : "vcc");
          const u64 X = ((u64)c << 32) | a, Y = ((u64)d << 32) | b, r = X - Y;
          bad[1] += (lo != (u32)r) | (hi != (u32)(r >> 32)) | (m != (X < Y ? ~0u : 0u)); }
        // T2: cmp -> cndmask through VCC
        { u32 r;
          asm volatile("v_cmp_lt_u32_e32 vcc, %1, %2\n\tv_cndmask_b32_e32 %0, %3, %4, vcc" : "=&v"(r) : "v"(a), "v"(b), "v"(c), "v"(d) : "vcc");
          bad[2] += r != (a < b ? d : c); }
        // T3: the renormalisation block: subrev_co -> subb_e64 -> addc
        { u32 t0, m, cnt = d;
          const u32 x = (t & 8) ? (a & 1u) : a;
          asm volatile("v_subrev_co_u32_e32 %1, vcc, 1, %3\n\tv_subb_co_u32_e64 %0, vcc, 0, 0, vcc\n\tv_addc_co_u32_e32 %2, vcc, 0, %2, vcc"
                       : "=&v"(m), "=&v"(t0), "+v"(cnt) : "v"(x) : "vcc");
          bad[3] += (m != (x == 0 ? ~0u : 0u)) | (cnt != d + (x == 0 ? 1u : 0u)) | (t0 != x - 1u); }
        // T4: cmp into an SGPR pair -> cndmask_e64
        { u32 r;
          asm volatile("v_cmp_lt_u32_e64 s[20:21], %1, %2\n\tv_cndmask_b32_e64 %0, %3, %4, s[20:21]" : "=&v"(r) : "v"(a), "v"(b), "v"(c), "v"(d) : "s20", "s21");
          bad[4] += r != (a < b ? d : c); }
        // T5: three limbs: add_co -> addc -> addc (low += cut with the carry limb of the bitwise encoder)
        { u32 l0 = a, l1 = c, l2 = d & 1u;
          asm volatile("v_add_co_u32_e32 %0, vcc, %0, %3\n\tv_addc_co_u32_e32 %1, vcc, %1, %4, vcc\n\tv_addc_co_u32_e32 %2, vcc, 0, %2, vcc"
                       : "+v"(l0), "+v"(l1), "+v"(l2) : "v"(b), "v"(d) : "vcc");
          const u64 r0 = (u64)a + b, r1 = (u64)c + d + (r0 >> 32);
          bad[5] += (l0 != (u32)r0) | (l1 != (u32)r1) | (l2 != (d & 1u) + (u32)(r1 >> 32)); }
        // T6: addc reading a carry written by a v_cmp (static range decoder's cursor advance)
        { u32 r = c;
          asm volatile("v_cmp_gt_u32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, 0, %0, vcc" : "+v"(r) : "v"(a), "v"(b) : "vcc");
          bad[6] += r != c + (a > b ? 1u : 0u); }
    }
    for (int i = 0; i < NT; i++) if (bad[i]) atomicAdd(&err[i], bad[i]);
}

__global__ __launch_bounds__(1024) void hz_kernel(u32 *err, int trips, int mix, u32 seed, u32 *gbuf, u32 gmask)
{
    extern __shared__ u32 lds[];
    const u32 wave = threadIdx.x >> 6, layer = wave >> 2, lane = threadIdx.x & 63u;
    volatile u32 *done = (volatile u32 *)&lds[0];
    if (threadIdx.x == 0) lds[0] = 0;
    for (u32 i = 1 + threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i * 2654435761u;
    __syncthreads();
    int role = layer == 0 ? 0 : mix == 5 ? 1 + (int)((layer - 1u) & 3u) : mix;
    if (role == 0) {
        tester(err, trips, seed);
        if (layer == 0 && lane == 0) atomicAdd((u32 *)&lds[0], 1u);
        return;
    }
    u32 acc = lane + wave * 64u;
    while (*done < 4u) {                                               // until the four layer-0 testers have finished
        if (role == 1) {                                               // scalar unit: dependent s_* chains, VCC / SGPR-pair writes from the SALU
            asm volatile("s_mov_b32 s22, 1\n\t"
                         ".rept 64\n\ts_add_u32 s22, s22, s22\n\ts_and_b64 s[24:25], exec, vcc\n\ts_lshl_b32 s23, s22, 3\n\ts_or_b64 s[24:25], s[24:25], exec\n\ts_mul_i32 s23, s23, s22\n\t.endr"
                         ::: "s22", "s23", "s24", "s25", "scc");
        } else if (role == 2) {                                        // LDS: reads and writes, mixed widths
            const u32 ad = 16u * (1u + ((acc * 37u) & 511u));
            asm volatile(".rept 16\n\tds_read_b64 v[40:41], %0\n\tds_write_b32 %0, %1 offset:8192\n\tds_read_u8 v42, %0 offset:3\n\tds_read_b128 v[44:47], %0\n\t.endr\n\ts_waitcnt lgkmcnt(0)"
                         :: "v"(ad), "v"(acc) : "v40", "v41", "v42", "v44", "v45", "v46", "v47", "memory");
            acc += 17u;
        } else if (role == 3) {                                        // vector memory: scattered loads
            u32 x = acc;
            for (int i = 0; i < 8; i++) { x = x * 1664525u + 1013904223u; acc ^= gbuf[x & gmask]; }
        } else {                                                       // VALU: quarter-rate multiplies, compares, selects (their own VCC)
            asm volatile(".rept 32\n\tv_mul_lo_u32 %0, %0, %0\n\tv_cmp_gt_u32_e32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %0, %1, vcc\n\tv_mul_hi_u32 %0, %0, %1\n\t.endr"
                         : "+v"(acc) : "v"(lane | 3u) : "vcc");
        }
    }
    if (acc == 0x12345u) gbuf[0] = acc;
}

int main(int argc, char **argv)
{
    int trips = 100000, reps = 1;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--trips") && i + 1 < argc) trips = atoi(argv[++i]);
        if (!strcmp(argv[i], "--reps") && i + 1 < argc) reps = atoi(argv[++i]);
    }
    u32 *err, *gbuf; const u32 gwords = 1u << 26;                      // 256 MiB: beyond the L2s
    (void)hipMalloc(&err, NT * 4); (void)hipMalloc(&gbuf, (size_t)gwords * 4); (void)hipMemset(gbuf, 1, (size_t)gwords * 4);
    const int pairs_per_iter = 1 + 2 + 1 + 2 + 1 + 2 + 1;
    const char *mixname[6] = { "testers only", "SALU", "LDS", "VMEM", "VALU", "one of each" };
    const char *tname[NT] = { "add_co>addc", "sub_co>subb>subb64", "cmp>cndmask", "subrev_co>subb64>addc", "cmp_e64>cndmask_e64", "add_co>addc>addc", "cmp>addc" };
    double total = 0; u32 total_bad = 0;
    (void)hipFuncSetAttribute((const void *)hz_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int rep = 0; rep < reps; rep++)
    for (int K : { 1, 2, 4, 8 }) for (int mix = 0; mix < 6; mix++) {
        if (K == 1 && mix) continue;
        const int wgs = K == 8 ? 2 : 1, waves = K == 8 ? 16 : 4 * K;
        const size_t lds = K == 8 ? 70 * 1024 : 100 * 1024;
        (void)hipMemset(err, 0, NT * 4);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(hz_kernel, dim3(256 * wgs * 4), dim3(64 * waves), lds, 0, err, trips, mix, 0x9e3779b9u * (u32)(rep * 64 + K * 8 + mix + 1), gbuf, gwords - 1);
        (void)hipEventRecord(e1);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        u32 h[NT]; (void)hipMemcpy(h, err, NT * 4, hipMemcpyDeviceToHost);
        const int testers = mix == 0 ? waves : 4;
        const double pairs = (double)256 * wgs * 4 * testers * 64 * trips * pairs_per_iter;
        total += pairs;
        u32 bad = 0; for (int i = 0; i < NT; i++) bad += h[i];
        total_bad += bad;
        printf("K=%d waves/SIMD  co-resident: %-13s  %8.1f ms  %.3g lane-pairs  mismatches:", K, mixname[mix], ms, pairs);
        for (int i = 0; i < NT; i++) printf(" %u", h[i]);
        printf("\n");
        if (bad) for (int i = 0; i < NT; i++) if (h[i]) printf("   !! %s: %u mismatching lane-iterations\n", tname[i], h[i]);
    }
    printf("TOTAL %.4g hazard pairs checked lane by lane, %u mismatching lane-iterations\n", total, total_bad);
    return total_bad ? 1 : 0;
}
