// trc_gather.h -- the payload gather done by the ENCODER's own waves (round 5, static rANS; the standalone kernel: trc_dir.hip).
//
// The container wants the chunks' payloads back to back, so a chunk's place is the sum of every earlier chunk's coded length:
// rounds 1-5 staged the words in per-chunk scratch regions and ran trc_gather_kernel behind the encoder -- 29.8 of the headline
// step's 148 us for a copy of 65 MB, its workgroups scheduled wherever, its reads served by the fabric.  In a launch that is ONE
// residency round every encoder wave is on the chip until the end anyway, and all of them end within a few microseconds of each
// other (TrcPace), so the prefix can be had inside the launch:
//   * workgroups take a TICKET (their place in the container) from a counter instead of blockIdx -- a workgroup only ever waits
//     for lower tickets, and those have started: no deadlock whatever else runs on the device;
//   * a workgroup's waves leave their group sums in LDS; wave 0 publishes the workgroup's total in the sync area (one u64 per
//     ticket, bit 63 = valid: the area is zero between calls -- trc_static_prep_kernel zeroes it, the launch's last poller
//     zeroes it again) and polls the totals of the lower tickets with device-scope loads: at most 255 words, four per lane;
//   * then every wave moves its own 64 pieces -- bytes it has just written, on its own XCD -- with the walk of
//     trc_gather_kernel<1> (four lanes per piece, 16 pieces per pass, dst-aligned 16-byte stores).
#pragma once
#include "trc_dev.h"

#define TRC_SYNC_PUB     0u        // u64[256]  workgroup totals by ticket
#define TRC_SYNC_TICKET  2048u     // u32
#define TRC_SYNC_DONE    2052u     // u32       workgroups that have finished polling
#define TRC_SYNC_BYTES   2112u
#define TRC_SYNC_MAX_WG  256u
#define TRC_SYNC_VALID   (1ull << 63)

// (round 6, ADVICE r5: publish with release, poll with acquire; the host zeroes the area in front of every fused launch --
// trc_launch_ans4s_enc -- so a launch that was aborted, or a workspace shared by two encodes, cannot leave stale tickets behind;
// and the poll is BOUNDED: a ticket that never arrives ends the kernel in a trap, a launch failure the caller sees, not a hang)
__device__ __forceinline__ u64 trc_sync_load(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void trc_sync_store(u64 *p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
#define TRC_SYNC_SPIN_MAX (1u << 24)                            // polls of ~0.3 us: seconds, where a launch lasts 50 us

// Sum of the totals published by tickets below `t` (one wave, all lanes call; t <= TRC_SYNC_MAX_WG).  Spins until each is there.
__device__ __forceinline__ u64 trc_sync_prefix(const u64 *pub, u32 t)
{
    const u32 lane = trc_lane();
    u64 v[4] = { 0, 0, 0, 0 };
    bool pend;
    u32 spins = 0;
    do {
        if (++spins > TRC_SYNC_SPIN_MAX) __builtin_trap();
        pend = false;
#pragma unroll
        for (u32 k = 0; k < 4u; k++) {
            const u32 j = lane + 64u * k;
            if (j < t && !v[k]) v[k] = trc_sync_load(pub + j);
        }
#pragma unroll
        for (u32 k = 0; k < 4u; k++) pend = pend || (lane + 64u * k < t && !v[k]);
        pend = __ballot(pend) != 0;
        if (pend) __builtin_amdgcn_s_sleep(4);
    } while (pend);
    u64 acc = (v[0] & ~TRC_SYNC_VALID) + (v[1] & ~TRC_SYNC_VALID) + (v[2] & ~TRC_SYNC_VALID) + (v[3] & ~TRC_SYNC_VALID);
    u32 lo = (u32)acc, hi = (u32)(acc >> 32);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const u32 l2 = (u32)__shfl_xor((int)lo, d, 64), h2 = (u32)__shfl_xor((int)hi, d, 64);
        const u64 s = ((((u64)hi) << 32) | lo) + ((((u64)h2) << 32) | l2);
        lo = (u32)s; hi = (u32)(s >> 32);
    }
    return (((u64)hi) << 32) | lo;
}

// One wave moves the 64 pieces of its group to dst0 (= payload + the group's base): lane i holds piece i's length `l` and where its
// bytes are (`src`, 2-byte aligned).  ex_s = u32[65], src_s = u64[64]: LDS of this wave alone.  All lanes call.
__device__ __forceinline__ void trc_wave_gather64(u8 *dst0, u32 *ex_s, u64 *src_s, u32 l, const u8 *src)
{
    constexpr u32 NP = 64u, TPP = 4u, VPT = 6u;
    const u32 lane = trc_lane();
    const u32 inc = trc_wave_incl_scan(l);
    ex_s[lane] = inc - l;
    src_s[lane] = (u64)(uintptr_t)src;
    if (lane == 63u) ex_s[NP] = inc;
    trc_wave_lds_fence();
    const u32 tot = ex_s[NP];
    auto src_of = [&](u32 p) -> const u8 * { return (const u8 *)(uintptr_t)src_s[p]; };

    u32 head = (u32)((16u - ((uintptr_t)dst0 & 15u)) & 15u);
    if (head > tot) head = tot;
    const u32 nvec = (tot - head) >> 4;
    const u32 tail0 = head + (nvec << 4);
    for (u32 b = lane; b < head + (tot - tail0); b += 64u) {    // bytes before the first / after the last aligned vector
        const u32 d = b < head ? b : tail0 + (b - head);
        u32 lo = 0, hi = NP;
        while (lo + 1 < hi) { const u32 mid = (lo + hi) >> 1; if (ex_s[mid] <= d) lo = mid; else hi = mid; }
        dst0[d] = src_of(lo)[d - ex_s[lo]];
    }
#pragma nounroll
    for (u32 pass = 0; pass < NP / 16u; pass++) {
        const u32 k = pass * 16u + (lane >> 2), sub = lane & 3u;
        const u32 e0 = ex_s[k], e1 = ex_s[k + 1];
        // vectors whose first byte lies in [e0, e1): first index = ceil((e0 - head)/16) (0 if e0 <= head), end likewise from e1
        const u32 v_lo = e0 <= head ? 0u : (e0 - head + 15u) >> 4;
        u32 v_hi = e1 <= head ? 0u : (e1 - head + 15u) >> 4;
        if (v_hi > nvec) v_hi = nvec;
        const u8 *sa = src_of(k);
        const u8 *sb = k + 1 < NP ? src_of(k + 1) : sa;
        const u32 e2 = ex_s[k + 2 > NP ? NP : k + 2];
        const u32 vl = v_hi - 1u;                               // only the piece's last vector can straddle its end
        for (u32 v0 = v_lo + sub; v0 < v_hi; v0 += VPT * TPP) {
            uint4 a[VPT], b[VPT];
            u32 d[VPT];
            bool ok[VPT];
#pragma unroll
            for (int j = 0; j < (int)VPT; j++) {
                const u32 v = v0 + TPP * (u32)j;
                ok[j] = v < v_hi;
                d[j] = head + ((ok[j] ? v : v0) << 4);
                a[j] = trc_ld16_a2(sa + (d[j] - e0));
                b[j] = a[j];
            }
#pragma unroll
            for (int j = 0; j < (int)VPT; j++) {
                const u32 v = v0 + TPP * (u32)j;
                if (ok[j] && v == vl && d[j] + 16u > e1 && d[j] + 16u <= e2) b[j] = trc_ld16_a2(sb - (e1 - d[j]));
            }
#pragma unroll
            for (int j = 0; j < (int)VPT; j++) {
                if (!ok[j]) continue;
                const u32 sp = e1 - d[j];                       // bytes of this vector inside piece k (>= 16: all)
                if (sp >= 16u) { *(uint4 *)(dst0 + d[j]) = a[j]; continue; }
                if (d[j] + 16u <= e2) {
                    const u32 aw[4] = { a[j].x, a[j].y, a[j].z, a[j].w }, bw[4] = { b[j].x, b[j].y, b[j].z, b[j].w };
                    u32 r[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int nb = (int)sp - 4 * q;
                        const u32 m = nb >= 4 ? 0xffffffffu : nb <= 0 ? 0u : ((1u << (8 * nb)) - 1u);
                        r[q] = (aw[q] & m) | (bw[q] & ~m);
                    }
                    *(uint4 *)(dst0 + d[j]) = make_uint4(r[0], r[1], r[2], r[3]);
                } else {                                       // three or more pieces inside 16 bytes: byte by byte
                    u32 kk = k;
                    const u8 *p = sa;
                    for (u32 q = 0; q < 16; q++) {
                        while (d[j] + q >= ex_s[kk + 1]) { kk++; p = src_of(kk); }
                        dst0[d[j] + q] = p[d[j] + q - ex_s[kk]];
                    }
                }
            }
        }
    }
}
