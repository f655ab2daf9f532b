// trc_vlc.h -- pieces shared by the Turbo-VLC integer coders (trc_rc_vlc.hip: over the CDF range coder,
// trc_ans_vlc.hip: over the CDF rANS).  Reference: vlcenc/vlcdec/bitvrput/bitvrget include_/vlcbit.h:24-63, reverse
// bit I/O rcutil_.h:163-189, zigzag rcutil_.h:142-149.
//
// An element x >= 2^(VN+1) is split into an exponent symbol and f = bsr(x) - VN mantissa bits:
//     expo = ((f+1) << VN) + bits [f, f+VN) of x,   mantissa = low f bits of x,   x = (((1 << VN) + (expo & (2^VN - 1))) << f) + mantissa
// with f = (expo >> VN) - 1 on the way back.  The value or the exponent becomes one or two CDF16 symbols: below T (8 for
// VN = 2, 12 for VN = 1) itself with table 0, else ((x-T)>>4)+T with table 0 and (x-T)&15 with table 1.  Mantissas go
// MSB-first into a bit string that grows DOWN from the end of the reference's output (byte k of the string at end[-1-k]);
// a little-endian u32 holding 32 MSB-first bits is exactly four bytes of that reversed string.
#pragma once
#include "trc_dev.h"

typedef u64 u64_a1 __attribute__((aligned(1)));

// MSB-first bit string growing DOWN from `end` (byte k of the string at end[-1-k])
struct LaneBitsDown {
    u8 *end;             // one past the region's last byte (4-byte aligned)
    u64 acc;             // pending bits, from bit 63 down
    u32 nacc;            // pending bits (< 32 between calls)
    u32 nwords;          // 32-bit groups already stored
    u32 total;           // bits appended so far
    __device__ __forceinline__ void start(u8 *e) { end = e; acc = 0; nacc = 0; nwords = 0; total = 0; }
    __device__ __forceinline__ void put_if(bool take, u32 f, u32 ma)       // f <= 30 bits of ma
    {
        const u32 ff = take ? f : 0u;
        acc |= (u64)(take ? ma : 0u) << ((64u - nacc - ff) & 63u);
        nacc += ff; total += ff;
        if (nacc >= 32u) {
            *(u32 *)(end - 4u * (nwords + 1u)) = (u32)(acc >> 32);
            acc <<= 32; nacc -= 32u; nwords++;
        }
    }
    __device__ __forceinline__ u32 bytes() const { return 4u * nwords + ((nacc + 7u) >> 3); }
    __device__ __forceinline__ void finish(bool ok)
    {
        const u32 nb = (nacc + 7u) >> 3;
        for (u32 j = 0; j < nb; j++) if (ok) end[-(int)(4u * nwords + 1u + j)] = (u8)(acc >> (56u - 8u * j));
    }
};

__device__ __forceinline__ u32 vlc_zigzag_enc(u32 d, bool wide) { return wide ? (d << 1) ^ (u32)((int)d >> 31) : ((d << 1) ^ (u32)((int)(short)d >> 15)) & 0xffffu; }
__device__ __forceinline__ u32 vlc_zigzag_dec(u32 x) { return (x >> 1) ^ (0u - (x & 1u)); }

