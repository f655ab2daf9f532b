// trc_nibmodel.h -- the adaptive 16-symbol CDF model ("CDF16", rate 7) of the reference, per lane,
// in LDS.  Reference: init CDF16DEC0/1 cdf_.h:26-27,40-41 (cdf[j] = j<<11); update cdf16upd
// cdf_.h:46-50 (AVX2) == :87-97 (SSE) -- the SIMD rule is normative, the scalar fallback of
// cdf_.h:112-117 is a different formula.  After coding symbol x whose lower bound was
// c = t[x] (read BEFORE the update), in int16 lanes with arithmetic shift:
//        t[i] += ((10*i - t[i]) + (t[i] > c ? 32736 : 0)) >> 7          i = 0..15
// A byte model is one "hi" table plus 16 "lo" tables selected by the hi nibble (rccdf.c:202,
// anscdf.c:574-575): 17 x 16 x u16 = 544 B per lane.  Entry 16 (= 32768) is never stored.
//
// gfx950 mapping: a table is 32 B = 8 dwords of packed int16 pairs; the update is 7 packed-16
// VALU ops per dword (v_pk_sub_i16 / v_pk_ashrrev_i16 / v_and / v_pk_add_i16), tables are moved
// with two ds_read_b128 + two ds_write_b128.  Lane rows are 560 B apart (140 dwords: 16
// consecutive lanes x 4 banks tile all 64 banks, so equal-offset b128 accesses are conflict free).
#pragma once
#include "trc_dev.h"

typedef short trc_s2 __attribute__((ext_vector_type(2)));

#define TRC_NIB_ROW    560u                       // bytes per lane
#define TRC_NIB_BYTES  (64u * TRC_NIB_ROW)        // 35840 per wave

struct NibTable { u32 d[8]; };                    // 16 x u16, entry 2k in the low half of d[k]

__device__ __forceinline__ u32 trc_pk(u32 lo, u32 hi) { return (lo & 0xffffu) | (hi << 16); }
__device__ __forceinline__ trc_s2 trc_as_s2(u32 v) { return __builtin_bit_cast(trc_s2, v); }
__device__ __forceinline__ u32 trc_as_u32(trc_s2 v) { return __builtin_bit_cast(u32, v); }

struct NibModel {
    u8 *row;                                      // this lane's 544 bytes in LDS
    __device__ __forceinline__ void reset()
    {
        for (u32 t = 0; t < 17; t++)
#pragma unroll
            for (u32 k = 0; k < 8; k++) ((u32 *)(row + t * 32u))[k] = trc_pk((2 * k) << 11, (2 * k + 1) << 11);
    }
    __device__ __forceinline__ u8 *table(u32 t) const { return row + t * 32u; }      // t = 0: hi, 1 + h: lo[h]
    __device__ __forceinline__ NibTable load(const u8 *tb) const
    {
        NibTable T;
        const uint4 a = *(const uint4 *)tb, b = *(const uint4 *)(tb + 16);
        T.d[0] = a.x; T.d[1] = a.y; T.d[2] = a.z; T.d[3] = a.w; T.d[4] = b.x; T.d[5] = b.y; T.d[6] = b.z; T.d[7] = b.w;
        return T;
    }
    __device__ __forceinline__ void store(u8 *tb, const NibTable &T) const
    {
        *(uint4 *)tb = make_uint4(T.d[0], T.d[1], T.d[2], T.d[3]);
        *(uint4 *)(tb + 16) = make_uint4(T.d[4], T.d[5], T.d[6], T.d[7]);
    }
    // bounds of symbol x straight from LDS (entry 16 is the constant 32768)
    __device__ __forceinline__ void bounds(const u8 *tb, u32 x, u32 &c0, u32 &c1) const
    {
        c0 = ((const u16 *)tb)[x];
        const u32 n = ((const u16 *)tb)[(x + 1u) & 15u];
        c1 = x == 15u ? TRC_PROB_ONE : n;
    }
};

// cdf16upd: every entry moves 1/128 of the way to 10*i (entries <= thr) or 10*i + 32736 (entries > thr).
// thr may be the coded symbol's lower bound c (encoders) or the decoded slot/quotient q with
// c <= q < next bound (decoders, as cdf16ansdec does): the tables are strictly increasing, so both
// select exactly the entries above the coded symbol.
__device__ __forceinline__ void trc_nib_adapt(NibTable &T, u32 thr)
{
    const trc_s2 tt = trc_as_s2(trc_pk(thr, thr));
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const trc_s2 v = trc_as_s2(T.d[k]);
        const trc_s2 gt = (tt - v) >> (trc_s2)15;                                   // -1 where v > thr
        const u32 bonus = trc_as_u32(gt) & 0x7fe07fe0u;                             // 32736 in those lanes
        trc_s2 d = trc_as_s2(trc_pk(20 * k, 20 * k + 10)) - v;
        d = (d + trc_as_s2(bonus)) >> (trc_s2)7;
        T.d[k] = trc_as_u32(v + d);
    }
}
// number of entries > q  (entry 0 is always 0, so this is 15 - x for the symbol x that contains q)
__device__ __forceinline__ u32 trc_nib_count_gt(const NibTable &T, u32 q)
{
    const trc_s2 qq = trc_as_s2(trc_pk(q, q));
    u32 acc = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) acc += (trc_as_u32(qq - trc_as_s2(T.d[k])) >> 15) & 0x00010001u;   // sign bits
    return (acc & 0xffu) + (acc >> 16);
}
