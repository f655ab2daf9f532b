// trc_nibmodel.h -- the adaptive 16-symbol CDF model ("CDF16", rate 7) of the reference, per lane, in LDS.
// Reference: init CDF16DEC0/1 cdf_.h:26-27,40-41 (cdf[j] = j<<11); update cdf16upd cdf_.h:46-50 (AVX2) ==
// :87-97 (SSE) -- the SIMD rule is normative, the scalar fallback of cdf_.h:112-117 is a different formula
// (SURVEY F7).  After coding symbol x whose lower bound was c = t[x] (read BEFORE the update), in int16 lanes
// with arithmetic shift:
//        t[i] += ((10*i - t[i]) + (t[i] > c ? 32736 : 0)) >> 7          i = 0..15
// A byte model is one "hi" table plus 16 "lo" tables selected by the hi nibble (rccdf.c:202, anscdf.c:574-575):
// 17 x 16 x u16 = 544 B per lane; the nibble coders (`turborc -n`) use a single table.  Entry 16 (= 32768) is
// never stored.
//
// gfx950 mapping.  A table is 32 B = 8 dwords of packed int16 pairs, moved with two ds_read_b128 / two
// ds_write_b128.  The tables are strictly increasing (the update keeps t[i+1] - t[i] >= 1: both targets are
// 10 apart and the step is 1/128 of the distance, floor), so "t[i] > t[x]" is "i > x" and the update is
//        t[i] += (K[x][i] - t[i]) >> 7,        K[x][i] = 10*i + (i > x ? 32736 : 0)   (mod 2^16)
// with K a 16 x 16 constant table (512 B per wave in LDS): three packed-16 VALU ops per dword
// (v_pk_sub_i16, v_pk_ashrrev_i16, v_pk_add_i16) instead of seven.  Decoders find the symbol by a binary
// search over the register copy of the table (10 v_cndmask), which also yields both bounds.
// Lane rows: 560 B apart for the byte model (140 dwords), 48 B for the single table (12 dwords): in both cases
// 16 consecutive lanes x 4 banks tile all 64 banks, so equal-offset b128 accesses are conflict free.
#pragma once
#include "trc_dev.h"

typedef short trc_s2 __attribute__((ext_vector_type(2)));

#define TRC_NIBK_BYTES  512u                      // K table, per wave
#ifndef TRC_NIB_ROW
#define TRC_NIB_ROW     560u                      // bytes per lane, byte model (17 tables)
#endif
#define TRC_NIB1_ROW    48u                       // bytes per lane, single table
#define TRC_NIB2_ROW    80u                       // bytes per lane, two tables (Turbo-VLC coders): 20 dwords, conflict free like the others
#define TRC_NIB3_ROW    112u                      // bytes per lane, three tables (vnibble coders): 28 dwords, lane*28 mod 64 hits 16 distinct bank groups
#define TRC_NIB_BYTES   (TRC_NIBK_BYTES + 64u * TRC_NIB_ROW)      // 36352 per wave
#define TRC_NIB1_BYTES  (TRC_NIBK_BYTES + 64u * TRC_NIB1_ROW)     // 3584 per wave
#define TRC_NIB2_BYTES  (TRC_NIBK_BYTES + 64u * TRC_NIB2_ROW)     // 5632 per wave
#define TRC_NIB3_BYTES  (TRC_NIBK_BYTES + 64u * TRC_NIB3_ROW)     // 7680 per wave

struct NibTable { u32 d[8]; };                    // 16 x u16, entry 2k in the low half of d[k]

__device__ __forceinline__ u32 trc_pk(u32 lo, u32 hi) { return (lo & 0xffffu) | (hi << 16); }
__device__ __forceinline__ trc_s2 trc_as_s2(u32 v) { return __builtin_bit_cast(trc_s2, v); }
__device__ __forceinline__ u32 trc_as_u32(trc_s2 v) { return __builtin_bit_cast(u32, v); }

// NT = tables per lane: 17 (byte model), 1 (nibble coders), 2 (Turbo-VLC coders), 3 (vnibble coders)
template <int NT>
struct NibModel {
    static constexpr u32 ROW = NT == 17 ? TRC_NIB_ROW : NT == 1 ? TRC_NIB1_ROW : NT == 3 ? TRC_NIB3_ROW : TRC_NIB2_ROW;
    u8 *kb;                                       // this wave's K table
    u8 *row;                                      // this lane's tables
    // smem = this wave's model area (TRC_NIB_BYTES / TRC_NIB1_BYTES); every lane of the wave must call
    __device__ __forceinline__ void init(u8 *smem)
    {
        const u32 lane = trc_lane();
        kb = smem;
        row = smem + TRC_NIBK_BYTES + lane * ROW;
#pragma unroll
        for (u32 j = 0; j < 2; j++) {             // 128 dwords of K, two per lane
            const u32 idx = lane * 2u + j, x = idx >> 3, k = idx & 7u;
            const u32 e0 = 2u * k, e1 = 2u * k + 1u;
            ((u32 *)kb)[idx] = trc_pk(10u * e0 + (e0 > x ? 32736u : 0u), 10u * e1 + (e1 > x ? 32736u : 0u));
        }
        for (u32 t = 0; t < (u32)NT; t++)
#pragma unroll
            for (u32 k = 0; k < 8; k++) ((u32 *)(row + t * 32u))[k] = trc_pk((2 * k) << 11, (2 * k + 1) << 11);
        trc_wave_lds_fence();                     // the K writes of this wave before its lanes' reads (the model belongs to ONE wave)
    }
    // a wave that only READS the rows another wave of its workgroup initialises and adapts (the two-wave decoders)
    __device__ __forceinline__ void attach(u8 *smem) { kb = smem; row = smem + TRC_NIBK_BYTES + trc_lane() * ROW; }
    // the same for a wave that owns only tables [first, first + count) of the rows (the two-wave model pass: a hi wave and a lo
    // wave share the rows; each initialises its own tables, both write the same K)
    __device__ __forceinline__ void init_part(u8 *smem, u32 first, u32 count)
    {
        const u32 lane = trc_lane();
        kb = smem;
        row = smem + TRC_NIBK_BYTES + lane * ROW;
#pragma unroll
        for (u32 j = 0; j < 2; j++) {
            const u32 idx = lane * 2u + j, x = idx >> 3, k = idx & 7u;
            const u32 e0 = 2u * k, e1 = 2u * k + 1u;
            ((u32 *)kb)[idx] = trc_pk(10u * e0 + (e0 > x ? 32736u : 0u), 10u * e1 + (e1 > x ? 32736u : 0u));
        }
        for (u32 t = first; t < first + count; t++)
#pragma unroll
            for (u32 k = 0; k < 8; k++) ((u32 *)(row + t * 32u))[k] = trc_pk((2 * k) << 11, (2 * k + 1) << 11);
        trc_wave_lds_fence();
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
    // bounds of symbol x straight from LDS (entry 16 is the constant 32768; the u16 after a table is in-row padding or
    // the next table, never out of the row)
    __device__ __forceinline__ void bounds(const u8 *tb, u32 x, u32 &c0, u32 &c1) const
    {
        c0 = ((const u16 *)tb)[x];
        const u32 n = ((const u16 *)tb)[x + 1u];
        c1 = x == 15u ? TRC_PROB_ONE : n;
    }
    // cdf16upd for coded symbol x
    __device__ __forceinline__ void adapt(NibTable &T, u32 x) const { adapt_k(T, load(kb + x * 32u)); }
    // the same with the K row of the symbol already in registers
    __device__ __forceinline__ void adapt_k(NibTable &T, const NibTable &K) const
    {
        // three passes over the eight dwords, not eight three-step chains: on gfx950 a packed shift that reads the result of
        // the packed subtract right before it costs a wait state (the compiler fills it with an s_nop)
        trc_s2 d[8];
#pragma unroll
        for (int k = 0; k < 8; k++) d[k] = trc_as_s2(K.d[k]) - trc_as_s2(T.d[k]);
#pragma unroll
        for (int k = 0; k < 8; k++) d[k] = d[k] >> (trc_s2)7;
#pragma unroll
        for (int k = 0; k < 8; k++) T.d[k] = trc_as_u32(trc_as_s2(T.d[k]) + d[k]);
    }
    // encoder side: {c0 << 15 | freq} of symbol x under table tb, then the table adapts
    __device__ __forceinline__ u32 record(u8 *tb, u32 x) const
    {
        u32 c0, c1; bounds(tb, x, c0, c1);
        NibTable T = load(tb); adapt(T, x); store(tb, T);
        return (c0 << TRC_PROB_BITS) | (c1 - c0);
    }
    // Encoder side, NB bytes at once (byte model).  An encoder knows its symbols, hence every table ADDRESS, before it
    // walks them: with one wave per SIMD nothing hides an LDS round trip, and record() per nibble puts four dependent
    // ones into every byte (table load -> adapt -> store -> the next load of the same table).  Here the K rows of all
    // 2 NB nibbles and the lo tables of all NB bytes are requested first; the hi table T0 lives in registers for the
    // whole chunk (its LDS copy is only written, for the bounds reads); a lo table that an EARLIER byte of the batch has
    // already updated is taken from that byte's registers instead (hi nibbles equal: 8 v_cndmask), so the chain from
    // byte to byte is VALU only.  LDS executes a wave's accesses in order: every bounds read sees the stores before it,
    // and the bounds are USED only after the walk, so nothing waits inside it.
    // r[2i] = hi record of byte i, r[2i + 1] = lo record.
    template <int NB>
    __device__ __forceinline__ void record_bytes(NibTable &T0, const u32 (&x)[NB], u32 (&r)[2 * NB]) const
    {
        NibTable Kh[NB], Kl[NB], L[NB], U[NB];
#pragma unroll
        for (int i = 0; i < NB; i++) {
            Kh[i] = load(kb + (x[i] >> 4) * 32u);
            Kl[i] = load(kb + (x[i] & 15u) * 32u);
            L[i] = load(table(1u + (x[i] >> 4)));
        }
        u32 c0[2 * NB], c1[2 * NB];
#pragma unroll
        for (int i = 0; i < NB; i++) {
            const u32 h = x[i] >> 4, l = x[i] & 15u;
            bounds(table(0), h, c0[2 * i], c1[2 * i]);
            adapt_k(T0, Kh[i]); store(table(0), T0);
            NibTable T = L[i];
#pragma unroll
            for (int j = 0; j < i; j++) {
                const bool same = (x[j] >> 4) == h;
#pragma unroll
                for (int k = 0; k < 8; k++) T.d[k] = same ? U[j].d[k] : T.d[k];
            }
            u8 *tb = table(1u + h);
            bounds(tb, l, c0[2 * i + 1], c1[2 * i + 1]);
            adapt_k(T, Kl[i]); store(tb, T);
            U[i] = T;
        }
#pragma unroll
        for (int i = 0; i < 2 * NB; i++) r[i] = (c0[i] << TRC_PROB_BITS) | (c1[i] - c0[i]);
    }
    // record_bytes split in two for the two-wave model pass of the adaptive rANS (trc_ans_adaptive.hip, round 4): the hi records of
    // NB bytes need the hi table only (registers; its LDS copy serves the bounds reads), the lo records the sixteen lo tables only
    // -- disjoint LDS, no data between the two, so a "hi" wave and a "lo" wave walk the same bytes independently.
    template <int NB>
    __device__ __forceinline__ void record_hi(NibTable &T0, const u32 (&x)[NB], u32 (&r)[NB]) const
    {
        NibTable Kh[NB];
#pragma unroll
        for (int i = 0; i < NB; i++) Kh[i] = load(kb + (x[i] >> 4) * 32u);
        u32 c0[NB], c1[NB];
#pragma unroll
        for (int i = 0; i < NB; i++) {
            bounds(table(0), x[i] >> 4, c0[i], c1[i]);
            adapt_k(T0, Kh[i]); store(table(0), T0);
        }
#pragma unroll
        for (int i = 0; i < NB; i++) r[i] = (c0[i] << TRC_PROB_BITS) | (c1[i] - c0[i]);
    }
    template <int NB>
    __device__ __forceinline__ void record_lo(const u32 (&x)[NB], u32 (&r)[NB]) const
    {
        NibTable Kl[NB], L[NB], U[NB];
#pragma unroll
        for (int i = 0; i < NB; i++) {
            Kl[i] = load(kb + (x[i] & 15u) * 32u);
            L[i] = load(table(1u + (x[i] >> 4)));
        }
        u32 c0[NB], c1[NB];
#pragma unroll
        for (int i = 0; i < NB; i++) {
            const u32 h = x[i] >> 4, l = x[i] & 15u;
            NibTable T = L[i];
#pragma unroll
            for (int j = 0; j < i; j++) {
                const bool same = (x[j] >> 4) == h;
#pragma unroll
                for (int k = 0; k < 8; k++) T.d[k] = same ? U[j].d[k] : T.d[k];
            }
            u8 *tb = table(1u + h);
            bounds(tb, l, c0[i], c1[i]);
            adapt_k(T, Kl[i]); store(tb, T);
            U[i] = T;
        }
#pragma unroll
        for (int i = 0; i < NB; i++) r[i] = (c0[i] << TRC_PROB_BITS) | (c1[i] - c0[i]);
    }
    // NN symbols through ONE table held in registers (nibble coders; T0's LDS copy serves the bounds reads)
    template <int NN>
    __device__ __forceinline__ void record_nibs(NibTable &T0, const u32 (&x)[NN], u32 (&r)[NN]) const
    {
        NibTable K[NN];
#pragma unroll
        for (int i = 0; i < NN; i++) K[i] = load(kb + x[i] * 32u);
        u32 c0[NN], c1[NN];
#pragma unroll
        for (int i = 0; i < NN; i++) {
            bounds(table(0), x[i], c0[i], c1[i]);
            adapt_k(T0, K[i]); store(table(0), T0);
        }
#pragma unroll
        for (int i = 0; i < NN; i++) r[i] = (c0[i] << TRC_PROB_BITS) | (c1[i] - c0[i]);
    }
    // record() for a table the encoder keeps in registers (T == the table at tb): the LDS copy is only written, for the
    // bounds reads of later symbols -- the table's own load, the dependent half of the round trip, is gone
    __device__ __forceinline__ u32 record_r(NibTable &T, u8 *tb, u32 x) const
    {
        u32 c0, c1; bounds(tb, x, c0, c1);
        adapt(T, x); store(tb, T);
        return (c0 << TRC_PROB_BITS) | (c1 - c0);
    }
    __device__ __forceinline__ u32 record_r_if(bool on, NibTable &T, u8 *tb, u32 x) const
    {
        u32 r = 0;
        if (on) r = record_r(T, tb, x);
        return r;
    }
    // the same where `on`; elsewhere the table stays as it is and the record is 0 (freq 0: never coded)
    __device__ __forceinline__ u32 record_if(bool on, u8 *tb, u32 x) const
    {
        u32 r = 0;
        if (on) r = record(tb, x);
        return r;
    }
};

// decoder side: the symbol x with t[x] <= q < t[x+1] and its bounds, by binary search over the register copy (entry 16
// = 32768 appended).  `ge(e)` answers "q >= e" for a 16-bit table entry e: the rANS decoders compare the slot itself, the
// range decoders compare code >= (range >> 15) * e -- the same predicate as floor(code / (range >> 15)) >= e without
// the division (trc_rc.h).
// Round 5, a negative result kept as a switch.  The compiler emits the twelve selects of a search as runs of three to five VOP2
// v_cndmask_b32, the encoding the rate probe prices at 16-23 cycles per instruction in runs (profiles/r05_valu_rates.txt).  The same
// search as BIT-selects under one mask per level (v_bfi_b32; BFI = true) has no such run -- and is SLOWER in every decoder that holds its
// tables in registers (`anscdf` decode 0.540 -> 0.559 ms, `rccdf` 0.62 -> 0.67, `rccdf4` 0.269 -> 0.286, `rccdf8` 0.916 -> 0.97;
// profiles/r05m_ab.txt): in these instruction streams the runs do not cost what the probe's steady state says, the four extra mask
// instructions do.  Only the order-1 decoder, whose waves wait on memory, gains (5.36 -> 5.11 ms) and uses it.
template <bool BFI = false, class GE>
__device__ __forceinline__ u32 trc_nib_search(const NibTable &T, GE ge, u32 &c0, u32 &c1)
{
    if constexpr (!BFI) {
    const bool b3 = ge(T.d[4] & 0xffffu);
    const u32 e0 = b3 ? T.d[4] : T.d[0], e1 = b3 ? T.d[5] : T.d[1], e2 = b3 ? T.d[6] : T.d[2],
              e3 = b3 ? T.d[7] : T.d[3], e4 = b3 ? TRC_PROB_ONE : T.d[4];
    const bool b2 = ge(e2 & 0xffffu);
    const u32 f0 = b2 ? e2 : e0, f1 = b2 ? e3 : e1, f2 = b2 ? e4 : e2;
    const bool b1 = ge(f1 & 0xffffu);
    const u32 g0 = b1 ? f1 : f0, g1 = b1 ? f2 : f1;
    const u32 gh = g0 >> 16;
    const bool b0 = ge(gh);
    c0 = b0 ? gh : (g0 & 0xffffu);
    c1 = b0 ? (g1 & 0xffffu) : gh;
    // the index as a carry chain (x = 2x + b: one v_addc per bit) instead of four selects and three adds
    u32 x = b3 ? 1u : 0u;
    x = x + x + (b2 ? 1u : 0u); x = x + x + (b1 ? 1u : 0u); x = x + x + (b0 ? 1u : 0u);
    return x;
    } else {
    const u32 m3 = ge(T.d[4] & 0xffffu) ? ~0u : 0u;
    const u32 e0 = trc_bfi(m3, T.d[4], T.d[0]), e1 = trc_bfi(m3, T.d[5], T.d[1]), e2 = trc_bfi(m3, T.d[6], T.d[2]),
              e3 = trc_bfi(m3, T.d[7], T.d[3]), e4 = trc_bfi(m3, TRC_PROB_ONE, T.d[4]);
    const u32 m2 = ge(e2 & 0xffffu) ? ~0u : 0u;
    const u32 f0 = trc_bfi(m2, e2, e0), f1 = trc_bfi(m2, e3, e1), f2 = trc_bfi(m2, e4, e2);
    const u32 m1 = ge(f1 & 0xffffu) ? ~0u : 0u;
    const u32 g0 = trc_bfi(m1, f1, f0), g1 = trc_bfi(m1, f2, f1);
    const u32 gh = g0 >> 16;
    const u32 m0 = ge(gh) ? ~0u : 0u;
    c0 = trc_bfi(m0, gh, g0 & 0xffffu);
    c1 = trc_bfi(m0, g1 & 0xffffu, gh);
    return (m3 & 8u) | (m2 & 4u) | (m1 & 2u) | (m0 & 1u);
    }
}
template <bool BFI = false>
__device__ __forceinline__ u32 trc_nib_find(const NibTable &T, u32 q, u32 &c0, u32 &c1)
{
    return trc_nib_search<BFI>(T, [q](u32 e) { return q >= e; }, c0, c1);
}
