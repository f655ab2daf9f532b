// trc_rc.h -- device-side 64-bit range coder core shared by the RC kernels (RCS1/RCS2/RCA/RCB).
//
// Arithmetic = reference turborc_.h with RC_SIZE=64, RC_IO=32, RC_BITS=15 (rccdf.c:36-37,
// rc_s.c:31-33): _rccdfenc_ :215, _rcenorm_ :105-109, _rccarry_ :103, rceflush :118-128,
// rcdinit :152-158, _rccdfupdate_ :219-223, _rcdnorm_ :111, rcbe_ :417-421, rcbd_ :447-452.
//
// Carry handling.  The reference writes every 32-bit word at once and, when `low` wraps, walks
// BACK through the words already written adding 1 (ripple).  Here output is append-only (words go
// through an LDS ring and leave as coalesced segments, trc_io.h), so the encoder holds back the
// last word (`cache`) and a count of 0xFFFFFFFF words behind it (`npend`): a carry turns
// cache, FF.. into cache+1, 00..; a word is released only once no later carry can reach it.  A word
// receives at most one carry in its lifetime (the value still to be added is always smaller than
// the range at the time the word was emitted), so the released stream is word-for-word what the
// reference's ripple produces.  `nwords` counts words the way the reference's pointer does
// (held-back words included): the incompressibility test (OVERFLOW, rcutil_.h:130) uses it.
#pragma once
#include "trc_io.h"
#include "trc_carry.h"

#define TRC_TOP32 ((u64)1 << 32)

// signed limit of OVERFLOW: op - out >= inlen*255/256 - 8   (rcutil_.h:130)
__device__ __forceinline__ int trc_rc_limit(u32 len) { return (int)((len * 255u) >> 8) - 8; }

struct RcEnc {
    static constexpr u32 WBYTES = 4;                                    // bytes per emitted word
    u64 range, low, mark;
    TrcCarry cw;                                                        // held-back words (trc_carry.h)

    __device__ __forceinline__ void start() { range = ~(u64)0; low = mark = 0; cw.start(); }

    // logical append of word W with carry flag cy (cy refers to the words BEFORE W)
    template <class SO>
    __device__ __forceinline__ void emit(SO &so, bool cy, u32 W) { cw.emit(so, cy, W); }
    template <class SO>
    __device__ __forceinline__ void renorm(SO &so)                      // single `if`: RC_IO = 32
    {
        if (range < TRC_TOP32) {
            emit(so, mark > low, (u32)(low >> 32));
            low <<= 32; range <<= 32; mark = low;
        }
    }
    template <class SO>
    __device__ __forceinline__ void sym(SO &so, u32 c0, u32 f)          // _rccdfenc_ + renorm
    {
        sym_if(so, true, c0, f);                                        // the common renorm without a branch (rccdfs: 192 -> 181 us)
    }
    // predicated, branch-free in the common case: where !act nothing changes (model-bound coders, trc_rc_adaptive.hip)
    template <class SO>
    __device__ __forceinline__ void sym_if(SO &so, bool act, u32 c0, u32 f)
    {
        const u64 r = range >> TRC_PROB_BITS;
        const u64 low2 = low + r * (act ? c0 : 0u);
        const u64 range2 = r * f;
        const bool rn = act && range2 < TRC_TOP32;
        cw.emit_if(so, rn, mark > low2, (u32)(low2 >> 32));
        low = rn ? low2 << 32 : low2;
        range = act ? (rn ? range2 << 32 : range2) : range;
        mark = rn ? low : mark;
    }
    // rceflush, then release everything still held back
    template <class SO>
    __device__ __forceinline__ void finish(SO &so)
    {
        renorm(so);
        if (range > ((u64)1 << 33)) {
            low += TRC_TOP32;
            emit(so, mark > low, (u32)(low >> 32));
        } else {
            low += 1;
            emit(so, mark > low, (u32)(low >> 32));
            emit(so, false, (u32)low);
        }
        cw.release(so);
    }
};

// The same encoder for the model-bound kernels (one wave per SIMD: their time is their instruction count).  Differences:
//  * a carry is the carry-out of `low +=` ORed into a flag, not a compare against the value of `low` at the last
//    renormalisation (same event: between two renormalisations `low` grows by less than the range it had at the first);
//  * a renormalisation only SHIFTS the state and remembers (word, carry); the emit logic runs in `flush`, which the caller
//    invokes after every SECOND symbol: two consecutive symbols cannot both renormalise (after one, range >= 2^17 << 32 =
//    2^49, and one more symbol leaves at least (2^49 >> 15) * 1 = 2^34 >= 2^32), so one remembered word is enough.
struct RcEncD {
    static constexpr u32 WBYTES = 4;
    u64 range, low;
    bool cy, pend, pcy;
    u32 pw;
    TrcCarry cw;
    __device__ __forceinline__ void start() { range = ~(u64)0; low = 0; cy = pend = pcy = false; pw = 0; cw.start(); }
    __device__ __forceinline__ void sym_rec(bool act, u32 c0, u32 f)    // _rccdfenc_ + renorm where act, nothing where !act
    {
        const u64 r = range >> TRC_PROB_BITS;
        const u64 low2 = low + r * (act ? c0 : 0u);
        const bool cyn = cy || low2 < low;
        const u64 range2 = r * f;
        const bool rn = act && range2 < TRC_TOP32;
        pw = rn ? (u32)(low2 >> 32) : pw;
        pcy = rn ? cyn : pcy;
        pend = pend || rn;
        cy = cyn && !rn;
        low = rn ? low2 << 32 : low2;
        range = act ? (rn ? range2 << 32 : range2) : range;
    }
    template <class SO>
    __device__ __forceinline__ void flush(SO &so)
    {
        cw.emit_if(so, pend, pcy, pw);
        pend = pcy = false; pw = 0;                                     // (constants at the start of the next pair: nothing of them is carried around the loops)
    }
    template <class SO>
    __device__ __forceinline__ void sym(SO &so, u32 c0, u32 f) { sym_rec(true, c0, f); flush(so); }   // one symbol, emitted at once
    template <class SO>
    __device__ __forceinline__ void finish(SO &so)                      // rceflush, then release everything still held back
    {
        flush(so);
        bool c0 = cy;
        if (range < TRC_TOP32) { cw.emit(so, c0, (u32)(low >> 32)); low <<= 32; range <<= 32; c0 = false; }
        if (range > ((u64)1 << 33)) {
            const u64 nl = low + TRC_TOP32;
            cw.emit(so, c0 || nl < low, (u32)(nl >> 32));
        } else {
            const u64 nl = low + 1;
            cw.emit(so, c0 || nl < low, (u32)(nl >> 32));
            cw.emit(so, false, (u32)nl);
        }
        cw.release(so);
    }
};

// RcEncD once more for the CODER WAVES of round 4 (trc_rca_enc_mc_kernel): the same arithmetic with its books on the vector side.
// With one or two other waves on the SIMD a coder wave's time is still made of its dependency chains, and every Boolean that the
// compiler keeps as a lane mask (carry flag, "renormalised", "a word is pending", the predicate "this lane is coding") is an SGPR
// operation between a vector compare and the selects that wait for it (profiles/r04_notes.md, sections 8 and 12).  Here the state is
// four 32-bit halves plus a carry limb `lx` fed by the add's carry-out, the pending word's flags are integers, and `sym<false>` carries
// no predicate at all: a lane that is not coding runs along on its own registers, and the caller keeps it from emitting.
struct RcEncV {
    static constexpr u32 WBYTES = 4;     // bytes per emitted word
    u32 rlo, rhi, llo, lhi, lx;          // range, low, carry limb of low (0 / 1)
    u32 pend, pcy, pw;                   // a word waits for flush(): its carry flag and itself
    TrcCarry cw;
    __device__ __forceinline__ void start() { rlo = rhi = ~0u; llo = lhi = lx = 0; pend = pcy = pw = 0; cw.start(); }
    template <bool PRED>
    __device__ __forceinline__ void sym(bool act, u32 c0, u32 f)        // _rccdfenc_ + renorm; PRED: nothing where !act
    {
        const u32 slo = __builtin_amdgcn_alignbit(rhi, rlo, TRC_PROB_BITS), shi = rhi >> TRC_PROB_BITS;      // r = range >> 15 (49 bits)
        const u32 cc = PRED ? (act ? c0 : 0u) : c0;
        const u64 p64 = (u64)slo * cc;                                   // low += r * c0 (< 2^64)
        const u32 plo = (u32)p64, phi = __umul24(shi, cc) + (u32)(p64 >> 32);
        u32 k1, k2, k3;
        llo = __builtin_addc(llo, plo, 0u, &k1);
        lhi = __builtin_addc(lhi, phi, k1, &k2);
        lx = __builtin_addc(lx, 0u, k2, &k3);
        const u64 q64 = (u64)slo * f;                                    // range = r * freq
        const u32 qlo = (u32)q64, qhi = __umul24(shi, f) + (u32)(q64 >> 32);
        const bool rn = PRED ? (act && qhi == 0u) : qhi == 0u;           // range < 2^32: the top word of low goes out
        pw = rn ? lhi : pw; pcy = rn ? lx : pcy; pend = rn ? 1u : pend;
        lx = rn ? 0u : lx; lhi = rn ? llo : lhi; llo = rn ? 0u : llo;
        const u32 nrh = rn ? qlo : qhi, nrl = rn ? 0u : qlo;
        rhi = PRED ? (act ? nrh : rhi) : nrh; rlo = PRED ? (act ? nrl : rlo) : nrl;
    }
    // after every second symbol (two consecutive symbols cannot both renormalise, see RcEncD): the pending word, where `on`
    template <class SO>
    __device__ __forceinline__ void flush(SO &so, bool on)
    {
        cw.emit_if(so, pend != 0u && on, pcy != 0u, pw);
        pend = pcy = 0;
    }
    template <class SO>
    __device__ __forceinline__ void sym(SO &so, u32 c0, u32 f) { sym<true>(true, c0, f); flush(so, true); }    // one symbol, emitted at once
    // RcEncD's names, for the static coders (every lane of a full piece codes: no predicate)
    __device__ __forceinline__ void sym_rec(bool, u32 c0, u32 f) { sym<false>(true, c0, f); }
    template <class SO>
    __device__ __forceinline__ void flush(SO &so) { flush(so, true); }
    template <class SO>
    __device__ __forceinline__ void finish(SO &so)                      // rceflush, then release everything still held back
    {
        u64 low = ((u64)lhi << 32) | llo, range = ((u64)rhi << 32) | rlo;
        bool c0 = lx != 0u;
        if (range < TRC_TOP32) { cw.emit(so, c0, (u32)(low >> 32)); low <<= 32; range <<= 32; c0 = false; }
        if (range > ((u64)1 << 33)) {
            const u64 nl = low + TRC_TOP32;
            cw.emit(so, c0 || nl < low, (u32)(nl >> 32));
        } else {
            const u64 nl = low + 1;
            cw.emit(so, c0 || nl < low, (u32)(nl >> 32));
            cw.emit(so, false, (u32)nl);
        }
        cw.release(so);
    }
};

struct RcDec {
    u64 range, code;
    __device__ __forceinline__ void start(u32 w0, u32 w1) { range = ~(u64)0; code = ((u64)w0 << 32) | w1; }
    template <class SI>
    __device__ __forceinline__ void init(SI &si)                        // rcdinit: two 32-bit words
    {
        const u32 a = si.peek32(); si.rpos += 4; const u32 b = si.peek32(); si.rpos += 4;
        start(a, b);
    }
    // t = code / r for r = range >> 15 (the caller has NOT shifted range), branch-free: the f32 estimate is within +-1 of the
    // exact quotient, so one correction step each way is exact for every valid stream.  Both operands go to f32 as
    // hi * 2^32 + lo (two conversions and an fma: relative error <= 2^-23 whatever the magnitude; rounds 1-3 normalised them
    // with a count-leading-zeros and two 64-bit shifts first, eight instructions more); v_rcp_f32 is 1 ulp, the product 2^-24:
    // relative 2^-21 at most, times t < 2^15 => < 0.02 absolute.
    __device__ __forceinline__ u32 quotient15() const
    {
        // (everything on explicit 32-bit halves: from `(float)(u32)(r >> 32)` the compiler builds a 64-bit integer conversion, normalising
        // shift and all; from `r * t` with an unbounded t two full 64-bit multiply-adds where a 24-bit one does for the high half)
        const u32 rh = (u32)(range >> 32), ch = (u32)(code >> 32);
        const u32 slo = __builtin_amdgcn_alignbit(rh, (u32)range, TRC_PROB_BITS), shi = rh >> TRC_PROB_BITS;     // r = range >> 15: 2^17 <= r < 2^49
        const u64 r = ((u64)shi << 32) | slo;
        const float rf = __builtin_fmaf(trc_u2f(shi), 4294967296.0f, trc_u2f(slo));
        const float cf = __builtin_fmaf(trc_u2f(ch), 4294967296.0f, trc_u2f((u32)code));
        u32 t = (u32)(cf * __builtin_amdgcn_rcpf(rf));             // (a corrupt stream's estimate can be anything: the clamp behind the correction keeps it in the table)
        const u64 plo = (u64)slo * t;
        const u64 p = ((u64)((u32)(plo >> 32) + __umul24(shi, t)) << 32) | (u32)plo;      // r * t for t < 2^24
        const bool dn = p > code, up = !dn && code - p >= r;
        t = dn ? t - 1u : up ? t + 1u : t;
        return t > TRC_PROB_ONE - 1 ? TRC_PROB_ONE - 1 : t;              // corrupt input: stay inside the table
    }
    // Round 5: the static decoders do not need the exact quotient either -- only the SYMBOL it falls into, and whether a symbol is the
    // right one can be read off the two products the state update needs anyway: x is the symbol iff r c0 <= code < r c1.  So: look the
    // symbol up from the f32 ESTIMATE (within +-1 of the quotient: wrong symbol only when the estimate is off AND the true slot is a
    // symbol's first or last), form rp = r c0, range2 = r (c1 - c0), code2 = code - rp, and accept if code >= rp and code2 < range2;
    // where some lane of the wave fails (a few percent of the wave-steps) the wave repeats the step with the exact quotient.  The
    // correction of quotient15 -- r * t on halves, two 64-bit compares, a 64-bit subtraction, three selects: a quarter of the step --
    // is off the common path (static range decoders: profiles/r05_notes.md).
    __device__ __forceinline__ u32 estimate15() const
    {
        const u32 rh = (u32)(range >> 32), ch = (u32)(code >> 32);
        const u32 slo = __builtin_amdgcn_alignbit(rh, (u32)range, TRC_PROB_BITS), shi = rh >> TRC_PROB_BITS;
        const float rf = __builtin_fmaf(trc_u2f(shi), 4294967296.0f, trc_u2f(slo));
        const float cf = __builtin_fmaf(trc_u2f(ch), 4294967296.0f, trc_u2f((u32)code));
        const u32 t = (u32)(cf * __builtin_amdgcn_rcpf(rf));
        return t > TRC_PROB_ONE - 1 ? TRC_PROB_ONE - 1 : t;
    }
    struct Probe { u64 range2, code2; bool fits; };
    __device__ __forceinline__ Probe probe(u32 c0, u32 c1) const
    {
        const u64 r = range >> TRC_PROB_BITS;
        const u64 rp = r * c0;
        Probe q;
        q.range2 = r * (u32)(c1 - c0); q.code2 = code - rp;
        q.fits = code >= rp && q.code2 < q.range2;
        return q;
    }
    __device__ __forceinline__ bool commit(const Probe &q, u32 w)       // consume_w's second half
    {
        const bool rn = q.range2 < TRC_TOP32;
        range = rn ? q.range2 << 32 : q.range2;
        code = rn ? (q.code2 << 32) | w : q.code2;
        return rn;
    }
    // The adaptive decoders do not need the quotient itself, only the table entry it falls behind: with r = range >> 15,
    // floor(code / r) >= e  <=>  code >= r * e, so the reference's search over code / r (cdflget16, turborc_.h:172-190) is a
    // binary search with four 49 x 16-bit products instead of a division and four compares (r * e < 2^64: e <= 2^15).
    // A corrupt stream (code / r >= 2^15) ends at the last symbol, as the reference's clamped quotient does.
    struct GeScaled {
        u64 r, code;
        __device__ __forceinline__ bool operator()(u32 e) const { return code >= r * e; }
    };
    __device__ __forceinline__ GeScaled scaled() const { return GeScaled{ range >> TRC_PROB_BITS, code }; }
    // the same step with the stream word handed in and the "renormalised" flag handed back: two consecutive symbols cannot
    // both renormalise (see RcEncD), so a caller decodes a PAIR against one look-ahead word and advances its stream once
    __device__ __forceinline__ bool consume_w(bool act, u32 c0, u32 c1, u32 w)
    {
        const u64 r = range >> TRC_PROB_BITS;
        const u64 rp = r * c0;
        const u64 range2 = r * (u32)(c1 - c0), code2 = code - rp;
        const bool rn = act && range2 < TRC_TOP32;
        range = act ? (rn ? range2 << 32 : range2) : range;
        code = act ? (rn ? (code2 << 32) | w : code2) : code;
        return rn;
    }
    // _rccdfupdate + renorm where act, nothing where !act (range is still the unshifted one)
    template <class SI>
    __device__ __forceinline__ void consume_if(SI &si, bool act, u32 c0, u32 c1)
    {
        const u64 r = range >> TRC_PROB_BITS;
        const u64 rp = r * c0;
        const u64 range2 = r * (u32)(c1 - c0), code2 = code - rp;           // (one 49 x 16-bit product: r*c1 - r*c0 with the difference taken first)
        const bool rn = act && range2 < TRC_TOP32;
        const u32 w = si.peek32();
        range = act ? (rn ? range2 << 32 : range2) : range;
        code = act ? (rn ? (code2 << 32) | w : code2) : code;
        si.skip_if(rn);
    }
};

// ------------------------------------------------------------------------------------------------------------
// The 32-bit geometry of `turborc -e44` (rccdfsm*, rccdf.c:648-694: RC_SIZE 32, RC_IO 16, RC_BITS 15): 32-bit
// range/low, 16-bit words, renorm below 2^16 (a single step: range >= 2 after the scale), flush adds 2^16 and
// emits one word when range > 2^17, else adds 1 and emits two (turborc_.h:118-128 with the 32-bit types).
// Same append-only carry scheme with 16-bit words.
template <class SO>
struct TrcSink16 {                                                      // the carry logic's sink interface on 16-bit words
    SO &so;
    __device__ __forceinline__ void put32(u32 v) { so.put16(v); }
    __device__ __forceinline__ void put32_slow(u32 v) { so.put16_slow(v); }
    __device__ __forceinline__ void put32_if(bool take, u32 v) { so.put16_if(take, v); }
};
struct RcEncSm {
    static constexpr u32 WBYTES = 2;
    u32 range, low, mark;
    TrcCarryT<0xffffu> cw;
    __device__ __forceinline__ void start() { range = ~0u; low = mark = 0; cw.start(); }
    template <class SO>
    __device__ __forceinline__ void renorm(SO &so)
    {
        if (range < (1u << 16)) {
            TrcSink16<SO> k{so};
            cw.emit(k, mark > low, low >> 16);
            low <<= 16; range <<= 16; mark = low;
        }
    }
    template <class SO>
    __device__ __forceinline__ void sym(SO &so, u32 c0, u32 f)
    {
        range >>= TRC_PROB_BITS;
        low += range * c0;
        range *= f;
        renorm(so);
    }
    template <class SO>
    __device__ __forceinline__ void finish(SO &so)
    {
        TrcSink16<SO> k{so};
        renorm(so);
        if (range > (1u << 17)) {
            low += 1u << 16;
            cw.emit(k, mark > low, low >> 16);
        } else {
            low += 1;
            cw.emit(k, mark > low, low >> 16);
            cw.emit(k, false, low & 0xffffu);
        }
        cw.release(k);
    }
};
struct RcDecSm {
    u32 range, code;
    template <class SI>
    __device__ __forceinline__ void init(SI &si)                        // rcdinit: two 16-bit words, first one on top
    {
        const u32 a = si.peek32(); si.rpos += 4;
        range = ~0u; code = (a << 16) | (a >> 16);
    }
    // scale, then code/range exactly (the reference goes through a reciprocal table, turborc_.h:172-190).  The
    // quotient can exceed 32767 because range>>15 truncates; every reference search then yields the last symbol.
    __device__ __forceinline__ u32 slot()
    {
        range >>= TRC_PROB_BITS;                                         // 2 <= range < 2^17
        u32 t = (u32)((float)code * __builtin_amdgcn_rcpf((float)range));   // < 49152, within +-1
        u64 p = (u64)t * range;
        if (p > code) { t--; p -= range; }
        if ((u64)code - p >= range) t++;
        return t > TRC_PROB_ONE - 1 ? TRC_PROB_ONE - 1 : t;
    }
    // round 5, as RcDec::estimate15 / probe: the symbol from the ESTIMATED quotient, accepted where code lies inside its bounds (the
    // two products the update needs anyway); slot_exact() only where some lane of the wave fails -- which includes every step whose
    // true quotient is 2^15 or more (range >> 15 truncates): no symbol's bounds hold such a code, the exact path clamps.
    __device__ __forceinline__ void scale() { range >>= TRC_PROB_BITS; }
    __device__ __forceinline__ u32 slot_estimate() const
    {
        const u32 t = (u32)((float)code * __builtin_amdgcn_rcpf((float)range));
        return t > TRC_PROB_ONE - 1 ? TRC_PROB_ONE - 1 : t;
    }
    __device__ __forceinline__ u32 slot_exact() const                   // (range already scaled)
    {
        u32 t = (u32)((float)code * __builtin_amdgcn_rcpf((float)range));
        u64 p = (u64)t * range;
        if (p > code) { t--; p -= range; }
        if ((u64)code - p >= range) t++;
        return t > TRC_PROB_ONE - 1 ? TRC_PROB_ONE - 1 : t;
    }
    __device__ __forceinline__ bool fits(u32 c0, u32 c1) const
    {
        const u64 rp = (u64)range * c0, top = (u64)range * c1;           // (range < 2^17, c <= 2^15: the products may reach 2^32)
        return code >= rp && code < top;
    }
    template <class SI>
    __device__ __forceinline__ void consume(SI &si, u32 c0, u32 c1)
    {
        const u32 rp = range * c0;
        range = range * c1 - rp;
        code -= rp;
        const bool rn = range < (1u << 16);
        const u32 w = si.peek16();
        if (rn) { range <<= 16; code = (code << 16) | w; }
        si.rpos += rn ? 2u : 0u;
    }
};
