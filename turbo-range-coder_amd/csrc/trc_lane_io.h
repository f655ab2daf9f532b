// trc_lane_io.h -- per-lane register-window stream I/O for the MODEL-BOUND coders (adaptive CDF16 coders).
//
// The static coders move their variable-rate streams through per-lane LDS rings and leave the CU in whole
// 64-byte segments (trc_io.h), because they run at more than 1 TB/s and every scattered 16-byte lane access
// costs ~5 TA cycles.  The adaptive coders are two orders of magnitude below that rate (their time goes into
// the 16-entry table update of every nibble) and their occupancy is set by LDS: the byte model is 544 B per
// lane, 35 KiB per wave.  For them a stream is a 16-byte register window per lane:
//   LaneOut32  appends 32-bit words; every fourth word the lane stores its 16 bytes itself;
//   LaneIn     keeps the current and the next 16 bytes of the lane's stream in registers; the next window is
//              requested when the current one is used up, i.e. 16 bytes (>= 4 renormalisations) ahead.
// No LDS at all, so four model-carrying waves fit a CU (one per SIMD).  At <= 1 scattered access per ~40
// coded nibbles the TA cost is noise (profiles/r01_notes.md).
// Late round 2 (profiles/r02_notes.md): LaneOutDirect (the range encoders: a plain 4-byte store per released word), the
// prefetch form of the input window (LaneIn::prefetch / advance_pre) and LaneInWide (rANS decoders) -- see each below.
#pragma once
#include "trc_dev.h"

typedef u32 trc_u32x4 __attribute__((ext_vector_type(4)));
typedef trc_u32x4 trc_u32x4_a4 __attribute__((aligned(4)));          // 16-byte store that only needs dword alignment

struct LaneOut32 {
    u8 *dst;             // this lane's region (4-byte aligned)
    u32 wpos;            // bytes appended
    u32 h0, h1, h2;      // the last three words, h2 newest (what is not stored yet lives here)

    __device__ __forceinline__ void start(u8 *d) { dst = d; wpos = 0; h0 = h1 = h2 = 0; }
    __device__ __forceinline__ void put32_if(bool take, u32 v)
    {
        if (take && (wpos & 12u) == 12u) {
            const trc_u32x4 q = { h0, h1, h2, v };
            *(trc_u32x4_a4 *)(dst + wpos - 12u) = q;
        }
        // bit-select form (v_bfi_b32): a ?: on the members can become a select of their ADDRESSES, which pins the
        // whole struct in scratch memory
        const u32 m = take ? 0xffffffffu : 0u;
        h0 = (h1 & m) | (h0 & ~m); h1 = (h2 & m) | (h1 & ~m); h2 = (v & m) | (h2 & ~m);
        wpos += 4u & m;
    }
    __device__ __forceinline__ void put32(u32 v) { put32_if(true, v); }
    __device__ __forceinline__ void put32_slow(u32 v) { put32_if(true, v); }
    // store the words still held (0..3)
    __device__ __forceinline__ void finish(bool ok)
    {
        const u32 cnt = (wpos >> 2) & 3u;
        if (ok && cnt >= 3u) *(u32 *)(dst + wpos - 12u) = h0;
        if (ok && cnt >= 2u) *(u32 *)(dst + wpos - 8u) = h1;
        if (ok && cnt >= 1u) *(u32 *)(dst + wpos - 4u) = h2;
    }
};

// The same sink without a window: every word is stored where it belongs as it is released (4-byte scattered stores,
// merged into lines by L2).  For the bitwise coder, whose time is its instruction count: a lane releases a word every
// ~6 input bytes, a store per wave and byte is ~2 % of the TA's time, and the window's seven selects per byte are gone.
struct LaneOutDirect {
    u8 *dst;             // this lane's region (4-byte aligned)
    u32 wpos;            // bytes appended
    __device__ __forceinline__ void start(u8 *d) { dst = d; wpos = 0; }
    __device__ __forceinline__ void put32_if(bool take, u32 v)
    {
        if (take) { *(u32 *)(dst + wpos) = v; wpos += 4u; }
    }
    __device__ __forceinline__ void put32(u32 v) { *(u32 *)(dst + wpos) = v; wpos += 4u; }
    __device__ __forceinline__ void put32_slow(u32 v) { put32(v); }
    __device__ __forceinline__ void finish(bool) {}
};

// UNIT = bytes per consumed unit (4: range coders, 2: rANS)
template <int UNIT>
struct LaneIn {
    const u8 *src;       // this lane's stream (2-byte aligned; the payload buffer carries TRC_PAD bytes of slack)
    u32 rpos;            // bytes consumed
    u32 lim;             // no window is fetched from beyond this stream offset (a valid stream never gets there; a corrupt
                         // one re-reads the last window instead of running off the payload buffer)
    uint4 cur, nxt;      // stream bytes [16*(rpos/16), +16) and the 16 after them

    __device__ __forceinline__ void prime(const u8 *s, bool alive, u32 limit)
    {
        src = s; rpos = 0; lim = limit;
        cur = nxt = make_uint4(0, 0, 0, 0);
        if (alive) { cur = trc_ld16_a2(src); nxt = trc_ld16_a2(src + 16); }
    }
    // (bit selects under sign masks of the position's bits, not compares and v_cndmask: in these one-wave-per-SIMD kernels the
    // all-VALU form is the faster one, and a group of ?: on one condition can come out as a divergent branch)
    __device__ __forceinline__ u32 word() const
    {
        const u32 m4 = (u32)__builtin_amdgcn_sbfe((int)rpos, 2, 1), m8 = (u32)__builtin_amdgcn_sbfe((int)rpos, 3, 1);
        return trc_bfi(m8, trc_bfi(m4, cur.w, cur.z), trc_bfi(m4, cur.y, cur.x));
    }
    __device__ __forceinline__ u32 peek32() const { return word(); }
    __device__ __forceinline__ u32 peek16() const { const u32 w = word(); return (rpos & 2u) ? w >> 16 : w & 0xffffu; }
    // the words at rpos (a) and rpos + 4 (b), for callers that keep their own look-ahead (UNIT == 4: rpos is a multiple of 4)
    __device__ __forceinline__ void two_words(u32 &a, u32 &b) const
    {
        const u32 m4 = (u32)__builtin_amdgcn_sbfe((int)rpos, 2, 1), m8 = (u32)__builtin_amdgcn_sbfe((int)rpos, 3, 1);
        a = trc_bfi(m8, trc_bfi(m4, cur.w, cur.z), trc_bfi(m4, cur.y, cur.x));
        b = trc_bfi(m8, trc_bfi(m4, nxt.x, cur.w), trc_bfi(m4, cur.z, cur.y));
    }
    // Window refill without a stall.  `advance` below loads the next window inside a divergent branch, and the compiler waits
    // for that load on the spot (it lands in a temporary that must be copied into the loop-carried registers): with 64 lanes
    // some lane crosses a 16-byte boundary at almost every step, so the wave ate a memory round trip per step.  The
    // prefetch form: EVERY lane requests the 16 bytes behind its current window at the START of a step (`prefetch`, mostly
    // L1 hits: the same block ~20 times) and takes them in at the END (`advance_pre`), a whole step of work later.  Valid
    // while a step consumes at most 8 bytes: after a crossing the position is in the first half of the new window, so the
    // window behind it is not touched before the next step's prefetch has landed.
    __device__ __forceinline__ uint4 prefetch() const { return trc_ld16_a2(src + trc_min((rpos & ~15u) + 16u, lim)); }
    __device__ __forceinline__ void advance_pre(u32 bytes, const uint4 pre)
    {
        nxt = pre;
        const u32 before = rpos;
        rpos += bytes;
        const u32 mc = (u32)__builtin_amdgcn_sbfe((int)(before ^ rpos), 4, 1);        // crossed a 16-byte boundary (<= 8 bytes per step: bit 4 flips)
        cur.x = trc_bfi(mc, nxt.x, cur.x); cur.y = trc_bfi(mc, nxt.y, cur.y); cur.z = trc_bfi(mc, nxt.z, cur.z); cur.w = trc_bfi(mc, nxt.w, cur.w);
    }
    // consume `bytes` (0, 4 or 8) at once
    __device__ __forceinline__ void advance(u32 bytes)
    {
        const u32 before = rpos;
        rpos += bytes;
        if ((before ^ rpos) & ~15u) { cur = nxt; nxt = trc_ld16_a2(src + trc_min((rpos & ~15u) + 16u, lim)); }
    }
    __device__ __forceinline__ void skip_if(bool take)
    {
        rpos += take ? (u32)UNIT : 0u;
        if (take && (rpos & 15u) == 0u) { cur = nxt; nxt = trc_ld16_a2(src + trc_min(rpos + 16u, lim)); }
    }
};

// The same for 16-bit units consumed at several points of a step (the rANS decoders: up to four renormalisations per group
// of symbols, 8 bytes): a 32-byte window [base, base + 32) in registers, reads anywhere in its first 24 bytes, no refill inside
// a step.  A step: `pre = prefetch()` (the 16 bytes behind the window, every lane), any number of peek16 / skip_if totalling
// <= 8 bytes, `end_step(pre)` (the window moves up by 16 bytes where the position has left its first half: selects only).
struct LaneInWide {
    const u8 *src;       // this lane's stream (2-byte aligned; the payload buffer carries TRC_PAD bytes of slack)
    u32 rpos;            // bytes consumed
    u32 base;            // stream offset of the window (multiple of 16)
    u32 lim;             // no bytes are fetched from beyond this stream offset (a corrupt stream re-reads its last window)
    uint4 lo, hi;        // stream bytes [base, +16) and [base + 16, +16)

    __device__ __forceinline__ void prime(const u8 *s, bool alive, u32 limit)
    {
        src = s; rpos = 0; base = 0; lim = limit;
        lo = hi = make_uint4(0, 0, 0, 0);
        if (alive) { lo = trc_ld16_a2(src); hi = trc_ld16_a2(src + trc_min(16u, lim)); }
    }
    __device__ __forceinline__ uint4 prefetch() const { return trc_ld16_a2(src + trc_min(base + 32u, lim)); }
    __device__ __forceinline__ u32 peek16() const
    {
        const u32 o = rpos - base;                                       // 0 .. 22
        // bit selects under sign masks of the offset's bits (as ?: the compiler made a divergent if / else of this)
        const u32 m2 = (u32)__builtin_amdgcn_sbfe((int)o, 2, 1), m3 = (u32)__builtin_amdgcn_sbfe((int)o, 3, 1), m4 = (u32)__builtin_amdgcn_sbfe((int)o, 4, 1);
        const u32 a0 = trc_bfi(m2, lo.y, lo.x), a1 = trc_bfi(m2, lo.w, lo.z), a2 = trc_bfi(m2, hi.y, hi.x);     // (offsets >= 24 are never reached)
        const u32 w = trc_bfi(m4, a2, trc_bfi(m3, a1, a0));
        return (w >> ((o & 2u) << 3)) & 0xffffu;
    }
    __device__ __forceinline__ void skip_if(bool take) { rpos += take ? 2u : 0u; }
    __device__ __forceinline__ void end_step(const uint4 pre)
    {
        const bool up = rpos - base >= 16u;
        lo.x = up ? hi.x : lo.x; lo.y = up ? hi.y : lo.y; lo.z = up ? hi.z : lo.z; lo.w = up ? hi.w : lo.w;
        hi.x = up ? pre.x : hi.x; hi.y = up ? pre.y : hi.y; hi.z = up ? pre.z : hi.z; hi.w = up ? pre.w : hi.w;
        base += up ? 16u : 0u;
    }
};

// Round 4, the rANS byte decoders (four renormalisation points per group of two bytes, in stream order): the next FOUR 16-bit units
// of the lane's stream ride in two registers (l0 = units 0-1, l1 = units 2-3).  A point takes the low half of l0 and moves the rest
// down where it fires -- mask and count come off the SIGN of `state - 2^15` (a state is below 2^31: v_subrev, v_ashrrev 31, v_sub; round 4 took
// them off a carry chain, v_subrev_co / v_subb / v_addc back to back -- three instructions as well, but a VALU read of VCC straight behind
// the VALU write of it, where the compiler keeps two wait states on gfx950: scripts/check_isa_hazards.py), the moves are bit-selects, and a point only
// moves what the points behind it can still reach (point 3 moves nothing).  One 16-byte load per group, from the stream position at
// the START of the group (units 0-7): the group consumes cnt <= 4 units, so the next look-ahead is units cnt .. cnt + 3 of it, taken
// out at the end of the group with a three-level bit-select.  PRECONDITION of the sign form: states below 2^31, which every state of a
// valid stream is (anscdf_.h:33-44: [2^15, 2^31)); a forged stream may load an initial state >= 2^31, which this form reads as "below
// 2^15" where the borrow form read it as "not below": the (garbage) output of a corrupt stream then differs from round 4's and from a
// host decoder's -- reads stay bounded (cnt <= 4 per group, `lim`), so it is a matter of defined garbage, not of safety.  LaneInWide's peek (a three-level select of a 32-byte register window
// per POINT) and window move were ~37 instructions per byte of the decoder's ~165; this is ~22.
struct LaneLook16 {
    const u8 *src;       // this lane's stream (2-byte aligned; the payload buffer carries TRC_PAD bytes of slack)
    u32 rpos;            // bytes consumed
    u32 lim;             // no bytes are fetched from beyond this stream offset (a corrupt stream re-reads its end)
    u32 l0, l1;          // the four units at rpos

    __device__ __forceinline__ void prime(const u8 *s, u32 limit)
    {
        src = s; rpos = 0; lim = limit;
        const uint4 q = trc_ld16_a2(src);
        l0 = q.x; l1 = q.y;
    }
    __device__ __forceinline__ uint4 fetch() const { return trc_ld16_a2(src + trc_min(rpos, lim)); }
    // ecdnorm (anscdf_.h:51-73) at point J of the group: if (st < 2^15) st = st << 16 | next unit
    template <int J>
    __device__ __forceinline__ void renorm(u32 &st, u32 &cnt)
    {
        u32 m, t;
        const u32 sel = 0x05040100u;                            // v_perm: { st.b1, st.b0, l0.b1, l0.b0 } = st << 16 | unit
        if (J == 0)
            asm("v_subrev_u32_e32 %1, 0x8000, %2\n\t"
                "v_ashrrev_i32_e32 %0, 31, %1\n\t"
                "v_sub_u32_e32 %3, %3, %0\n\t"
                "v_perm_b32 %1, %2, %4, %6\n\t"
                "v_bfi_b32 %2, %0, %1, %2\n\t"
                "v_alignbit_b32 %1, %5, %4, 16\n\t"
                "v_bfi_b32 %4, %0, %1, %4\n\t"
                "v_lshrrev_b32_e32 %1, 16, %5\n\t"
                "v_bfi_b32 %5, %0, %1, %5"
                : "=&v"(m), "=&v"(t), "+v"(st), "+v"(cnt), "+v"(l0), "+v"(l1) : "s"(sel));
        else if (J == 1)
            asm("v_subrev_u32_e32 %1, 0x8000, %2\n\t"
                "v_ashrrev_i32_e32 %0, 31, %1\n\t"
                "v_sub_u32_e32 %3, %3, %0\n\t"
                "v_perm_b32 %1, %2, %4, %6\n\t"
                "v_bfi_b32 %2, %0, %1, %2\n\t"
                "v_alignbit_b32 %1, %5, %4, 16\n\t"
                "v_bfi_b32 %4, %0, %1, %4"
                : "=&v"(m), "=&v"(t), "+v"(st), "+v"(cnt), "+v"(l0) : "v"(l1), "s"(sel));
        else if (J == 2)
            asm("v_subrev_u32_e32 %1, 0x8000, %2\n\t"
                "v_ashrrev_i32_e32 %0, 31, %1\n\t"
                "v_sub_u32_e32 %3, %3, %0\n\t"
                "v_perm_b32 %1, %2, %4, %5\n\t"
                "v_bfi_b32 %2, %0, %1, %2\n\t"
                "v_lshrrev_b32_e32 %1, 16, %4\n\t"
                "v_bfi_b32 %4, %0, %1, %4"
                : "=&v"(m), "=&v"(t), "+v"(st), "+v"(cnt), "+v"(l0) : "s"(sel));
        else
            asm("v_subrev_u32_e32 %1, 0x8000, %2\n\t"
                "v_ashrrev_i32_e32 %0, 31, %1\n\t"
                "v_sub_u32_e32 %3, %3, %0\n\t"
                "v_perm_b32 %1, %2, %4, %5\n\t"
                "v_bfi_b32 %2, %0, %1, %2"
                : "=&v"(m), "=&v"(t), "+v"(st), "+v"(cnt) : "v"(l0), "s"(sel));
    }
    // the group took cnt (0..4) units; W = fetch() of the group's start
    __device__ __forceinline__ void end_group(u32 cnt, const uint4 W)
    {
        rpos += cnt << 1;
        const u32 m0 = (u32)__builtin_amdgcn_sbfe((int)cnt, 0, 1), m1 = (u32)__builtin_amdgcn_sbfe((int)cnt, 1, 1), m2 = (u32)__builtin_amdgcn_sbfe((int)cnt, 2, 1);
        const u32 a0 = trc_bfi(m2, W.z, W.x), a1 = trc_bfi(m2, W.w, W.y), a2 = W.z;
        const u32 b0 = trc_bfi(m1, a1, a0), b1 = trc_bfi(m1, a2, a1), b2 = trc_bfi(m1, W.w, a2);
        l0 = trc_bfi(m0, __builtin_amdgcn_alignbit(b1, b0, 16), b0);
        l1 = trc_bfi(m0, __builtin_amdgcn_alignbit(b2, b1, 16), b1);
    }
};

// The same for the range decoders of the byte coders (32-bit words, at most two per group of two bytes): the two words at the stream
// position ride in registers, one 16-byte load per group, the next pair is W[cnt], W[cnt + 1].
struct LaneLook32 {
    const u8 *src;
    u32 rpos, lim;
    u32 w0, w1;          // the words at rpos and rpos + 4
    // rcdinit's two words come back in a / b, the look-ahead starts behind them
    __device__ __forceinline__ void prime(const u8 *s, u32 limit, u32 &a, u32 &b)
    {
        src = s; rpos = 8u; lim = limit;
        const uint4 q = trc_ld16_a2(src);
        a = q.x; b = q.y; w0 = q.z; w1 = q.w;
    }
    __device__ __forceinline__ uint4 fetch() const { return trc_ld16_a2(src + trc_min(rpos, lim)); }
    __device__ __forceinline__ void end_group(u32 cnt, const uint4 W)      // cnt = 0, 1, 2 words taken
    {
        rpos += cnt << 2;
        const u32 m0 = (u32)__builtin_amdgcn_sbfe((int)cnt, 0, 1), m1 = (u32)__builtin_amdgcn_sbfe((int)cnt, 1, 1);
        w0 = trc_bfi(m1, W.z, trc_bfi(m0, W.y, W.x));
        w1 = trc_bfi(m1, W.w, trc_bfi(m0, W.z, W.y));
    }
    __device__ __forceinline__ void end_group1(u32 cnt, const uint4 W)     // groups that take at most ONE word: only w0 is carried
    {
        rpos += cnt << 2;
        w0 = trc_bfi((u32)__builtin_amdgcn_sbfe((int)cnt, 0, 1), W.y, W.x);
    }
};

