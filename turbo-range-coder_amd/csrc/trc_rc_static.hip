// trc_rc_static.hip -- static-CDF range coder: one stream (TRC_RCS1), two interleaved streams (TRC_RCS2, what
// `turborc -e45` runs), and the 32-bit-range / 16-bit-I/O variant of the one-stream coder (TRC_RCSM, `turborc -e44`).
//
// Per chunk the payload is exactly what the reference returns for that slice:
//   RCS1  rccdfsenc  (rccdf.c:71-81)   : [u32 words of one 64-bit range coder]
//   RCS2  rccdfs2enc (rccdf.c:125-143) : [u32 len0][stream 0: even-index bytes (+ odd tail byte)][stream 1: odd-index bytes]
//   RCSM  rccdfsmenc (rccdf.c:655-665) : [u16 words of one 32-bit range coder]   (GEO = 1 below; trc_rc.h RcEncSm/RcDecSm)
// including the raw fallbacks (OVERFLOW rcutil_.h:130, OVERFLOWI rccdf.c:46).  All reference
// decoders of a stream (linear/binary/division search, rccdf.c:84-122,146-184) select the same
// symbols; here the symbol comes from code/range (exact, trc_rc.h) through the slot->symbol LUT.
//
// One lane = one chunk (RCS2: both of its coders).  Both overflow tests are monotone in the number
// of words written, so evaluating them at 16-symbol periods and once after the last symbol decides
// exactly like the reference's per-symbol test; the early exit only bounds the scratch writes.
#include <stdlib.h>
#include "trc_rc.h"
#include "trc_launch.h"

#define RCS_WAVE_LDS(NS) ((NS) * TRC_SRING_BYTES)     // chunk bytes travel through in-register quad transposes
// the decoders' slot -> symbol LUT sits at LDS offset 0: read on the integer address (through the generic pointer the compiler adds
// the segment base, a relocated 0, to every index); the kernels trap at entry if the dynamic segment does not start at 0
#define RCS_LUT(t) ((u32)*(const trc_lds_u8 *)(uintptr_t)(t))

#ifndef TRC_RCS_EXACT_Q
#define TRC_RCS_EXACT_Q 0       // 1: every step through the exact quotient (rounds 2-4)
#endif
// one symbol of the 64-bit static range decoder against the look-ahead word w: symbol from the estimated quotient, verified against
// its own bounds (RcDec::probe); the exact quotient only where some lane of the wave fails.  Returns the symbol, `rn` = renormalised.
__device__ __forceinline__ u32 rcs_step(RcDec &d, const u32 *tab, u32 w, bool &rn)
{
#if TRC_RCS_EXACT_Q
    const u32 x = RCS_LUT(d.quotient15());
    const u32 t = tab[x];
    rn = d.consume_w(true, t & 0xffffu, (t & 0xffffu) + (t >> 16), w);
    return x;
#else
    u32 x = RCS_LUT(d.estimate15());
    u32 t = tab[x];
    RcDec::Probe q = d.probe(t & 0xffffu, (t & 0xffffu) + (t >> 16));
    if (__ballot(!q.fits)) {                                   // rare, wave-uniform (and every step of a corrupt stream: the exact quotient is clamped into the table)
        x = RCS_LUT(d.quotient15());
        t = tab[x];
        q = d.probe(t & 0xffffu, (t & 0xffffu) + (t >> 16));
    }
    rn = d.commit(q, w);
    return x;
#endif
}
template <int GEO> struct RcGeo;
template <> struct RcGeo<0> { typedef RcEncV Enc; typedef RcDec Dec; };      // (RcEncV: the state on 32-bit halves with a carry limb, trc_rc.h)
template <> struct RcGeo<1> { typedef RcEncSm Enc; typedef RcDecSm Dec; };

// WPB waves per workgroup: 1 (rounds 1-4; more than one residency round: small workgroups move in as old ones end) or 12 with
// TrcPace (round 5: a launch that is ONE round -- the waves of a SIMD then sit in one workgroup and keep each other's pace)
#define RCS_ENC_FIXED (1024u + 64u)                           // symbol table, TrcPace's progress counters
template <int NS, int GEO, int WPB>
__global__ __launch_bounds__(64 * WPB) void trc_rcs_enc_kernel(
    const u8 *__restrict__ in, u64 n, u32 chunk, u32 nchunks, const u32 *__restrict__ tab_g,
    u8 *__restrict__ scrA, u32 strideA, u8 *__restrict__ scrB, u32 strideB,
    u32 *__restrict__ clen, u32 *__restrict__ gsum)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    u32 *tab = (u32 *)smem;                                    // 256 x {f<<16 | c0}
    const u32 tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    u8 *wbase = smem + RCS_ENC_FIXED + wv * RCS_WAVE_LDS(NS);
    for (u32 i = tid; i < 256; i += 64 * WPB) tab[i] = tab_g[i];
    TrcPace pace; pace.init(trc_lds_addr(smem) + 1024u, tid, wv);
    __syncthreads();

    WaveChunks wc;
    wc.c0 = (blockIdx.x * WPB + wv) * 64u; wc.chunk = chunk; wc.nchunks = nchunks;
    wc.lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    if (wc.c0 >= nchunks) return;
    wc.rows = nchunks - wc.c0 < 64u ? nchunks - wc.c0 : 64u;
    const bool alive = lane < wc.rows;
    const u32 c = wc.c0 + lane;
    const u32 len = alive ? wc.len_of(lane) : 0u;
    const int lim = trc_rc_limit(len);

    QuadIn tin; tin.base = in + (u64)wc.c0 * chunk;
    StreamOut<false, false, false, NS == 1> so0, so1;          // (one stream: no length header stored behind the drains -> write-through, trc_io.h)
    so0.rings = wbase;
    so0.scratch = scrA; so0.stride = strideA; so0.c0 = wc.c0; so0.wpos = (NS == 2) ? 4u : 0u; so0.nfl = 0;
    so1 = so0;
    if (NS == 2) { so1.rings = so0.rings + TRC_SRING_BYTES; so1.scratch = scrB; so1.stride = strideB; so1.wpos = 0; }
    typedef typename RcGeo<GEO>::Enc Enc;
    constexpr u32 WB = Enc::WBYTES;
    Enc e0, e1; e0.start(); e1.start();
    const u32 off1 = (NS == 2 && len >= 4u) ? 4u + ((len - 4u) * 37u) / 64u : 0u;   // stream-1 base inside `out` (rccdf.c:126)
    bool ovf = alive && (NS == 2 ? len < 10u : lim <= 0);       // tiny inputs: always raw
    const u32 pairs = len & ~1u;

    const u32 S = chunk / TRC_SEG;
    tin.issue(wc, 0);
    for (u32 s = 0; s < S; s++) {
        if (WPB > 4) pace.step(s + 1u);
        tin.commit();
        if (s + 1 < S) tin.issue(wc, (s + 1) * TRC_SEG);
        // The piece and dword loops are kept as loops (registers rotate instead of being indexed): fully unrolled, the
        // 64 coding steps of a segment with their carry paths were 94-174 KB of code, more than the instruction cache.
        uint4 pc0 = tin.read(0), pc1 = tin.read(1), pc2 = tin.read(2), pc3 = tin.read(3);
#pragma nounroll
        for (u32 k = 0; k < 4; k++) {
            uint4 v = pc0; pc0 = pc1; pc1 = pc2; pc2 = pc3;
            const u32 p0 = s * TRC_SEG + k * 16u;
            const bool act = alive && !ovf && p0 < len;
            if (act && p0 + 16u <= len) {
#pragma nounroll
                for (u32 d = 0; d < 4; d++) {
                    const u32 wd = v.x; v.x = v.y; v.y = v.z; v.z = v.w;
                    const u32 t0 = tab[wd & 255u], t1 = tab[(wd >> 8) & 255u], t2 = tab[(wd >> 16) & 255u], t3 = tab[wd >> 24];
                    if constexpr (GEO == 0) {                   // 64-bit geometry: at most one word per two symbols (trc_rc.h RcEncD)
                        if (NS == 1) {
                            e0.sym_rec(true, t0 & 0xffffu, t0 >> 16); e0.sym_rec(true, t1 & 0xffffu, t1 >> 16); e0.flush(so0);
                            e0.sym_rec(true, t2 & 0xffffu, t2 >> 16); e0.sym_rec(true, t3 & 0xffffu, t3 >> 16); e0.flush(so0);
                        } else {
                            e0.sym_rec(true, t0 & 0xffffu, t0 >> 16); e1.sym_rec(true, t1 & 0xffffu, t1 >> 16);
                            e0.sym_rec(true, t2 & 0xffffu, t2 >> 16); e1.sym_rec(true, t3 & 0xffffu, t3 >> 16);
                            e0.flush(so0); e1.flush(so1);
                        }
                    } else if (NS == 1) {
                        e0.sym(so0, t0 & 0xffffu, t0 >> 16); e0.sym(so0, t1 & 0xffffu, t1 >> 16);
                        e0.sym(so0, t2 & 0xffffu, t2 >> 16); e0.sym(so0, t3 & 0xffffu, t3 >> 16);
                    } else {
                        e0.sym(so0, t0 & 0xffffu, t0 >> 16); e1.sym(so1, t1 & 0xffffu, t1 >> 16);
                        e0.sym(so0, t2 & 0xffffu, t2 >> 16); e1.sym(so1, t3 & 0xffffu, t3 >> 16);
                    }
                }
            } else if (act) {                                   // last chunk's final partial piece
                const u8 *mine = in + (u64)c * chunk;               // (one lane in the whole grid: plain byte loads)
                for (u32 pos = p0; pos < len; pos++) {
                    const u32 t = tab[mine[pos]];
                    if (NS == 2 && pos < pairs && (pos & 1u)) {
                        e1.sym(so1, t & 0xffffu, t >> 16);          // pair complete: OVERFLOWI (never after the odd tail byte)
                        ovf = ovf || ((int)(off1 + WB * e1.cw.nwords) >= lim) || (4u + WB * e0.cw.nwords >= off1);
                    } else e0.sym(so0, t & 0xffffu, t >> 16);
                }
            }
            so0.drain(false, alive);
            if (NS == 2) so1.drain(false, alive);
            // monotone overflow tests (position = last byte coded so far is < pairs for every full piece)
            if (NS == 1) ovf = ovf || (alive && (int)(WB * e0.cw.nwords) >= lim);
            else if (act && p0 + 16u <= len) ovf = ovf || ((int)(off1 + WB * e1.cw.nwords) >= lim) || (4u + WB * e0.cw.nwords >= off1);
        }
    }
    u32 out_len = 0;
    if (alive) {
        if (!ovf) {
            e0.finish(so0);
            if (NS == 2) {
                e1.finish(so1);
                out_len = so0.wpos + so1.wpos;                  // 4 + len0 + len1
                if ((int)out_len >= lim) ovf = true;
            } else out_len = so0.wpos;
        }
        if (ovf) out_len = len;
    }
    so0.drain(true, alive && !ovf);
    if (NS == 2) {
        so1.drain(true, alive && !ovf);
        if (alive && !ovf) *(u32 *)(scrA + (u64)c * strideA) = so0.wpos - 4u;   // header: len0 (the segment holding it is already in place)
    }
    if (alive) clen[c] = out_len;
    const u32 gs = trc_wave_sum(out_len);
    if (lane == 0) gsum[wc.c0 >> 6] = gs;
}

#define RCS_DEC_FIXED (32768u + 1024u + 64u)                 // slot -> symbol LUT, symbol table, TrcPace's progress counters (trc_dev.h)
template <int NS, int GEO>
__global__ __launch_bounds__(896) void trc_rcs_dec_kernel(
    const u8 *__restrict__ payload, const u32 *__restrict__ clen, const u64 *__restrict__ goff, const u32 *__restrict__ gsum,
    u64 n, u32 chunk, u32 nchunks, const u8 *__restrict__ lut_g, const u32 *__restrict__ tab_g, u8 *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    u8 *lut = smem;                                            // 32768
    u32 *tab = (u32 *)(smem + 32768);                          // 256 x {f<<16 | c0}
    if (trc_lds_addr(smem) != 0u) __builtin_trap();            // (RCS_LUT: absolute LDS offsets)
    const u32 tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6, BLOCK = blockDim.x;
    u8 *wbase = smem + RCS_DEC_FIXED + wv * RCS_WAVE_LDS(NS);
    if (BLOCK >= 704u) {                                        // one batch of loads (a copy loop waits for each load before the next: trc_ans_static.hip)
        uint4 t0 = ((const uint4 *)lut_g)[tid], t1 = ((const uint4 *)lut_g)[tid + BLOCK], t2 = make_uint4(0, 0, 0, 0);
        u32 td = 0;
        if (tid + 2u * BLOCK < 2048u) t2 = ((const uint4 *)lut_g)[tid + 2u * BLOCK];
        if (tid < 256u) td = tab_g[tid];
        ((uint4 *)lut)[tid] = t0; ((uint4 *)lut)[tid + BLOCK] = t1;
        if (tid + 2u * BLOCK < 2048u) ((uint4 *)lut)[tid + 2u * BLOCK] = t2;
        if (tid < 256u) tab[tid] = td;
    } else {
        for (u32 i = tid; i < 2048; i += BLOCK) ((uint4 *)lut)[i] = ((const uint4 *)lut_g)[i];
        for (u32 i = tid; i < 256; i += BLOCK) tab[i] = tab_g[i];
    }
    TrcPace pace; pace.init(32768u + 1024u, tid, wv);          // the waves of a SIMD keep each other's pace (trc_dev.h)
    __syncthreads();

    WaveChunks wc;
    wc.c0 = (blockIdx.x * (BLOCK / 64) + wv) * 64u; wc.chunk = chunk; wc.nchunks = nchunks;
    wc.lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    if (wc.c0 >= nchunks) return;
    wc.rows = nchunks - wc.c0 < 64u ? nchunks - wc.c0 : 64u;
    const bool alive = lane < wc.rows;
    const u32 c = wc.c0 + lane;
    const u32 len = alive ? wc.len_of(lane) : 0u;
    const u32 cl = alive ? trc_min(clen[c], len) : 0u;        // a directory entry above the chunk length (corrupt input) reads as raw
    const u32 ex = trc_wave_incl_scan(cl) - cl;
    const u64 off = trc_group_base(goff, gsum, wc.c0 >> 6) + ex;
    const bool coded = alive && cl != len;

    QuadOut tout; tout.base = out + (u64)wc.c0 * chunk;
    StreamIn s0, s1;
    s0.rings = wbase;
    const u32 len0 = (NS == 2 && coded) ? trc_min(trc_ld32_a2(payload + off), trc_sub_sat(cl, 4u)) : 0u;   // a corrupt header cannot point outside the chunk's payload
    s0.gbase = payload; s0.soff = off + (NS == 2 ? 4u : 0u); s0.lim = NS == 2 ? len0 : cl;
    s1 = s0;
    if (NS == 2) {
        s1.rings = s0.rings + TRC_SRING_BYTES;
        s1.soff = off + 4u + len0; s1.lim = trc_sub_sat(cl, 4u + len0);
    }
    const u32 r00 = s0.align_start(coded), r01 = NS == 2 ? s1.align_start(coded) : 0u;      // segments aligned to the payload's 64-byte sectors (trc_io.h)
    s0.prime(coded); s0.rpos = r00;
    if (NS == 2) { s1.prime(coded); s1.rpos = r01; }
    typedef typename RcGeo<GEO>::Dec Dec;
    Dec d0, d1;
    d0.init(s0);
    if (NS == 2) d1.init(s1); else d1 = d0;

    auto get = [&](Dec &d, StreamIn &si) -> u32 {
        if constexpr (GEO == 0) {                                  // one correction step each way, no loops, predicated renorm
                                                                   // (rccdfs decode 194 -> 182 us, rccdfs2 288 -> 231 us)
            const u32 x = RCS_LUT(d.quotient15());
            const u32 t = tab[x];
            d.consume_if(si, true, t & 0xffffu, (t & 0xffffu) + (t >> 16));
            return x;
        } else {
            // (the estimate-and-verify step of the 64-bit decoder, rcs_step, was tried here too -- RcDecSm::slot_estimate / fits -- and is
            // slower, 0.123 -> 0.137 ms: this geometry's exact quotient is one 32 x 17-bit product and two predicated corrections)
            const u32 x = RCS_LUT(d.slot());
            const u32 t = tab[x];
            d.consume(si, t & 0xffffu, (t & 0xffffu) + (t >> 16));
            return x;
        }
    };

    // two symbols of one stream (64-bit geometry): at most one of them renormalises (trc_rc.h RcEncD), so the pair shares
    // one look-ahead word (one ring read) and one stream advance
    auto get2 = [&](Dec &d, StreamIn &si, u32 &xa, u32 &xb) {
        if constexpr (GEO == 0) {
            const u32 w = si.peek32();
            bool ra, rb;
            xa = rcs_step(d, tab, w, ra);
            xb = rcs_step(d, tab, w, rb);
            si.skip_if(ra || rb);
        } else { xa = get(d, si); xb = get(d, si); }
    };

    const u32 S = chunk / TRC_SEG;
    const u32 pairs = len & ~1u;
    u8 *dst = out + (u64)c * chunk;
    for (u32 s = 0; s < S; s++) {
        pace.step(s + 1u);
        // rolled like the encoder's loops (the refill protocol wants the period parity as a literal: pieces go in pairs)
        uint4 pc0 = make_uint4(0, 0, 0, 0), pc1 = pc0, pc2 = pc0, pc3 = pc0;
#pragma nounroll
        for (u32 kk = 0; kk < 2; kk++) {
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const u32 p0 = s * TRC_SEG + (2u * kk + (u32)j) * 16u;
                s0.period(coded && p0 < len, j);
                if (NS == 2) s1.period(coded && p0 < len, j);
                uint4 v = make_uint4(0, 0, 0, 0);
                if (coded && p0 + 16u <= len) {
#pragma nounroll
                    for (u32 d = 0; d < 4; d++) {
                        u32 x0, x1, x2, x3;
                        if (NS == 1) { get2(d0, s0, x0, x1); get2(d0, s0, x2, x3); }
                        else if (GEO == 0) { get2(d0, s0, x0, x2); get2(d1, s1, x1, x3); }      // the streams are independent of each other
                        else         { x0 = get(d0, s0); x1 = get(d1, s1); x2 = get(d0, s0); x3 = get(d1, s1); }
                        v.x = v.y; v.y = v.z; v.z = v.w; v.w = x0 | (x1 << 8) | (x2 << 16) | (x3 << 24);
                    }
                } else if (coded && p0 < len) {
                    for (u32 pos = p0; pos < len; pos++)
                        dst[pos] = (u8)((NS == 2 && pos < pairs && (pos & 1u)) ? get(d1, s1) : get(d0, s0));
                }
                pc0 = pc1; pc1 = pc2; pc2 = pc3; pc3 = v;
            }
        }
        tout.put(0, pc0); tout.put(1, pc1); tout.put(2, pc2); tout.put(3, pc3);
        tout.flush(wc, s * TRC_SEG);
    }
    trc_wave_copy_raw(__ballot(alive && cl == len && len != 0), off, len, out + (u64)wc.c0 * chunk, chunk, payload);
}

// -------------------------------------------------------------------- RCS2, one LANE per stream ---
// Round 3.  The two streams of rccdfs2enc are independent byte ranges (rccdf.c:125-143: even-index bytes + an odd tail on
// stream 0, odd-index bytes on stream 1): lanes 2i and 2i + 1 take chunk i of the wave, one stream each -- twice the lanes
// for the same bytes, one ring per lane instead of two (16 waves per CU fit where the one-lane form held 9), and a lane's
// step is the one-stream coder's.  The streams meet only in the overflow tests (word counts exchanged by DPP once per
// period) and at the end (lengths); the decoder merges the two lanes' bytes by DPP before the output transpose.
// A workgroup is two waves = one group of 64 chunks, so that gsum keeps its meaning.
// GPW groups of 64 chunks (= pairs of waves) per workgroup: 1, or 6 with TrcPace when the launch is one residency round (round 5)
template <int GPW>
__global__ __launch_bounds__(128 * GPW) void trc_rcs2p_enc_kernel(
    const u8 *__restrict__ in, u64 n, u32 chunk, u32 nchunks, const u32 *__restrict__ tab_g,
    u8 *__restrict__ scrA, u32 strideA, u8 *__restrict__ scrB, u32 strideB,
    u32 *__restrict__ clen, u32 *__restrict__ gsum)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    u32 *tab = (u32 *)smem;                                    // 256 x {f<<16 | c0}
    u32 *wsum = (u32 *)(smem + 1024);                          // the waves' byte counts (a group = two waves)
    const u32 tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    u8 *wbase = smem + 1024 + 64 + 64 + wv * RCS_WAVE_LDS(1);
    for (u32 i = tid; i < 256; i += 128 * GPW) tab[i] = tab_g[i];
    TrcPace pace; pace.init(trc_lds_addr(smem) + 1024u + 64u, tid, wv);
    __syncthreads();

    WaveChunks wc;
    wc.c0 = (blockIdx.x * GPW) * 64u + wv * 32u; wc.chunk = chunk; wc.nchunks = nchunks;
    wc.lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    wc.rows = wc.c0 >= nchunks ? 0u : nchunks - wc.c0 < 32u ? nchunks - wc.c0 : 32u;
    u32 gs = 0;
    if (wc.rows) {
        const u32 ci = lane >> 1, b = lane & 1u;
        const bool alive = ci < wc.rows;
        const u32 c = wc.c0 + ci;
        const u32 len = alive ? wc.len_of(ci) : 0u;
        const int lim = trc_rc_limit(len);

        QuadIn tin; tin.base = in + (u64)wc.c0 * chunk;
        StreamOut<false, true> so;
        so.rings = wbase;
        so.scratch = scrA; so.stride = strideA; so.scratch_b = scrB; so.stride_b = strideB; so.c0 = wc.c0;
        so.wpos = b ? 0u : 4u; so.nfl = 0;
        RcEncV e; e.start();                                   // (state on 32-bit halves with a carry limb: trc_rc.h; RcEncD: 0.141 ms)
        const u32 off1 = len >= 4u ? 4u + ((len - 4u) * 37u) / 64u : 0u;       // stream-1 base inside `out` (rccdf.c:126)
        bool ovf = alive && len < 10u;                                         // tiny inputs: always raw
        const u32 pairs = len & ~1u;
        // OVERFLOWI (rccdf.c:46) on the two lanes' word counts
        auto overflowi = [&]() {
            const u32 mine = e.cw.nwords, other = trc_quad_xor1(mine), n0 = b ? other : mine, n1 = b ? mine : other;
            return ((int)(off1 + 4u * n1) >= lim) || (4u + 4u * n0 >= off1);
        };

        const u32 S = chunk / TRC_SEG, sh0 = 8u * b, sh1 = 16u + 8u * b;
        tin.issue(wc, 0, true);
        for (u32 s = 0; s < S; s++) {
            if (GPW > 2) pace.step(s + 1u);
            tin.commit();
            if (s + 1 < S) tin.issue(wc, (s + 1) * TRC_SEG, true);
            uint4 pc0 = tin.read(0), pc1 = tin.read(1), pc2 = tin.read(2), pc3 = tin.read(3);
#pragma nounroll
            for (u32 k = 0; k < 4; k++) {
                uint4 v = pc0; pc0 = pc1; pc1 = pc2; pc2 = pc3;
                const u32 p0 = s * TRC_SEG + k * 16u;
                const bool act = alive && !ovf && p0 < len;
                const bool full = act && p0 + 16u <= len;
                if (full) {
#pragma nounroll
                    for (u32 d = 0; d < 4; d++) {               // a dword = two pairs: this lane's bytes are b and b + 2
                        const u32 wd = v.x; v.x = v.y; v.y = v.z; v.z = v.w;
                        const u32 ta = tab[(wd >> sh0) & 255u], tb = tab[(wd >> sh1) & 255u];
                        e.sym<false>(true, ta & 0xffffu, ta >> 16); e.sym<false>(true, tb & 0xffffu, tb >> 16); e.flush(so, true);
                    }
                } else if (act) {                               // last chunk's final partial piece (both lanes of its pair walk it)
                    const u8 *mine = in + (u64)c * chunk;
                    for (u32 pos = p0; pos < len; pos++) {
                        const u32 t = tab[mine[pos]];
                        const bool second = pos < pairs && (pos & 1u);          // odd-index byte of a pair; the odd tail byte is stream 0's
                        if ((second ? 1u : 0u) == b) e.sym(so, t & 0xffffu, t >> 16);
                        if (second) ovf = ovf || overflowi();   // pair complete (never after the odd tail byte)
                    }
                }
                so.drain(false, alive);
                const bool o = overflowi();                     // (all lanes: the exchange must not sit in a branch)
                if (full) ovf = ovf || o;
            }
        }
        u32 out_len = 0;
        if (alive && !ovf) e.finish(so);
        {
            const u32 mine = so.wpos, other = trc_quad_xor1(mine);
            if (alive) {
                out_len = mine + other;                         // 4 + len0 + len1
                if (!ovf && (int)out_len >= lim) ovf = true;
                if (ovf) out_len = len;
            }
        }
        so.drain(true, alive && !ovf);
        if (alive && !ovf && b == 0u) *(u32 *)(scrA + (u64)c * strideA) = so.wpos - 4u;   // header: len0 (the segment holding it is already in place)
        if (alive && b == 0u) clen[c] = out_len;
        gs = trc_wave_sum(b == 0u ? out_len : 0u);
    }
    if (lane == 0) wsum[wv] = gs;
    __syncthreads();
    if (lane == 0 && !(wv & 1u) && (blockIdx.x * GPW + (wv >> 1)) * 64u < nchunks) gsum[blockIdx.x * GPW + (wv >> 1)] = wsum[wv] + wsum[wv + 1];
}

__global__ __launch_bounds__(896) void trc_rcs2p_dec_kernel(
    const u8 *__restrict__ payload, const u32 *__restrict__ clen, const u64 *__restrict__ goff, const u32 *__restrict__ gsum,
    u64 n, u32 chunk, u32 nchunks, const u8 *__restrict__ lut_g, const u32 *__restrict__ tab_g, u8 *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    u8 *lut = smem;                                            // 32768
    u32 *tab = (u32 *)(smem + 32768);                          // 256 x {f<<16 | c0}
    if (trc_lds_addr(smem) != 0u) __builtin_trap();            // (RCS_LUT: absolute LDS offsets)
    const u32 tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6, BLOCK = blockDim.x;
    u8 *wbase = smem + RCS_DEC_FIXED + wv * RCS_WAVE_LDS(1);
    if (BLOCK >= 704u) {
        uint4 t0 = ((const uint4 *)lut_g)[tid], t1 = ((const uint4 *)lut_g)[tid + BLOCK], t2 = make_uint4(0, 0, 0, 0);
        u32 td = 0;
        if (tid + 2u * BLOCK < 2048u) t2 = ((const uint4 *)lut_g)[tid + 2u * BLOCK];
        if (tid < 256u) td = tab_g[tid];
        ((uint4 *)lut)[tid] = t0; ((uint4 *)lut)[tid + BLOCK] = t1;
        if (tid + 2u * BLOCK < 2048u) ((uint4 *)lut)[tid + 2u * BLOCK] = t2;
        if (tid < 256u) tab[tid] = td;
    } else {
        for (u32 i = tid; i < 2048; i += BLOCK) ((uint4 *)lut)[i] = ((const uint4 *)lut_g)[i];
        for (u32 i = tid; i < 256; i += BLOCK) tab[i] = tab_g[i];
    }
    TrcPace pace; pace.init(32768u + 1024u, tid, wv);
    __syncthreads();

    WaveChunks wc;
    wc.c0 = (blockIdx.x * (BLOCK / 64) + wv) * 32u; wc.chunk = chunk; wc.nchunks = nchunks;
    wc.lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    if (wc.c0 >= nchunks) return;
    wc.rows = nchunks - wc.c0 < 32u ? nchunks - wc.c0 : 32u;
    // the directory of the whole group of 64 chunks (payload offsets are per group): lane L reads chunk (group base + L), the
    // pair takes its chunk's numbers from there
    const u32 g0 = wc.c0 & ~63u, cg = g0 + lane;
    const u32 lenL = cg < nchunks ? ((cg == nchunks - 1u) ? wc.lastlen : chunk) : 0u;
    const u32 clL = cg < nchunks ? trc_min(clen[cg], lenL) : 0u;   // a directory entry above the chunk length (corrupt input) reads as raw
    const u32 exL = trc_wave_incl_scan(clL) - clL;
    const u64 gbase = trc_group_base(goff, gsum, wc.c0 >> 6);
    const u32 ci = lane >> 1, b = lane & 1u, srcl = (wc.c0 & 32u) + ci;
    const u32 cl = (u32)__shfl((int)clL, (int)srcl, 64), ex = (u32)__shfl((int)exL, (int)srcl, 64), len = (u32)__shfl((int)lenL, (int)srcl, 64);
    const bool alive = ci < wc.rows;
    const u32 c = wc.c0 + ci;
    const u64 off = gbase + ex;
    const bool coded = alive && cl != len;

    QuadOut tout; tout.base = out + (u64)wc.c0 * chunk;
    StreamIn si;
    si.rings = wbase;
    const u32 len0 = coded ? trc_min(trc_ld32_a2(payload + off), trc_sub_sat(cl, 4u)) : 0u;   // a corrupt header cannot point outside the chunk's payload
    si.gbase = payload;
    si.soff = off + 4u + (b ? len0 : 0u);
    si.lim = b ? trc_sub_sat(cl, 4u + len0) : len0;
    // (segments aligned to the payload's sectors, StreamInT::align_start: measured here and not taken -- with one lane per stream the
    // streams are half as long, the skipped bytes of the first segment cost relatively more refills: 0.140 -> 0.143-0.145 ms)
    si.prime(coded);
    RcDec d; d.init(si);

    const u32 S = chunk / TRC_SEG;
    const u32 pairs = len & ~1u;
    u8 *dst = out + (u64)c * chunk;
    for (u32 s = 0; s < S; s++) {
        pace.step(s + 1u);
        uint4 pc0 = make_uint4(0, 0, 0, 0), pc1 = pc0, pc2 = pc0, pc3 = pc0;
#pragma nounroll
        for (u32 kk = 0; kk < 2; kk++) {
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const u32 p0 = s * TRC_SEG + (2u * kk + (u32)j) * 16u;
                si.period(coded && p0 < len, j);
                uint4 v = make_uint4(0, 0, 0, 0);
                const bool full = coded && p0 + 16u <= len;
#pragma nounroll
                for (u32 q = 0; q < 4; q++) {                   // a dword = two pairs: this lane decodes bytes b and b + 2 of it
                    u32 m = 0;
                    if (full) {
                        // two symbols of one stream: at most one of them renormalises (trc_rc.h RcEncD), one look-ahead word
                        const u32 w = si.peek32();
                        bool ra, rb;
                        const u32 xa = rcs_step(d, tab, w, ra);
                        const u32 xb = rcs_step(d, tab, w, rb);
                        si.skip_if(ra || rb);
                        m = (xa | (xb << 16)) << (8u * b);
                    }
                    const u32 both = m | trc_quad_xor1(m);      // (every lane: the exchange must not sit in a branch)
                    v.x = v.y; v.y = v.z; v.z = v.w; v.w = both;
                }
                if (!full && coded && p0 < len) {               // last chunk's final partial piece (both lanes of its pair walk it)
                    for (u32 pos = p0; pos < len; pos++) {
                        const bool second = pos < pairs && (pos & 1u);
                        if ((second ? 1u : 0u) == b) {
                            const u32 x = RCS_LUT(d.quotient15());
                            const u32 t = tab[x];
                            d.consume_if(si, true, t & 0xffffu, (t & 0xffffu) + (t >> 16));
                            dst[pos] = (u8)x;
                        }
                    }
                }
                pc0 = pc1; pc1 = pc2; pc2 = pc3; pc3 = v;
            }
        }
        tout.put(0, pc0); tout.put(1, pc1); tout.put(2, pc2); tout.put(3, pc3);
        tout.flush(wc, s * TRC_SEG, true);
    }
    // raw chunks: lanes 0..31 carry their chunks' numbers for the wave copy
    {
        const u32 from = (lane & 31u) << 1;
        const u32 olo = (u32)__shfl((int)(u32)off, (int)from, 64), ohi = (u32)__shfl((int)(u32)(off >> 32), (int)from, 64);
        const u32 l2 = (u32)__shfl((int)len, (int)from, 64), c2 = (u32)__shfl((int)cl, (int)from, 64);
        const bool raw = lane < wc.rows && c2 == l2 && l2 != 0u;
        trc_wave_copy_raw(__ballot(raw), ((u64)ohi << 32) | olo, l2, out + (u64)wc.c0 * chunk, chunk, payload);
    }
}

// ------------------------------------------------------------------------------------- launch ---
// TRC_RCS2_PAIR=0: the one-lane-per-chunk form of the two-stream coder (rounds 1-2), kept for A/B measurements
static bool rcs2_pair() { static const bool on = !(getenv("TRC_RCS2_PAIR") && atoi(getenv("TRC_RCS2_PAIR")) == 0); return on; }

void trc_launch_rcs_enc(int nstreams, const uint8_t *d_in, size_t n, uint32_t chunk, const TrcWork &w,
                        uint32_t *d_clen, hipStream_t s)
{
    const u32 *tab = (const u32 *)(w.tables + TRC_TAB_DEC);
    // one residency round (at most twelve waves per CU): workgroups of twelve waves that keep each other's pace (TrcPace)
    static const int env_wpb = getenv("TRC_RCS_ENC_WPB") ? atoi(getenv("TRC_RCS_ENC_WPB")) : 0;      // tuning aid: 1 / 12 force the form
    const bool big = env_wpb ? env_wpb == 12 : (w.ngroups >= 2048u && w.ngroups <= 12u * 256u);
#define RCS_ENC_LAUNCH(NS, GEO, SB, STB)                                                                                             \
    do {                                                                                                                             \
        if (big) {                                                                                                                   \
            TRC_RAISE_LDS_ONCE((trc_rcs_enc_kernel<NS, GEO, 12>), RCS_ENC_FIXED + 12 * RCS_WAVE_LDS(NS));                            \
            TRC_LAUNCH_TIMED((trc_rcs_enc_kernel<NS, GEO, 12>), dim3((w.ngroups + 11u) / 12u), dim3(64 * 12), RCS_ENC_FIXED + 12 * RCS_WAVE_LDS(NS), s, \
                               d_in, (u64)n, chunk, w.nchunks, tab, w.scratch, w.stride, SB, STB, d_clen, w.gsum);                   \
        } else                                                                                                                       \
            TRC_LAUNCH_TIMED((trc_rcs_enc_kernel<NS, GEO, 1>), dim3(w.ngroups), dim3(64), RCS_ENC_FIXED + RCS_WAVE_LDS(NS), s,       \
                               d_in, (u64)n, chunk, w.nchunks, tab, w.scratch, w.stride, SB, STB, d_clen, w.gsum);                   \
    } while (0)
    if (nstreams == 1) RCS_ENC_LAUNCH(1, 0, w.scratch, w.stride);
    else if (nstreams == 2 && rcs2_pair())
    {
        // (a group of 64 chunks is two waves here: 2 x ngroups waves in the launch)
        const bool big2 = env_wpb ? env_wpb == 12 : (w.ngroups >= 1024u && w.ngroups <= 6u * 256u);
        if (big2) {
            TRC_RAISE_LDS_ONCE(trc_rcs2p_enc_kernel<6>, 1024 + 128 + 12 * RCS_WAVE_LDS(1));
            TRC_LAUNCH_TIMED(trc_rcs2p_enc_kernel<6>, dim3((w.ngroups + 5u) / 6u), dim3(128 * 6), 1024 + 128 + 12 * RCS_WAVE_LDS(1), s,
                               d_in, (u64)n, chunk, w.nchunks, tab, w.scratch, w.stride, w.scratch2, w.stride2, d_clen, w.gsum);
        } else
            TRC_LAUNCH_TIMED(trc_rcs2p_enc_kernel<1>, dim3(w.ngroups), dim3(128), 1024 + 128 + 2 * RCS_WAVE_LDS(1), s,
                               d_in, (u64)n, chunk, w.nchunks, tab, w.scratch, w.stride, w.scratch2, w.stride2, d_clen, w.gsum);
    }
    else if (nstreams == 2) {
        const bool big2 = false;                               // (two rings per wave: six waves per workgroup would fit; not measured)
        (void)big2;
        TRC_LAUNCH_TIMED((trc_rcs_enc_kernel<2, 0, 1>), dim3(w.ngroups), dim3(64), RCS_ENC_FIXED + RCS_WAVE_LDS(2), s,
                           d_in, (u64)n, chunk, w.nchunks, tab, w.scratch, w.stride, w.scratch2, w.stride2, d_clen, w.gsum);
    } else RCS_ENC_LAUNCH(1, 1, w.scratch, w.stride);          // nstreams == -1: one stream, 32-bit range / 16-bit words (RCSM)
#undef RCS_ENC_LAUNCH
}

template <int NS, int GEO>
static void launch_dec(const uint8_t *d_payload, const uint32_t *d_clen, size_t n, uint32_t chunk,
                       const TrcWork &w, uint8_t *d_out, hipStream_t s)
{
    const u32 maxw = NS == 1 ? 14u : 7u;                       // 34 KiB tables + waves x (NS rings) must fit 160 KiB
    TRC_RAISE_LDS_ONCE((trc_rcs_dec_kernel<NS, GEO>), RCS_DEC_FIXED + maxw * RCS_WAVE_LDS(NS));
    u32 wpb = (w.ngroups + 255u) / 256u;                       // just enough waves per workgroup to give every CU one
    wpb = wpb < 1u ? 1u : wpb > maxw ? maxw : wpb;
    const size_t sm = RCS_DEC_FIXED + wpb * RCS_WAVE_LDS(NS);
    TRC_LAUNCH_TIMED((trc_rcs_dec_kernel<NS, GEO>), dim3((w.ngroups + wpb - 1) / wpb), dim3(64 * wpb), sm, s,
                       d_payload, d_clen, w.goff, w.gsum, (u64)n, chunk, w.nchunks, w.tables + TRC_TAB_LUT,
                       (const u32 *)(w.tables + TRC_TAB_DEC), d_out);
}
void trc_launch_rcs_dec(int nstreams, const uint8_t *d_payload, const uint32_t *d_clen, size_t n, uint32_t chunk,
                        const TrcWork &w, uint8_t *d_out, hipStream_t s)
{
    if (nstreams == 1)      launch_dec<1, 0>(d_payload, d_clen, n, chunk, w, d_out, s);
    else if (nstreams == 2 && rcs2_pair()) {
        const u32 nwaves = (w.nchunks + 31u) / 32u;
        TRC_RAISE_LDS_ONCE(trc_rcs2p_dec_kernel, RCS_DEC_FIXED + 14u * RCS_WAVE_LDS(1));
        u32 wpb = (nwaves + 255u) / 256u;
        wpb = wpb < 1u ? 1u : wpb > 14u ? 14u : wpb;
        TRC_LAUNCH_TIMED(trc_rcs2p_dec_kernel, dim3((nwaves + wpb - 1) / wpb), dim3(64 * wpb), RCS_DEC_FIXED + wpb * RCS_WAVE_LDS(1), s,
                           d_payload, d_clen, w.goff, w.gsum, (u64)n, chunk, w.nchunks, w.tables + TRC_TAB_LUT,
                           (const u32 *)(w.tables + TRC_TAB_DEC), d_out);
    }
    else if (nstreams == 2) launch_dec<2, 0>(d_payload, d_clen, n, chunk, w, d_out, s);
    else                    launch_dec<1, 1>(d_payload, d_clen, n, chunk, w, d_out, s);
}
