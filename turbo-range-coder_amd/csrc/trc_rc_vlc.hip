// trc_rc_vlc.hip -- Turbo-VLC integer coders over the adaptive CDF range coder:
//   rccdfuenc16/32, rccdfudec16/32     VN = 1 ("vlc6", 6-bit exponent)          rccdf.c:555-632   `turborc -e50`
//   rccdfvenc16/32, rccdfvdec16/32     VN = 2 ("vlc7", 7-bit exponent)          rccdf.c:392-430,473-513   `-e52`
//   rccdfvzenc16/32, rccdfvzdec16/32   VN = 2 on the zigzag of the delta to the previous element   rccdf.c:432-471,515-553   `-e53`
// (cdfe7/cdfd7, cdfe6/cdfd6 rccdf_.h:100-123; vlcenc/vlcdec/bitvrput/bitvrget include_/vlcbit.h:24-63; reverse bit I/O
// rcutil_.h:163-189; zigzag rcutil_.h:142-149.)  Per chunk the payload is exactly what the reference returns for that
// slice (chunks are multiples of the element size):
//     [u32 total][range-coder words][mantissa bytes]
// An element x >= 2^(VN+1) is split into an exponent symbol and f = bsr(x) - VN mantissa bits:
//     expo = ((f+1) << VN) + bits [f, f+VN) of x,   mantissa = low f bits of x.
// The value or the exponent is coded with two CDF16 tables: below T (8 for VN = 2, 12 for VN = 1) one symbol with
// table 0, else ((x-T)>>4)+T with table 0 and (x-T)&15 with table 1.  Mantissas go MSB-first into a bit string that
// the reference grows DOWN from the end of its output while the range coder grows up from out+4; it gives up (raw)
// as soon as the two come within 8 bytes of each other -- tested after every element with the bit side at
// out+inlen-8-floor(bits/8), monotone, so tested once per period here -- and on OVERFLOW of the total.
//
// One lane = one chunk.  Two tables per lane in LDS (80-byte rows, trc_nibmodel.h), nothing else: the chunk's region
// in scratch takes the range-coder words from its start (16-byte register window, trc_lane_io.h) and the bit string
// from its end (32 bits at a time: a little-endian u32 holding 32 MSB-first bits IS four bytes of the reversed
// string), and the gather joins the two pieces (lengths: aux[c] = 4 + range-coder bytes, the rest is bits).
#include "trc_rc.h"
#include "trc_lane_io.h"
#include "trc_nibmodel.h"
#include "trc_vlc.h"
#include "trc_launch.h"

template <int ES, int VN, bool ZZ>
__global__ __launch_bounds__(64) void trc_vlc_enc_kernel(
    const u8 *__restrict__ in, u64 n, u32 chunk, u32 nchunks,
    u8 *__restrict__ scratch, u32 stride, u32 *__restrict__ aux, u32 *__restrict__ clen, u32 *__restrict__ gsum)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    const u32 lane = threadIdx.x;
    NibModel<2> m; m.init(smem);
    constexpr u32 T = VN == 2 ? 8u : 12u, FIRST = 1u << (VN + 1), VM = (1u << VN) - 1u;

    WaveChunks wc;
    wc.c0 = blockIdx.x * 64u; wc.chunk = chunk; wc.nchunks = nchunks;
    wc.lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    wc.rows = nchunks - wc.c0 < 64u ? nchunks - wc.c0 : 64u;
    const bool alive = lane < wc.rows;
    const u32 c = wc.c0 + lane;
    const u32 len = alive ? wc.len_of(lane) : 0u;
    const int lim = trc_rc_limit(len);

    QuadIn qin; qin.base = in + (u64)wc.c0 * chunk;
    LaneOutDirect so; so.start(scratch + (u64)c * stride + 4u);
    LaneBitsDown bo; bo.start(scratch + (u64)(c + 1u) * stride);
    RcEncD e; e.start();
    u32 prev = 0;
    bool ovf = false;

    // one element: mantissa bits, then one or two range-coder symbols
    NibTable T0 = m.load(m.table(0)), T1 = T0;                 // both tables in registers (record_r, trc_nibmodel.h); all tables start alike
    auto put_elem = [&](u32 v, bool act) {
        u32 x = v;
        if (ZZ) { x = vlc_zigzag_enc(v - prev, ES == 4); prev = act ? v : prev; }
        const bool big = x >= FIRST;
        const u32 f = (31u - (u32)__clz((int)(x | 1u))) - (u32)VN;             // only meaningful when big
        const u32 expo = ((f + 1u) << VN) + ((x >> (f & 31u)) & VM);
        bo.put_if(act && big, f & 31u, x & ((1u << (f & 31u)) - 1u));
        const u32 xs = big ? expo : x;
        const bool two = xs >= T;
        const u32 y0 = two ? ((xs - T) >> 4) + T : xs, y1 = (xs - T) & 15u;
        const u32 r0 = m.record_r(T0, m.table(0), y0 & 15u);
        u32 r1 = 1u;
        if (act && two) r1 = m.record_r(T1, m.table(1), y1);    // table 1 adapts only where its symbol is coded
        e.sym_rec(act, r0 >> TRC_PROB_BITS, r0 & 0x7fffu);
        e.sym_rec(act && two, r1 >> TRC_PROB_BITS, r1 & 0x7fffu);
        e.flush(so);                                            // at most one word per two steps (trc_rc.h RcEncD)
    };

    const u32 S = chunk / TRC_SEG;
    qin.issue(wc, 0);
    for (u32 s = 0; s < S; s++) {
        qin.commit();
        if (s + 1 < S) qin.issue(wc, (s + 1) * TRC_SEG);
        uint4 pc0 = qin.read(0), pc1 = qin.read(1), pc2 = qin.read(2), pc3 = qin.read(3);
#pragma nounroll
        for (u32 k = 0; k < 4; k++) {
            uint4 v = pc0; pc0 = pc1; pc1 = pc2; pc2 = pc3;
            if (!__ballot(alive && !ovf && s * TRC_SEG + k * 16u < len)) continue;
#pragma nounroll
            for (u32 d = 0; d < 4; d++) {
                u32 w = v.x; v.x = v.y; v.y = v.z; v.z = v.w;
                const u32 q0 = s * TRC_SEG + k * 16u + d * 4u;
                const bool run = alive && !ovf;
                const u32 nb = len > q0 ? (len - q0 < 4u ? len - q0 : 4u) : 0u;   // a partial last element is zero-extended
                w = nb >= 4u ? w : (w & ((1u << (8u * nb)) - 1u));
                if (ES == 2) { put_elem(w & 0xffffu, run && q0 < len); put_elem(w >> 16, run && q0 + 2u < len); }
                else put_elem(w, run && q0 < len);
                // the reference's collision test, after every element there: range coder at out+4+words, bits at out+len-8-floor(bits/8)
                ovf = ovf || (run && q0 < len && (int)(4u + 4u * e.cw.nwords) + 8 >= (int)len - 8 - (int)(bo.total >> 3));
            }
        }
    }
    u32 out_len = 0, la = 0;
    if (alive) {
        if (!ovf) {
            e.finish(so);
            la = 4u + so.wpos;
            out_len = la + bo.bytes();
            if ((int)out_len >= lim) ovf = true;                // OVERFLOW on the total
        }
        if (ovf) out_len = len;
    }
    so.finish(alive && !ovf);
    bo.finish(alive && !ovf);
    if (alive && !ovf) { *(u32 *)(scratch + (u64)c * stride) = out_len; aux[2u * c] = la; }   // header: total
    if (alive) clen[c] = out_len;
    const u32 gs = trc_wave_sum(out_len);
    if (lane == 0) gsum[wc.c0 >> 6] = gs;
}

template <int ES, int VN, bool ZZ>
__global__ __launch_bounds__(64) void trc_vlc_dec_kernel(
    const u8 *__restrict__ payload, const u32 *__restrict__ clen, const u64 *__restrict__ goff, const u32 *__restrict__ gsum,
    u64 n, u32 chunk, u32 nchunks, u8 *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    const u32 lane = threadIdx.x;
    NibModel<2> m; m.init(smem);
    constexpr u32 T = VN == 2 ? 8u : 12u, FIRST = 1u << (VN + 1), VM = (1u << VN) - 1u;

    WaveChunks wc;
    wc.c0 = blockIdx.x * 64u; wc.chunk = chunk; wc.nchunks = nchunks;
    wc.lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    wc.rows = nchunks - wc.c0 < 64u ? nchunks - wc.c0 : 64u;
    const bool alive = lane < wc.rows;
    const u32 c = wc.c0 + lane;
    const u32 len = alive ? wc.len_of(lane) : 0u;
    const u32 cl = alive ? trc_min(clen[c], len) : 0u;        // a directory entry above the chunk length (corrupt input) reads as raw
    const u32 ex = trc_wave_incl_scan(cl) - cl;
    const u64 off = trc_group_base(goff, gsum, wc.c0 >> 6) + ex;
    const bool coded = alive && cl != len && cl >= 8u;         // a coded chunk has at least its header and one flushed range-coder word

    LaneIn<4> si; si.prime(payload + off + 4u, coded, trc_sub_sat(cl, 4u));
    RcDec dc;
    { const u32 a = si.peek32(); si.skip_if(coded); const u32 b = si.peek32(); si.skip_if(coded); dc.start(a, b); }
    const u8 *bend = payload + off + cl;                       // the bit string is read downward from here
    u32 bpos = 0, prev = 0;

    // both tables live in registers for the whole chunk (a decoder's table loads and stores are dependent LDS round trips that
    // one wave per SIMD cannot hide; LDS keeps only the K rows of the update)
    NibTable T0 = m.load(m.table(0)), T1 = T0;                 // all tables start alike
    auto get = [&](NibTable &Tb, bool act) -> u32 {
        u32 c0, c1;
        const u32 x = trc_nib_search(Tb, dc.scaled(), c0, c1);
        dc.consume_if(si, act, c0, c1);
        m.adapt(Tb, x);
        return x;
    };
    auto get_elem = [&](bool act) -> u32 {
        u32 x = get(T0, act);
        if (act && x >= T) { const u32 z = get(T1, true); x = ((x - T) << 4 | z) + T; }
        if (act && x >= FIRST) {
            u32 f = (x >> VN) - 1u;
            f = f > 30u ? 30u : f;                             // (corrupt input)
            const u32 byteoff = trc_min(bpos >> 3, cl - 8u);   // 8-byte window ending at bend - byteoff, never below the chunk's payload
            const u64 bw = *(const u64_a1 *)(bend - 8u - byteoff);
            const u32 ma = (u32)((bw << (bpos & 7u)) >> (64u - f));
            bpos += f;
            x = (((1u << VN) + (x & VM)) << f) + ma;
        }
        if (ZZ) {
            x = prev + vlc_zigzag_dec(x);
            if (ES == 2) x &= 0xffffu;
            prev = act ? x : prev;
        }
        return x;
    };

    QuadOut qout; qout.base = out + (u64)wc.c0 * chunk;
    u8 *dst = out + (u64)c * chunk;
    const u32 S = chunk / TRC_SEG;
    for (u32 s = 0; s < S; s++) {
        uint4 pc0 = make_uint4(0, 0, 0, 0), pc1 = pc0, pc2 = pc0, pc3 = pc0;
#pragma nounroll
        for (u32 k = 0; k < 4; k++) {
            const u32 p0 = s * TRC_SEG + k * 16u;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (__ballot(coded && p0 < len)) {
#pragma nounroll
                for (u32 d = 0; d < 4; d++) {
                    const u32 q0 = p0 + d * 4u;
                    u32 w;
                    if (ES == 2) { const u32 a = get_elem(coded && q0 < len) & 0xffffu, b = get_elem(coded && q0 + 2u < len); w = a | b << 16; }
                    else w = get_elem(coded && q0 < len);
                    v.x = v.y; v.y = v.z; v.z = v.w; v.w = w;
                }
                if (coded && p0 < len && p0 + 16u > len) {      // ragged end of the last chunk: byte stores
                    const u32 ww[4] = { v.x, v.y, v.z, v.w };
                    for (u32 pos = p0; pos < len; pos++) dst[pos] = (u8)(ww[(pos - p0) >> 2] >> (8 * ((pos - p0) & 3u)));
                }
            }
            pc0 = pc1; pc1 = pc2; pc2 = pc3; pc3 = v;
        }
        qout.put(0, pc0); qout.put(1, pc1); qout.put(2, pc2); qout.put(3, pc3);
        qout.flush(wc, s * TRC_SEG);
    }
    trc_wave_copy_raw(__ballot(alive && cl == len && len != 0), off, len, out + (u64)wc.c0 * chunk, chunk, payload);
}

// ------------------------------------------------------------------------------------- launch ---
template <int ES, int VN, bool ZZ>
static void launch_vlc_enc(const uint8_t *d_in, size_t n, uint32_t chunk, const TrcWork &w, uint32_t *d_clen, hipStream_t s)
{
    TRC_LAUNCH_TIMED((trc_vlc_enc_kernel<ES, VN, ZZ>), dim3(w.ngroups), dim3(64), TRC_NIB2_BYTES, s,
                     d_in, (u64)n, chunk, w.nchunks, w.scratch, w.stride, w.aux, d_clen, w.gsum);
}
template <int ES, int VN, bool ZZ>
static void launch_vlc_dec(const uint8_t *d_payload, const uint32_t *d_clen, size_t n, uint32_t chunk,
                           const TrcWork &w, uint8_t *d_out, hipStream_t s)
{
    TRC_LAUNCH_TIMED((trc_vlc_dec_kernel<ES, VN, ZZ>), dim3(w.ngroups), dim3(64), TRC_NIB2_BYTES, s,
                     d_payload, d_clen, w.goff, w.gsum, (u64)n, chunk, w.nchunks, d_out);
}
// variant: 0 = u (vlc6), 1 = v (vlc7), 2 = vz (vlc7 on zigzag deltas); elem = 2 or 4 bytes
void trc_launch_vlc_enc(int variant, int elem, const uint8_t *d_in, size_t n, uint32_t chunk, const TrcWork &w, uint32_t *d_clen, hipStream_t s)
{
    if (elem == 2) { if (variant == 0) launch_vlc_enc<2, 1, false>(d_in, n, chunk, w, d_clen, s); else if (variant == 1) launch_vlc_enc<2, 2, false>(d_in, n, chunk, w, d_clen, s); else launch_vlc_enc<2, 2, true>(d_in, n, chunk, w, d_clen, s); }
    else           { if (variant == 0) launch_vlc_enc<4, 1, false>(d_in, n, chunk, w, d_clen, s); else if (variant == 1) launch_vlc_enc<4, 2, false>(d_in, n, chunk, w, d_clen, s); else launch_vlc_enc<4, 2, true>(d_in, n, chunk, w, d_clen, s); }
}
void trc_launch_vlc_dec(int variant, int elem, const uint8_t *d_payload, const uint32_t *d_clen, size_t n, uint32_t chunk,
                        const TrcWork &w, uint8_t *d_out, hipStream_t s)
{
    if (elem == 2) { if (variant == 0) launch_vlc_dec<2, 1, false>(d_payload, d_clen, n, chunk, w, d_out, s); else if (variant == 1) launch_vlc_dec<2, 2, false>(d_payload, d_clen, n, chunk, w, d_out, s); else launch_vlc_dec<2, 2, true>(d_payload, d_clen, n, chunk, w, d_out, s); }
    else           { if (variant == 0) launch_vlc_dec<4, 1, false>(d_payload, d_clen, n, chunk, w, d_out, s); else if (variant == 1) launch_vlc_dec<4, 2, false>(d_payload, d_clen, n, chunk, w, d_out, s); else launch_vlc_dec<4, 2, true>(d_payload, d_clen, n, chunk, w, d_out, s); }
}
