// trc_rc_vnib.hip -- the "vnibble" adaptive-CDF range coders:
//   TRC_RCV8   rccdfenc8 / rccdfdec8     rccdf.c:326-352   `turborc -e48`   one stream
//   TRC_RCVI8  rccdfienc8 / rccdfidec8   rccdf.c:355-390   `turborc -e49`   two streams
// Symbol split cdfe8 / cdfd8 (rccdf_.h:76-98): a byte x becomes one to three CDF16 symbols on three adaptive tables
//     x < 13        : m0 <- x
//     13 <= x < 45  : m0 <- 13 + ((x-13) >> 4),   m1 <- (x-13) & 15
//     x >= 45       : m0 <- 15,   m1 <- (x-45) >> 4,   m2 <- (x-45) & 15
// -- a byte-wise variable-length code for data that is mostly small values.  One-stream form: everything on one range
// coder, OVERFLOW (rcutil_.h:130) after every byte.  Two-stream form: the m0 and m2 symbols on stream 0, the m1 symbols on
// stream 1 whose base is out+4+inlen*37/64; OVERFLOWI (rccdf.c:46) after every full group of 4 bytes, payload
// [u32 len0][stream 0][stream 1], OVERFLOW on the total.  Where the reference lets the tail of stream 0 run into stream 1
// (its result cannot be decoded: oracle/trc_oracle.c) the chunk is stored raw.
// Per chunk the payload is exactly what the reference function returns for that slice.  Same structure as the other
// model-bound coders (trc_rc_adaptive.hip): one lane = one chunk, a period is 4 input bytes -- first the model walks them
// and leaves {cdf_lo, freq} records in registers, then the range coder(s) consume the records with predicated steps.
// The model is three tables, 112 B per lane (encoder: LDS; decoder: registers for the whole chunk), so occupancy is not
// bound by LDS here.
#include "trc_rc.h"
#include "trc_lane_io.h"
#include "trc_nibmodel.h"
#include "trc_launch.h"

template <int NS>
__global__ __launch_bounds__(64) void trc_rcv_enc_kernel(
    const u8 *__restrict__ in, u64 n, u32 chunk, u32 nchunks,
    u8 *__restrict__ scratch, u32 stride, u8 *__restrict__ scratch2, u32 stride2,
    u32 *__restrict__ clen, u32 *__restrict__ gsum)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    const u32 lane = threadIdx.x;
    NibModel<3> m; m.init(smem);

    WaveChunks wc;
    wc.c0 = blockIdx.x * 64u; wc.chunk = chunk; wc.nchunks = nchunks;
    wc.lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    wc.rows = nchunks - wc.c0 < 64u ? nchunks - wc.c0 : 64u;
    const bool alive = lane < wc.rows;
    const u32 c = wc.c0 + lane;
    const u32 len = alive ? wc.len_of(lane) : 0u;
    const int lim = trc_rc_limit(len);
    const u32 off1 = 4u + (u32)(((u64)len * 37u) / 64u);       // stream-1 base inside `out` (rccdf.c:374)

    QuadIn qin; qin.base = in + (u64)wc.c0 * chunk;
    LaneOutDirect o0, o1;
    o0.start(scratch + (u64)c * stride + (NS == 2 ? 4u : 0u));
    o1.start(NS == 2 ? scratch2 + (u64)c * stride2 : scratch);
    RcEncD e0, e1; e0.start(); e1.start();
    bool ovf = alive && NS == 1 && lim <= 0;

    NibTable T0 = m.load(m.table(0)), T1 = T0, T2 = T0;        // the three tables in registers (record_r, trc_nibmodel.h); all tables start alike
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
                const u32 w = v.x; v.x = v.y; v.y = v.z; v.z = v.w;
                const u32 q0 = s * TRC_SEG + k * 16u + d * 4u;
                const bool run = alive && !ovf;
                // ---- model: 4 bytes -> up to 3 records each (no coder state involved); a record of 0 = symbol absent
                u32 ra[4], rb[4], rc[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const u32 x = (w >> (8 * i)) & 255u;
                    const bool two = x >= 13u, three = x >= 45u;
                    const u32 y = three ? x - 45u : x - 13u;            // (unused when x < 13)
                    const u32 a = three ? 15u : two ? 13u + (y >> 4) : x;
                    ra[i] = m.record_r(T0, m.table(0), a);
                    rb[i] = m.record_r_if(two, T1, m.table(1), three ? y >> 4 : y & 15u);
                    rc[i] = m.record_r_if(three, T2, m.table(2), y & 15u);
                }
                // ---- range coder(s): predicated steps, in the reference's order m0, m1, m2
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const bool act = run && q0 + (u32)i < len;
                    // (an encoder's remembered word goes out after every second step: trc_rc.h RcEncD)
                    e0.sym_rec(act, ra[i] >> TRC_PROB_BITS, ra[i] & 0x7fffu);
                    if (NS == 1) {
                        if (i & 1) e0.flush(o0);                               // steps 3i, 3i+1, 3i+2 of the period: flush after the odd ones
                        e0.sym_rec(act && rb[i] != 0u, rb[i] >> TRC_PROB_BITS, rb[i] & 0x7fffu);
                        if (!(i & 1)) e0.flush(o0);
                    } else e1.sym_rec(act && rb[i] != 0u, rb[i] >> TRC_PROB_BITS, rb[i] & 0x7fffu);
                    e0.sym_rec(act && rc[i] != 0u, rc[i] >> TRC_PROB_BITS, rc[i] & 0x7fffu);
                    if (NS == 1) { if (i & 1) e0.flush(o0); }
                    else { e0.flush(o0); if (i & 1) e1.flush(o1); }
                }
                // ---- incompressibility tests (monotone in the word counts)
                if (NS == 1) ovf = ovf || (run && q0 < len && (int)(4u * e0.cw.nwords) >= lim);
                else ovf = ovf || (run && q0 + 4u <= len &&
                                   ((int)(off1 + 4u * e1.cw.nwords) >= lim || 4u + 4u * e0.cw.nwords >= off1));
            }
        }
    }
    u32 out_len = 0;
    if (alive) {
        if (!ovf) {
            e0.finish(o0);
            if (NS == 2) {
                e1.finish(o1);
                out_len = 4u + o0.wpos + o1.wpos;
                if ((int)out_len >= lim || 4u + o0.wpos > off1) ovf = true;      // total, and stream 0 running into stream 1
            } else out_len = o0.wpos;
        }
        if (ovf) out_len = len;
    }
    o0.finish(alive && !ovf);
    if (NS == 2) {
        o1.finish(alive && !ovf);
        if (alive && !ovf) *(u32 *)(scratch + (u64)c * stride) = o0.wpos;          // header: len0
    }
    if (alive) clen[c] = out_len;
    const u32 gs = trc_wave_sum(out_len);
    if (lane == 0) gsum[wc.c0 >> 6] = gs;
}

template <int NS>
__global__ __launch_bounds__(64) void trc_rcv_dec_kernel(
    const u8 *__restrict__ payload, const u32 *__restrict__ clen, const u64 *__restrict__ goff, const u32 *__restrict__ gsum,
    u64 n, u32 chunk, u32 nchunks, u8 *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    const u32 lane = threadIdx.x;
    NibModel<3> m; m.init(smem);                               // (only the K table of it is used: the three tables live in registers)

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
    const bool coded = alive && cl != len;

    LaneIn<4> s0, s1;
    const u32 len0 = (NS == 2 && coded) ? trc_min(trc_ld32_a2(payload + off), trc_sub_sat(cl, 4u)) : 0u;   // a corrupt header cannot point outside the chunk's payload
    s0.prime(payload + off + (NS == 2 ? 4u : 0u), coded, NS == 2 ? len0 : cl);
    s1.prime(payload + off + 4u + len0, NS == 2 && coded, trc_sub_sat(cl, 4u + len0));
    RcDec d0, d1;
    { const u32 a = s0.peek32(); s0.skip_if(coded); const u32 b = s0.peek32(); s0.skip_if(coded); d0.start(a, b); }
    { const u32 a = s1.peek32(); s1.skip_if(NS == 2 && coded); const u32 b = s1.peek32(); s1.skip_if(NS == 2 && coded); d1.start(a, b); }

    NibTable T0 = m.load(m.table(0)), T1 = T0, T2 = T0;        // all tables start alike
    // one symbol where `on`: search the register table, consume, adapt (nothing moves where !on)
    auto get = [&](RcDec &dq, LaneIn<4> &sq, NibTable &T, bool on) -> u32 {
        u32 c0, c1;
        const u32 x = trc_nib_search(T, dq.scaled(), c0, c1);
        dq.consume_if(sq, on, c0, c1);
        if (on) m.adapt(T, x);
        return x;
    };
    // the same against a look-ahead word; bit 4 of the result: renormalised
    auto getw = [&](RcDec &dq, u32 w, NibTable &T, bool on) -> u32 {
        u32 c0, c1;
        const u32 x = trc_nib_search(T, dq.scaled(), c0, c1);
        const bool rn = dq.consume_w(on, c0, c1, w);
        if (on) m.adapt(T, x);
        return x | (rn ? 16u : 0u);
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
                    u32 w = 0;
#pragma nounroll
                    for (u32 i = 0; i < 4; i++) {
                        const bool act = coded && q0 + i < len;
                        u32 a, b, cc;
                        if (NS == 1) {
                            // three steps on one stream: two consecutive ones cannot both renormalise (trc_rc.h), so the byte
                            // takes at most two words -- both looked at up front, the stream advanced once, its next window
                            // prefetched (trc_lane_io.h)
                            const uint4 pre = s0.prefetch();
                            u32 w0, w1; s0.two_words(w0, w1);
                            const u32 ra = getw(d0, w0, T0, act);
                            a = ra & 15u;
                            const bool two_ = act && a >= 13u, three_ = act && a == 15u;
                            const u32 rb = getw(d0, w0, T1, two_);
                            const u32 took = (ra | rb) & 16u;
                            const u32 rc = getw(d0, took ? w1 : w0, T2, three_);
                            b = rb & 15u; cc = rc & 15u;
                            s0.advance_pre((took >> 2) + ((rc & 16u) >> 2), pre);
                        } else {
                            a = get(d0, s0, T0, act);
                            b = get(d1, s1, T1, act && a >= 13u);
                            cc = get(d0, s0, T2, act && a == 15u);
                        }
                        const bool two = act && a >= 13u, three = act && a == 15u;
                        const u32 x = three ? ((b << 4) | cc) + 45u : two ? (((a - 13u) << 4) | b) + 13u : a;
                        w |= (x & 255u) << (8 * i);
                    }
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

void trc_launch_rcv_enc(int nstreams, const uint8_t *d_in, size_t n, uint32_t chunk, const TrcWork &w, uint32_t *d_clen, hipStream_t s)
{
    if (nstreams == 2)
        TRC_LAUNCH_TIMED((trc_rcv_enc_kernel<2>), dim3(w.ngroups), dim3(64), TRC_NIB3_BYTES, s,
                         d_in, (u64)n, chunk, w.nchunks, w.scratch, w.stride, w.scratch2, w.stride2, d_clen, w.gsum);
    else
        TRC_LAUNCH_TIMED((trc_rcv_enc_kernel<1>), dim3(w.ngroups), dim3(64), TRC_NIB3_BYTES, s,
                         d_in, (u64)n, chunk, w.nchunks, w.scratch, w.stride, w.scratch2, w.stride2, d_clen, w.gsum);
}
void trc_launch_rcv_dec(int nstreams, const uint8_t *d_payload, const uint32_t *d_clen, size_t n, uint32_t chunk,
                        const TrcWork &w, uint8_t *d_out, hipStream_t s)
{
    if (nstreams == 2)
        TRC_LAUNCH_TIMED((trc_rcv_dec_kernel<2>), dim3(w.ngroups), dim3(64), TRC_NIB3_BYTES, s,
                         d_payload, d_clen, w.goff, w.gsum, (u64)n, chunk, w.nchunks, d_out);
    else
        TRC_LAUNCH_TIMED((trc_rcv_dec_kernel<1>), dim3(w.ngroups), dim3(64), TRC_NIB3_BYTES, s,
                         d_payload, d_clen, w.goff, w.gsum, (u64)n, chunk, w.nchunks, d_out);
}
