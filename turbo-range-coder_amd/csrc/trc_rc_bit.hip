// trc_rc_bit.hip -- bitwise order-0 range coder with the "s" predictor (codec TRC_RCB).
//
// Per chunk the payload is exactly what rcsenc returns for that slice (reference rc_.c:47-58;
// mb8enc/mb8dec mb_o0.h:89-112; rcbe_/rcbd_ turborc_.h:417-452; predictor mbc_s.h:29-37,53-55;
// geometry rc_s.c:31-33 = 64-bit range, 32-bit I/O, 15-bit probabilities):
//   255-node binary tree of 16-bit probabilities P(bit=1)*2^15, all 0x4000 at chunk start;
//   per byte, MSB first: cut = (range>>15)*p; bit 1 -> range = cut; bit 0 -> low += cut, range -= cut;
//   p -= ((p - (bit<<15)) >> 5) + bit (32-bit unsigned arithmetic, low 16 bits kept);
//   renormalisation ONLY before bits 7,5,3,1 (the reference's _RCENORM1 is empty, _RCENORM2 live).
//
// One lane = one chunk = one range-coder state.  The lane's 256 x u16 model lives in LDS in a [context][lane] layout
// (2 lanes per bank, independent of the context each lane is at), 32 KiB per wave, and it is the ONLY thing in LDS:
// chunk bytes move through in-register quad transposes (trc_io.h QuadIn/QuadOut), the coded stream through a 16-byte
// register window per lane (trc_lane_io.h), so five waves share a CU.  What bounds the coder is the latency of the
// model accesses, eight per byte:
//   encoder  the eight nodes a byte visits are known from the byte itself ((0x100|x) >> (8-k)) and are all different,
//            so their probabilities are read in one batch, the eight coding steps run on registers (predicated renorm,
//            no branches), and the eight updated probabilities are written back in one batch;
//   decoder  the path depends on the decoded bits; both children of the current node are requested before the bit is
//            resolved, so the next probability is already on its way while the current step computes.
#include "trc_rc.h"
#include "trc_lane_io.h"
#include "trc_launch.h"

#define RCB_MODEL_BYTES (256u * 64u * 2u)                  // [ctx][lane] u16
#define RCB_WAVE_LDS    RCB_MODEL_BYTES

__device__ __forceinline__ u32 rcb_adapt(u32 p, u32 bit) { return (p - (((p - (bit << TRC_PROB_BITS)) >> 5) + bit)) & 0xffffu; }

__global__ __launch_bounds__(64) void trc_rcb_enc_kernel(
    const u8 *__restrict__ in, u64 n, u32 chunk, u32 nchunks,
    u8 *__restrict__ scratch, u32 stride, u32 *__restrict__ clen, u32 *__restrict__ gsum)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    const u32 lane = threadIdx.x;
    u16 *mb = (u16 *)smem + lane;                              // mb[ctx * 64]
    for (u32 i = 0; i < 256; i++) mb[i * 64] = (u16)(TRC_PROB_ONE >> 1);

    WaveChunks wc;
    wc.c0 = blockIdx.x * 64u; wc.chunk = chunk; wc.nchunks = nchunks;
    wc.lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    wc.rows = nchunks - wc.c0 < 64u ? nchunks - wc.c0 : 64u;
    const bool alive = lane < wc.rows;
    const u32 c = wc.c0 + lane;
    const u32 len = alive ? wc.len_of(lane) : 0u;
    const int lim = trc_rc_limit(len);

    QuadIn qin; qin.base = in + (u64)wc.c0 * chunk;
    LaneOut32 so; so.start(scratch + (u64)c * stride);
    RcEnc e; e.start();
    bool ovf = alive && lim <= 0;
    // `live`: this lane is still coding.  A lane that is not (dead lane, incompressible chunk, past the end of a short
    // last chunk) keeps running the same arithmetic on its own registers and its own model column -- nothing of it is
    // observable, because only live lanes emit words -- so the eight steps of a byte carry no per-lane predication at
    // all.  A lane whose chunk ends before the wave's does (the input's last chunk) flushes at that byte boundary.
    bool live = alive && !ovf;
    u32 out_len = alive ? len : 0u;                            // raw until proven otherwise

    // one byte.  The body is a 4-trip loop over bit pairs (renorm + two steps) so that the kernel stays small in the
    // instruction cache.  A renormalisation SHIFTS the state at once, but the word it produces is only remembered
    // (`pend`): a lane emits a word every ~6 bytes, and the emit logic (held-back word, carry, 16-byte register window,
    // its store) is by far the largest block of the loop, so it runs ONCE per byte -- for the one word almost every
    // lane has at most -- instead of at each of the four renormalisation points.  A second word inside one byte (>= 32
    // bits of range spent on <= 6 bits) first flushes the remembered one, in order, behind a wave-uniform branch.
    const u32 mcol = trc_lds_addr(smem) + lane * 2u;           // this lane's model column as an LDS byte address
    auto put_byte = [&](u32 x) {
        const u32 path = (0x100u | x) << 7;                    // node k of the byte's path, times the row stride: (path >> (8-k)) & ~127
        u32 pp0, pp1, pp2, pp3;                                // the eight probabilities, two per register
        {
            const u32 a0 = trc_ldsr16(mcol + 128u), a1 = trc_ldsr16(((path >> 7) & ~127u) + mcol);
            const u32 a2 = trc_ldsr16(((path >> 6) & ~127u) + mcol), a3 = trc_ldsr16(((path >> 5) & ~127u) + mcol);
            const u32 a4 = trc_ldsr16(((path >> 4) & ~127u) + mcol), a5 = trc_ldsr16(((path >> 3) & ~127u) + mcol);
            const u32 a6 = trc_ldsr16(((path >> 2) & ~127u) + mcol), a7 = trc_ldsr16(((path >> 1) & ~127u) + mcol);
            pp0 = a0 | a1 << 16; pp1 = a2 | a3 << 16; pp2 = a4 | a5 << 16; pp3 = a6 | a7 << 16;
        }
        u32 xs = x << 24, node = 1;
        bool pend = false, pcy = false;
        u32 pw = 0;
#pragma nounroll
        for (u32 j = 0; j < 4; j++) {
            {                                                  // renorm before bits 7,5,3,1 only (_RCENORM2)
                const bool rn = e.range < TRC_TOP32;
                if (__ballot(rn && pend)) {                    // second word within this byte (rare): the first one goes out now
                    e.cw.emit_if(so, rn && pend && live, pcy, pw);
                    pend = pend && !rn;
                }
                pcy = rn ? e.mark > e.low : pcy;
                pw = rn ? (u32)(e.low >> 32) : pw;
                pend = pend || rn;
                e.low = rn ? e.low << 32 : e.low;
                e.range = rn ? e.range << 32 : e.range;
                e.mark = rn ? e.low : e.mark;
            }
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const u32 p = h ? pp0 >> 16 : pp0 & 0xffffu;
                const u32 bit = xs >> 31; xs <<= 1;
                const u64 cut = (e.range >> TRC_PROB_BITS) * p;                  // rcbe_
                e.low += bit ? 0 : cut;
                e.range = bit ? cut : e.range - cut;
                trc_ldsw16((node << 7) + mcol, rcb_adapt(p, bit));
                node = node * 2 + bit;
            }
            pp0 = pp1; pp1 = pp2; pp2 = pp3;
        }
        e.cw.emit_if(so, pend && live, pcy, pw);
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
            if (!__ballot(live)) continue;
#pragma nounroll
            for (u32 d = 0; d < 4; d++) {
                const u32 w = v.x; v.x = v.y; v.y = v.z; v.z = v.w;
                const u32 q0 = s * TRC_SEG + k * 16u + d * 4u;
#pragma nounroll
                for (u32 i = 0; i < 4; i++) {
                    if (__ballot(live && q0 + i == len)) {     // a short last chunk ends here: decide and flush it now (once per grid)
                        if (live && q0 + i == len) {
                            if ((int)(4u * e.cw.nwords) < lim) { e.finish(so); out_len = so.wpos; so.finish(true); }
                            live = false;                      // (else: incompressible, out_len stays the raw length)
                        }
                    }
                    put_byte((w >> (8 * i)) & 255u);
                }
                ovf = ovf || (live && (int)(4u * e.cw.nwords) >= lim);           // OVERFLOW per byte, monotone
                live = live && !ovf;
            }
        }
    }
    if (live) { e.finish(so); out_len = so.wpos; }
    so.finish(live);
    if (alive) clen[c] = out_len;
    const u32 gs = trc_wave_sum(out_len);
    if (lane == 0) gsum[wc.c0 >> 6] = gs;
}

__global__ __launch_bounds__(64) void trc_rcb_dec_kernel(
    const u8 *__restrict__ payload, const u32 *__restrict__ clen, const u64 *__restrict__ goff, const u32 *__restrict__ gsum,
    u64 n, u32 chunk, u32 nchunks, u8 *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    const u32 lane = threadIdx.x;
    u16 *mb = (u16 *)smem + lane;
    for (u32 i = 0; i < 256; i++) mb[i * 64] = (u16)(TRC_PROB_ONE >> 1);

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

    LaneIn<4> si; si.prime(payload + off, coded, cl);
    RcDec dc;
    { const u32 a = si.peek32(); si.skip_if(coded); const u32 b = si.peek32(); si.skip_if(coded); dc.start(a, b); }

    const u32 mcol = trc_lds_addr(smem) + lane * 2u;           // this lane's model column as an LDS byte address
    auto get_byte = [&](bool act) -> u32 {
        u32 ctx = 1;
        u32 p = trc_ldsr16(mcol + 128u);
#pragma nounroll
        for (u32 j = 0; j < 4; j++) {
            {                                                  // renorm before bits 7,5,3,1 only
                const bool rn = act && dc.range < TRC_TOP32;
                const u32 w = si.peek32();
                dc.range = rn ? dc.range << 32 : dc.range;
                dc.code = rn ? (dc.code << 32) | w : dc.code;
                si.skip_if(rn);
            }
#pragma unroll
            for (int h = 0; h < 2; h++) {
                // both children are requested before this bit is known (below the last level the index wraps into the
                // model and the values are unused)
                const u32 lc = ((ctx << 8) & 0x7f00u) + mcol;  // row 2*ctx (mod 256), 128 bytes per row
                const u32 pl = trc_ldsr16(lc), pr = trc_ldsr16(lc + 128u);
                const u64 cut = (dc.range >> TRC_PROB_BITS) * p;
                const bool one = dc.code < cut;                // rcbd_
                dc.range = one ? cut : dc.range - cut;        // (lanes that are not decoding run along on their own registers
                dc.code = one ? dc.code : dc.code - cut;       //  and model column: only `act` lanes consume stream words)
                trc_ldsw16((ctx << 7) + mcol, rcb_adapt(p, one ? 1u : 0u));
                ctx = ctx * 2 + (one ? 1u : 0u);
                p = one ? pr : pl;
            }
        }
        return ctx & 255u;
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
                    for (u32 i = 0; i < 4; i++) w |= get_byte(coded && q0 + i < len) << (8 * i);
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
    u64 rawmask = __ballot(alive && cl == len && len != 0);
    while (rawmask) {
        const int k = __ffsll((long long)rawmask) - 1;
        rawmask &= rawmask - 1;
        const u32 olo = (u32)__shfl((int)(u32)off, k, 64), ohi = (u32)__shfl((int)(u32)(off >> 32), k, 64);
        const u32 l = (u32)__shfl((int)len, k, 64);
        trc_wave_copy(out + (u64)(wc.c0 + (u32)k) * chunk, payload + (((u64)ohi << 32) | olo), l);
    }
}

void trc_launch_rcb_enc(const uint8_t *d_in, size_t n, uint32_t chunk, const TrcWork &w, uint32_t *d_clen, hipStream_t s)
{
    TRC_RAISE_LDS_ONCE(trc_rcb_enc_kernel, RCB_WAVE_LDS);
    TRC_LAUNCH_TIMED(trc_rcb_enc_kernel, dim3(w.ngroups), dim3(64), RCB_WAVE_LDS, s,
                       d_in, (u64)n, chunk, w.nchunks, w.scratch, w.stride, d_clen, w.gsum);
}
void trc_launch_rcb_dec(const uint8_t *d_payload, const uint32_t *d_clen, size_t n, uint32_t chunk,
                        const TrcWork &w, uint8_t *d_out, hipStream_t s)
{
    TRC_RAISE_LDS_ONCE(trc_rcb_dec_kernel, RCB_WAVE_LDS);
    TRC_LAUNCH_TIMED(trc_rcb_dec_kernel, dim3(w.ngroups), dim3(64), RCB_WAVE_LDS, s,
                       d_payload, d_clen, w.goff, w.gsum, (u64)n, chunk, w.nchunks, d_out);
}
