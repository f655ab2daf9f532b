// trc_ans_adaptive.hip -- adaptive-CDF byte rANS, 4 interleaved states (codec TRC_ANSA; `turborc -e56/57/58`).
//
// Per chunk the payload is exactly what anscdfenc returns for that slice (a chunk is far below the
// reference's 4 MiB block, so it is one block): reference anscdf.c:567-586 (encoder), :588-605
// (decoder); mnenc8x2/mnflush/mndec8x2 anscdf_.h:114-119,128-138,152-162; model trc_nibmodel.h.
//     [u32 st3][u32 st2][u32 st1][u32 st0][u16 renorm words in decode order]        (raw if it does not fit)
//
// The encoder is inherently two-pass (the model adapts forward, rANS codes backward):
//   pass 1  trc_ansa_model_kernel : walk the chunk forward through the adaptive model and record
//           {cdf_lo << 15 | freq} per nibble -- 4 records per byte pair in the reference's push
//           order (x0.hi -> state 3, x0.lo -> 2, x1.hi -> 1, x1.lo -> 0; an odd tail byte pairs
//           with a CODED dummy 0).  Records stream to HBM scratch as uniform 64-byte segments
//           (8 B per input byte: the reference keeps the same stack on the heap, anscdf.c:570).
//   pass 2  trc_ansa_code_kernel  : pop the records in reverse, one rANS step each on state
//           3-(r&3); words grow downward from the end of the chunk's scratch region.  The divisor
//           changes every symbol, so st/f is an f32 estimate plus exact correction (st < 2^31).
//   Raw rule (mnflush): before EVERY record the reference tests ep <= op + 2 + 16; the test is
//   monotone, so "true before the last record" decides, which is what pass 2 evaluates.
#include "trc_io.h"
#include "trc_nibmodel.h"
#include "trc_launch.h"

#define ANSA_MODEL_LDS (TRC_NIB_BYTES + 2u * TRC_TILE_BYTES)
#define ANSA_CODE_LDS  (TRC_TILE_BYTES + TRC_SRING_BYTES + TRC_SEL_BYTES)
#define ANSA_DEC_LDS   (TRC_NIB_BYTES + TRC_TILE_BYTES + TRC_SRING_BYTES + TRC_SEL_BYTES)

// ------------------------------------------------------------------------------ encode, pass 1 ---
__global__ __launch_bounds__(64) void trc_ansa_model_kernel(
    const u8 *__restrict__ in, u64 n, u32 chunk, u32 nchunks, u8 *__restrict__ recs /* 8*chunk bytes per chunk */)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    const u32 lane = threadIdx.x;
    NibModel m; m.row = smem + lane * TRC_NIB_ROW; m.reset();

    WaveChunks wc;
    wc.c0 = blockIdx.x * 64u; wc.chunk = chunk; wc.nchunks = nchunks;
    wc.lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    wc.rows = nchunks - wc.c0 < 64u ? nchunks - wc.c0 : 64u;
    WaveChunks wr = wc;                                        // the same chunks in record space
    wr.chunk = 8u * chunk; wr.lastlen = 8u * (wc.lastlen + (wc.lastlen & 1u));
    const bool alive = lane < wc.rows;
    const u32 len = alive ? wc.len_of(lane) : 0u;
    const u32 plen = len + (len & 1u);                         // bytes coded, dummy included

    TileIn tin; tin.tile = smem + TRC_NIB_BYTES; tin.base = in + (u64)wc.c0 * chunk;
    TileOut tout; tout.tile = smem + TRC_NIB_BYTES + TRC_TILE_BYTES; tout.base = recs + (u64)wc.c0 * wr.chunk;

    auto rec_nibble = [&](u8 *tb, u32 x) -> u32 {
        u32 c0, c1; m.bounds(tb, x, c0, c1);
        NibTable T = m.load(tb); trc_nib_adapt(T, c0); m.store(tb, T);
        return (c0 << 15) | (c1 - c0);
    };

    const u32 S = chunk / TRC_SEG;
    tin.issue(wc, 0);
    for (u32 s = 0; s < S; s++) {
        tin.commit();
        if (s + 1 < S) tin.issue(wc, (s + 1) * TRC_SEG);
        for (u32 k = 0; k < 4; k++) {
            const uint4 v = tin.read(k);
            const u32 w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
            for (int h = 0; h < 2; h++) {                      // 8 input bytes -> 16 records = one 64-byte record segment
                const u32 q0 = s * TRC_SEG + k * 16u + (u32)h * 8u;
                u32 r[16];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    u32 x = (w[2 * h + (i >> 2)] >> (8 * (i & 3))) & 255u;
                    const u32 pos = q0 + (u32)i;
                    if (pos >= len) x = 0;                     // the dummy (and never-used padding)
                    if (alive && pos < plen) {
                        r[2 * i] = rec_nibble(m.table(0), x >> 4);
                        r[2 * i + 1] = rec_nibble(m.table(1u + (x >> 4)), x & 15u);
                    } else r[2 * i] = r[2 * i + 1] = 0;
                }
#pragma unroll
                for (int j = 0; j < 4; j++) tout.put((u32)j, make_uint4(r[4 * j], r[4 * j + 1], r[4 * j + 2], r[4 * j + 3]));
                tout.flush(wr, q0 * 8u);
            }
        }
    }
}

// ------------------------------------------------------------------------------ encode, pass 2 ---
__device__ __forceinline__ void ansa_put(u32 &st, u32 rec, StreamOut<true> &so)
{
    const u32 f = rec & 0x7fffu, c0 = rec >> 15;
    const bool emit = st >= (f << 16);
    so.put16_if(emit, st);
    st = emit ? st >> 16 : st;
    u32 q = (u32)((float)st * __builtin_amdgcn_rcpf((float)f));          // st/f within +-1
    u32 r = st - __umul24(q, f);                                         // q < 2^16+1, f < 2^15
    if ((int)r < 0) { q--; r += f; }
    if (r >= f) { q++; r -= f; }
    st = (q << TRC_PROB_BITS) + r + c0;
}

__global__ __launch_bounds__(64) void trc_ansa_code_kernel(
    const u8 *__restrict__ recs, u64 n, u32 chunk, u32 nchunks,
    u8 *__restrict__ scratch, u32 stride, u32 *__restrict__ clen, u32 *__restrict__ gsum)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    const u32 lane = threadIdx.x;
    WaveChunks wc;
    wc.c0 = blockIdx.x * 64u; wc.chunk = chunk; wc.nchunks = nchunks;
    wc.lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    wc.rows = nchunks - wc.c0 < 64u ? nchunks - wc.c0 : 64u;
    WaveChunks wr = wc; wr.chunk = 8u * chunk; wr.lastlen = 8u * (wc.lastlen + (wc.lastlen & 1u));
    const bool alive = lane < wc.rows;
    const u32 c = wc.c0 + lane;
    const u32 len = alive ? wc.len_of(lane) : 0u;
    const u32 nrec = 2u * (len + (len & 1u));                  // 4 per byte pair

    TileIn tin; tin.tile = smem; tin.base = recs + (u64)wc.c0 * wr.chunk;
    StreamOut<true> so;
    so.rings = smem + TRC_TILE_BYTES; so.sel = smem + TRC_TILE_BYTES + TRC_SRING_BYTES;
    so.scratch = scratch; so.stride = stride; so.c0 = wc.c0; so.wpos = 0; so.nfl = 0;
    u32 st[4] = { TRC_ANS_LOW, TRC_ANS_LOW, TRC_ANS_LOW, TRC_ANS_LOW };
    bool ovf = false;
    u32 wpos_last = 0;                                         // words position just before the LAST record (record 0)

    const u32 S = (8u * chunk) / TRC_SEG;                      // record segments (16 records each) in a full chunk
    const u32 top = alive ? (nrec - 1u) / 16u : 0u;
    tin.issue(wr, (S - 1u) * TRC_SEG);
    for (u32 s = S - 1u;; s--) {
        tin.commit();
        if (s) tin.issue(wr, (s - 1u) * TRC_SEG);
        const bool act = alive && s <= top && !ovf;
        // the top segment of the last chunk may hold fewer than 16 records
        const u32 hi = (act && s == top) ? nrec - 16u * top : 16u;
        if (act) {
            uint4 q[4] = { tin.read(0), tin.read(1), tin.read(2), tin.read(3) };
            const u32 *rr = (const u32 *)q;
#pragma unroll
            for (int i = 15; i >= 0; i--) {
                if ((u32)i < hi && !ovf) {
                    if (so.wpos + 18u >= len) ovf = true;      // mnflush: ep <= op + 2 + 16  ->  goto ovr (raw)
                    else {
                        if (s == 0 && i == 0) wpos_last = so.wpos;
                        ansa_put(st[3 - (i & 3)], rr[i], so);  // record index 16*s + i, 16*s is a multiple of 4
                    }
                }
            }
        }
        so.drain(false, alive);                                // <= 32 new bytes (16 records) per lane
        if (s == 0) break;
    }
    (void)wpos_last;
    u32 out_len = 0;
    if (alive) {
        if (!ovf) {
            for (int k = 0; k < 4; k++) { so.put16(st[k] >> 16); so.put16(st[k]); }
            if (so.wpos >= len) ovf = true;
        }
        out_len = ovf ? len : so.wpos;
    }
    so.drain(true, alive && !ovf);
    if (alive) clen[c] = out_len;
    const u32 gs = trc_wave_sum(out_len);
    if (lane == 0) gsum[wc.c0 >> 6] = gs;
}

// ------------------------------------------------------------------------------------- decode ---
__global__ __launch_bounds__(64) void trc_ansa_dec_kernel(
    const u8 *__restrict__ payload, const u32 *__restrict__ clen, const u64 *__restrict__ goff, const u32 *__restrict__ gsum,
    u64 n, u32 chunk, u32 nchunks, u8 *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    const u32 lane = threadIdx.x;
    NibModel m; m.row = smem + lane * TRC_NIB_ROW; m.reset();
    u8 *wbase = smem + TRC_NIB_BYTES;

    WaveChunks wc;
    wc.c0 = blockIdx.x * 64u; wc.chunk = chunk; wc.nchunks = nchunks;
    wc.lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    wc.rows = nchunks - wc.c0 < 64u ? nchunks - wc.c0 : 64u;
    const bool alive = lane < wc.rows;
    const u32 c = wc.c0 + lane;
    const u32 len = alive ? wc.len_of(lane) : 0u;
    const u32 cl = alive ? clen[c] : 0u;
    const u32 ex = trc_wave_incl_scan(cl) - cl;
    const u64 off = trc_group_base(goff, gsum, wc.c0 >> 6) + ex;
    const bool coded = alive && cl != len;

    TileOut tout; tout.tile = wbase; tout.base = out + (u64)wc.c0 * chunk;
    StreamIn si;
    si.rings = wbase + TRC_TILE_BYTES; si.sel = wbase + TRC_TILE_BYTES + TRC_SRING_BYTES;
    si.gbase = payload; si.soff = off + 16;                    // words follow the four states
    u32 st[4] = { 0, 0, 0, 0 };
    if (coded) for (int k = 0; k < 4; k++) st[k] = trc_ld32_a2(payload + off + 4 * k);   // decoder st[i] = encoder st[3-i] (mnfill)
    si.prime(coded);

    // cdf16ansdec: search + state update + model update; the renorm comes later (order matters)
    auto get_nibble = [&](u32 &s, u8 *tb) -> u32 {
        const u32 slot = s & (TRC_PROB_ONE - 1);
        NibTable T = m.load(tb);
        const u32 x = 15u - trc_nib_count_gt(T, slot);
        u32 c0, c1; m.bounds(tb, x, c0, c1);
        s = __umul24(c1 - c0, s >> TRC_PROB_BITS) + slot - c0;
        trc_nib_adapt(T, slot); m.store(tb, T);
        return x;
    };
    auto renorm = [&](u32 &s) {
        const u32 w = si.peek16();
        const bool rn = s < TRC_ANS_LOW;
        s = rn ? (s << 16) | w : s;
        si.rpos += rn ? 2u : 0u;
    };
    auto get_pair = [&]() -> u32 {                             // mndec8x2: two bytes, then four renorms in order st0..st3
        const u32 h0 = get_nibble(st[0], m.table(0)), l0 = get_nibble(st[1], m.table(1u + h0));
        const u32 h1 = get_nibble(st[2], m.table(0)), l1 = get_nibble(st[3], m.table(1u + h1));
        renorm(st[0]); renorm(st[1]); renorm(st[2]); renorm(st[3]);
        return (h0 << 4 | l0) | (h1 << 4 | l1) << 8;
    };

    const u32 S = chunk / TRC_SEG;
    u8 *dst = out + (u64)c * chunk;
    for (u32 s = 0; s < S; s++) {
        for (u32 k = 0; k < 4; k++) {
            const u32 p0 = s * TRC_SEG + k * 16u;
            u32 w[4] = { 0, 0, 0, 0 };
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {                   // period = 8 bytes = 4 pairs: <= 16 renorm words = 32 B
                const u32 q0 = p0 + (u32)hh * 8u;
                si.period(coded && q0 < len, hh);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const u32 pos = q0 + 2u * (u32)j;
                    if (coded && pos < len) {
                        const u32 two = get_pair();
                        w[2 * hh + (j >> 1)] |= two << (16 * (j & 1));
                    }
                }
            }
            if (coded && p0 + 16u <= len) tout.put(k, make_uint4(w[0], w[1], w[2], w[3]));
            else if (coded && p0 < len)
                for (u32 pos = p0; pos < len; pos++) dst[pos] = (u8)(w[(pos - p0) >> 2] >> (8 * ((pos - p0) & 3u)));
        }
        tout.flush(wc, s * TRC_SEG);
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

// ------------------------------------------------------------------------------------- launch ---
void trc_launch_ansa_enc(const uint8_t *d_in, size_t n, uint32_t chunk, const TrcWork &w, uint32_t *d_clen, hipStream_t s)
{
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void *)trc_ansa_model_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ANSA_MODEL_LDS);
        attr = true;
    }
    hipLaunchKernelGGL(trc_ansa_model_kernel, dim3(w.ngroups), dim3(64), ANSA_MODEL_LDS, s, d_in, (u64)n, chunk, w.nchunks, w.scratch2);
    hipLaunchKernelGGL(trc_ansa_code_kernel, dim3(w.ngroups), dim3(64), ANSA_CODE_LDS, s,
                       (const u8 *)w.scratch2, (u64)n, chunk, w.nchunks, w.scratch, w.stride, d_clen, w.gsum);
}
void trc_launch_ansa_dec(const uint8_t *d_payload, const uint32_t *d_clen, size_t n, uint32_t chunk,
                         const TrcWork &w, uint8_t *d_out, hipStream_t s)
{
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void *)trc_ansa_dec_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ANSA_DEC_LDS); attr = true; }
    hipLaunchKernelGGL(trc_ansa_dec_kernel, dim3(w.ngroups), dim3(64), ANSA_DEC_LDS, s,
                       d_payload, d_clen, w.goff, w.gsum, (u64)n, chunk, w.nchunks, d_out);
}
