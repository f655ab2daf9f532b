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
// One lane = one chunk = one range-coder state.  The lane's 256 x u16 model lives in LDS in a
// [context][lane] layout (2 lanes per bank, independent of the context each lane is at), 32 KiB per
// wave: that, not registers, bounds occupancy (3 waves per CU).  Eight dependent LDS
// read-modify-writes per byte make this the slowest coder of the family; HBM traffic is coalesced
// through the same tiles/rings as everywhere else (trc_io.h).
#include "trc_rc.h"
#include "trc_launch.h"

#define RCB_MODEL_BYTES (256u * 64u * 2u)                  // [ctx][lane] u16
#define RCB_WAVE_LDS    (RCB_MODEL_BYTES + TRC_TILE_BYTES + TRC_SRING_BYTES + TRC_SEL_BYTES)

__device__ __forceinline__ u32 rcb_adapt(u32 p, u32 bit) { return (p - (((p - (bit << TRC_PROB_BITS)) >> 5) + bit)) & 0xffffu; }

__global__ __launch_bounds__(64) void trc_rcb_enc_kernel(
    const u8 *__restrict__ in, u64 n, u32 chunk, u32 nchunks,
    u8 *__restrict__ scratch, u32 stride, u32 *__restrict__ clen, u32 *__restrict__ gsum)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    const u32 lane = threadIdx.x;
    u16 *mb = (u16 *)smem + lane;                              // mb[ctx * 64]
    u8 *wbase = smem + RCB_MODEL_BYTES;
    for (u32 i = 0; i < 256; i++) mb[i * 64] = (u16)(TRC_PROB_ONE >> 1);

    WaveChunks wc;
    wc.c0 = blockIdx.x * 64u; wc.chunk = chunk; wc.nchunks = nchunks;
    wc.lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    wc.rows = nchunks - wc.c0 < 64u ? nchunks - wc.c0 : 64u;
    const bool alive = lane < wc.rows;
    const u32 c = wc.c0 + lane;
    const u32 len = alive ? wc.len_of(lane) : 0u;
    const int lim = trc_rc_limit(len);

    TileIn tin; tin.tile = wbase; tin.base = in + (u64)wc.c0 * chunk;
    StreamOut<false> so;
    so.rings = wbase + TRC_TILE_BYTES; so.sel = wbase + TRC_TILE_BYTES + TRC_SRING_BYTES;
    so.scratch = scratch; so.stride = stride; so.c0 = wc.c0; so.wpos = 0; so.nfl = 0;
    RcEnc e; e.start();
    bool ovf = alive && lim <= 0;

    auto put_byte = [&](u32 x) {
        u32 ctx = 1;
#pragma unroll
        for (int b = 7; b >= 0; b--) {
            if (b & 1) e.renorm(so);                           // before bits 7,5,3,1 only
            const u32 p = mb[ctx * 64], bit = (x >> b) & 1u;
            e.bit(so, p, bit);
            mb[ctx * 64] = (u16)rcb_adapt(p, bit);
            ctx = ctx * 2 + bit;
        }
    };

    const u32 S = chunk / TRC_SEG;
    tin.issue(wc, 0);
    for (u32 s = 0; s < S; s++) {
        tin.commit();
        if (s + 1 < S) tin.issue(wc, (s + 1) * TRC_SEG);
        for (u32 k = 0; k < 4; k++) {
            const u32 p0 = s * TRC_SEG + k * 16u;
            const uint4 v = tin.read(k);
            const u32 w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
            for (int d = 0; d < 4; d++) {                      // period = 4 bytes: <= 8 words (<= 2 per byte)
                const u32 q0 = p0 + (u32)d * 4u;
                if (alive && !ovf && q0 < len) {
                    const u32 nb = len - q0 < 4u ? len - q0 : 4u;
                    for (u32 i = 0; i < nb; i++) put_byte((w[d] >> (8 * i)) & 255u);
                }
                so.drain(false, alive);
                ovf = ovf || (alive && (int)(4u * e.cw.nwords) >= lim);
            }
        }
    }
    u32 out_len = 0;
    if (alive) {
        if (!ovf) { e.finish(so); out_len = so.wpos; }
        else out_len = len;
    }
    so.drain(true, alive && !ovf);
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
    u8 *wbase = smem + RCB_MODEL_BYTES;
    for (u32 i = 0; i < 256; i++) mb[i * 64] = (u16)(TRC_PROB_ONE >> 1);

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
    si.gbase = payload; si.soff = off;
    si.prime(coded);
    RcDec dc;
    { const u32 a = si.peek32(); si.rpos += 4; const u32 b = si.peek32(); si.rpos += 4; dc.start(a, b); }

    auto get_byte = [&]() -> u32 {
        u32 ctx = 1;
#pragma unroll
        for (int b = 7; b >= 0; b--) {
            if (b & 1) dc.renorm(si);
            const u32 p = mb[ctx * 64];
            const u64 cut = (dc.range >> TRC_PROB_BITS) * p;
            const bool one = dc.code < cut;                    // rcbd_
            dc.range = one ? cut : dc.range - cut;
            dc.code -= one ? 0 : cut;
            mb[ctx * 64] = (u16)rcb_adapt(p, one ? 1u : 0u);
            ctx = ctx * 2 + (one ? 1u : 0u);
        }
        return ctx & 255u;
    };

    const u32 S = chunk / TRC_SEG;
    u8 *dst = out + (u64)c * chunk;
    for (u32 s = 0; s < S; s++) {
        for (u32 k = 0; k < 4; k++) {
            const u32 p0 = s * TRC_SEG + k * 16u;
            u32 w[4] = { 0, 0, 0, 0 };
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const u32 q0 = p0 + (u32)d * 4u;
                si.period(coded && q0 < len, d & 1);
                if (coded && q0 < len) {
                    const u32 nb = len - q0 < 4u ? len - q0 : 4u;
                    for (u32 i = 0; i < nb; i++) w[d] |= get_byte() << (8 * i);
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

void trc_launch_rcb_enc(const uint8_t *d_in, size_t n, uint32_t chunk, const TrcWork &w, uint32_t *d_clen, hipStream_t s)
{
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void *)trc_rcb_enc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RCB_WAVE_LDS); attr = true; }
    hipLaunchKernelGGL(trc_rcb_enc_kernel, dim3(w.ngroups), dim3(64), RCB_WAVE_LDS, s,
                       d_in, (u64)n, chunk, w.nchunks, w.scratch, w.stride, d_clen, w.gsum);
}
void trc_launch_rcb_dec(const uint8_t *d_payload, const uint32_t *d_clen, size_t n, uint32_t chunk,
                        const TrcWork &w, uint8_t *d_out, hipStream_t s)
{
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void *)trc_rcb_dec_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RCB_WAVE_LDS); attr = true; }
    hipLaunchKernelGGL(trc_rcb_dec_kernel, dim3(w.ngroups), dim3(64), RCB_WAVE_LDS, s,
                       d_payload, d_clen, w.goff, w.gsum, (u64)n, chunk, w.nchunks, d_out);
}
