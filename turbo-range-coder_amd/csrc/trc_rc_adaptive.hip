// trc_rc_adaptive.hip -- adaptive-CDF byte range coder (codec TRC_RCA; `turborc -e46`).
//
// Per chunk the payload is exactly what rccdfenc returns for that slice (reference rccdf.c:201-211,
// cdf8e/cdf4e rccdf_.h:28-34; decoder rccdf.c:187-200, cdf8d/cdf4d rccdf_.h:48-54 with the 16-way
// search cdflget16 turborc_.h:259-304): per byte the hi nibble is coded with the "hi" CDF16 table,
// then that table adapts; the lo nibble is coded with the "lo" table selected by the hi nibble,
// then that table adapts (model: trc_nibmodel.h); OVERFLOW (rcutil_.h:130) after every byte.
// Range coder core, carry scheme and the exact code/range quotient: trc_rc.h.
//
// One lane = one chunk; the lane's 544-byte model sits in LDS (35 KiB per wave), which caps
// occupancy at 3 waves per CU -- the coder is VALU-bound on the 56-op packed table update anyway.
#include "trc_rc.h"
#include "trc_nibmodel.h"
#include "trc_launch.h"

#define RCA_WAVE_LDS (TRC_NIB_BYTES + TRC_TILE_BYTES + TRC_SRING_BYTES + TRC_SEL_BYTES)

__global__ __launch_bounds__(64) void trc_rca_enc_kernel(
    const u8 *__restrict__ in, u64 n, u32 chunk, u32 nchunks,
    u8 *__restrict__ scratch, u32 stride, u32 *__restrict__ clen, u32 *__restrict__ gsum)
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
    const int lim = trc_rc_limit(len);

    TileIn tin; tin.tile = wbase; tin.base = in + (u64)wc.c0 * chunk;
    StreamOut<false> so;
    so.rings = wbase + TRC_TILE_BYTES; so.sel = wbase + TRC_TILE_BYTES + TRC_SRING_BYTES;
    so.scratch = scratch; so.stride = stride; so.c0 = wc.c0; so.wpos = 0; so.nfl = 0;
    RcEnc e; e.start();
    bool ovf = alive && lim <= 0;

    auto put_nibble = [&](u8 *tb, u32 x) {
        u32 c0, c1; m.bounds(tb, x, c0, c1);
        e.sym(so, c0, c1 - c0);
        NibTable T = m.load(tb); trc_nib_adapt(T, c0); m.store(tb, T);
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
            for (int d = 0; d < 4; d++) {                      // period = 4 bytes = 8 nibbles: <= 8 words
                const u32 q0 = p0 + (u32)d * 4u;
                if (alive && !ovf && q0 < len) {
                    const u32 nb = len - q0 < 4u ? len - q0 : 4u;
                    for (u32 i = 0; i < nb; i++) {
                        const u32 x = (w[d] >> (8 * i)) & 255u;
                        put_nibble(m.table(0), x >> 4);
                        put_nibble(m.table(1u + (x >> 4)), x & 15u);
                    }
                }
                so.drain(false, alive);
                ovf = ovf || (alive && (int)(4u * e.nwords) >= lim);
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

__global__ __launch_bounds__(64) void trc_rca_dec_kernel(
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
    si.gbase = payload; si.soff = off;
    si.prime(coded);
    RcDec dc;
    { const u32 a = si.peek32(); si.rpos += 4; const u32 b = si.peek32(); si.rpos += 4; dc.start(a, b); }

    auto get_nibble = [&](u8 *tb) -> u32 {
        dc.range >>= TRC_PROB_BITS;
        const u32 q = dc.quotient();
        NibTable T = m.load(tb);
        const u32 x = 15u - trc_nib_count_gt(T, q);            // first i with t[i+1] > q, else 15
        u32 c0, c1; m.bounds(tb, x, c0, c1);
        dc.consume(si, c0, c1);
        trc_nib_adapt(T, q); m.store(tb, T);
        return x;
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
                    for (u32 i = 0; i < nb; i++) {
                        const u32 h = get_nibble(m.table(0));
                        const u32 l = get_nibble(m.table(1u + h));
                        w[d] |= (h << 4 | l) << (8 * i);
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

void trc_launch_rca_enc(const uint8_t *d_in, size_t n, uint32_t chunk, const TrcWork &w, uint32_t *d_clen, hipStream_t s)
{
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void *)trc_rca_enc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RCA_WAVE_LDS); attr = true; }
    hipLaunchKernelGGL(trc_rca_enc_kernel, dim3(w.ngroups), dim3(64), RCA_WAVE_LDS, s,
                       d_in, (u64)n, chunk, w.nchunks, w.scratch, w.stride, d_clen, w.gsum);
}
void trc_launch_rca_dec(const uint8_t *d_payload, const uint32_t *d_clen, size_t n, uint32_t chunk,
                        const TrcWork &w, uint8_t *d_out, hipStream_t s)
{
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void *)trc_rca_dec_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RCA_WAVE_LDS); attr = true; }
    hipLaunchKernelGGL(trc_rca_dec_kernel, dim3(w.ngroups), dim3(64), RCA_WAVE_LDS, s,
                       d_payload, d_clen, w.goff, w.gsum, (u64)n, chunk, w.nchunks, d_out);
}
