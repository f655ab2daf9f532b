// trc_rc_adaptive.hip -- adaptive-CDF byte range coder: one stream (TRC_RCA, `turborc -e46`) and the
// interleaved variant (TRC_RCAI, `turborc -e47`: hi nibbles on stream 0, lo nibbles on stream 1, one model;
// rccdfienc/rccdfidec rccdf.c:213-249, payload [u32 len0][stream 0][stream 1], OVERFLOWI after each full
// group of 4 bytes -- SURVEY 8f rank 1).
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

#define RCA_WAVE_LDS(NS) (TRC_NIB_BYTES + TRC_TILE_BYTES + (NS) * TRC_SRING_BYTES + TRC_SEL_BYTES)

template <int NS>
__global__ __launch_bounds__(64) void trc_rca_enc_kernel(
    const u8 *__restrict__ in, u64 n, u32 chunk, u32 nchunks,
    u8 *__restrict__ scratch, u32 stride, u8 *__restrict__ scratch2, u32 stride2,
    u32 *__restrict__ clen, u32 *__restrict__ gsum)
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
    StreamOut<false> so, so1;
    so.rings = wbase + TRC_TILE_BYTES; so.sel = wbase + TRC_TILE_BYTES + NS * TRC_SRING_BYTES;
    so.scratch = scratch; so.stride = stride; so.c0 = wc.c0; so.wpos = (NS == 2) ? 4u : 0u; so.nfl = 0;
    so1 = so;
    if (NS == 2) { so1.rings = so.rings + TRC_SRING_BYTES; so1.scratch = scratch2; so1.stride = stride2; so1.wpos = 0; }
    RcEnc e, e1; e.start(); e1.start();
    const u32 off1 = 4u + len / 2u;                            // stream-1 base inside `out` (rccdf.c:215)
    bool ovf = alive && NS == 1 && lim <= 0;

    auto put_nibble = [&](RcEnc &en, StreamOut<false> &sq, u8 *tb, u32 x) {
        u32 c0, c1; m.bounds(tb, x, c0, c1);
        en.sym(sq, c0, c1 - c0);
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
                        put_nibble(e, so, m.table(0), x >> 4);
                        if (NS == 1) put_nibble(e, so, m.table(1u + (x >> 4)), x & 15u);
                        else         put_nibble(e1, so1, m.table(1u + (x >> 4)), x & 15u);
                    }
                    // OVERFLOW after every byte (NS=1) / OVERFLOWI after every FULL group of 4 (NS=2): both monotone
                    if (NS == 1) ovf = (int)(4u * e.cw.nwords) >= lim;
                    else if (nb == 4u) ovf = ((int)(off1 + 4u * e1.cw.nwords) >= lim) || (4u + 4u * e.cw.nwords >= off1);
                }
                so.drain(false, alive);
                if (NS == 2) so1.drain(false, alive);
            }
        }
    }
    u32 out_len = 0;
    if (alive) {
        if (!ovf) {
            e.finish(so);
            if (NS == 2) { e1.finish(so1); out_len = so.wpos + so1.wpos; if ((int)out_len >= lim) ovf = true; }
            else out_len = so.wpos;
        }
        if (ovf) out_len = len;
    }
    so.drain(true, alive && !ovf);
    if (NS == 2) {
        so1.drain(true, alive && !ovf);
        if (alive && !ovf) *(u32 *)(scratch + (u64)c * stride) = so.wpos - 4u;       // header: len0
    }
    if (alive) clen[c] = out_len;
    const u32 gs = trc_wave_sum(out_len);
    if (lane == 0) gsum[wc.c0 >> 6] = gs;
}

template <int NS>
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
    StreamIn si, si1;
    si.rings = wbase + TRC_TILE_BYTES; si.sel = wbase + TRC_TILE_BYTES + NS * TRC_SRING_BYTES;
    si.gbase = payload; si.soff = off + (NS == 2 ? 4u : 0u);
    si1 = si;
    if (NS == 2) { si1.rings = si.rings + TRC_SRING_BYTES; si1.soff = off + 4u + (coded ? trc_ld32_a2(payload + off) : 0u); }
    si.prime(coded);
    if (NS == 2) si1.prime(coded);
    RcDec dc, dc1;
    { const u32 a = si.peek32(); si.rpos += 4; const u32 b = si.peek32(); si.rpos += 4; dc.start(a, b); }
    if (NS == 2) { const u32 a = si1.peek32(); si1.rpos += 4; const u32 b = si1.peek32(); si1.rpos += 4; dc1.start(a, b); }
    else dc1 = dc;

    auto get_nibble = [&](RcDec &dq, StreamIn &sq, u8 *tb) -> u32 {
        dq.range >>= TRC_PROB_BITS;
        const u32 q = dq.quotient();
        NibTable T = m.load(tb);
        const u32 x = 15u - trc_nib_count_gt(T, q);            // first i with t[i+1] > q, else 15
        u32 c0, c1; m.bounds(tb, x, c0, c1);
        dq.consume(sq, c0, c1);
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
                if (NS == 2) si1.period(coded && q0 < len, d & 1);
                if (coded && q0 < len) {
                    const u32 nb = len - q0 < 4u ? len - q0 : 4u;
                    for (u32 i = 0; i < nb; i++) {
                        const u32 h = get_nibble(dc, si, m.table(0));
                        const u32 l = NS == 1 ? get_nibble(dc, si, m.table(1u + h)) : get_nibble(dc1, si1, m.table(1u + h));
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

template <int NS>
static void launch_rca_enc(const uint8_t *d_in, size_t n, uint32_t chunk, const TrcWork &w, uint32_t *d_clen, hipStream_t s)
{
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void *)trc_rca_enc_kernel<NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RCA_WAVE_LDS(NS)); attr = true; }
    hipLaunchKernelGGL(trc_rca_enc_kernel<NS>, dim3(w.ngroups), dim3(64), RCA_WAVE_LDS(NS), s,
                       d_in, (u64)n, chunk, w.nchunks, w.scratch, w.stride, w.scratch2, w.stride2, d_clen, w.gsum);
}
template <int NS>
static void launch_rca_dec(const uint8_t *d_payload, const uint32_t *d_clen, size_t n, uint32_t chunk,
                           const TrcWork &w, uint8_t *d_out, hipStream_t s)
{
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void *)trc_rca_dec_kernel<NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RCA_WAVE_LDS(NS)); attr = true; }
    hipLaunchKernelGGL(trc_rca_dec_kernel<NS>, dim3(w.ngroups), dim3(64), RCA_WAVE_LDS(NS), s,
                       d_payload, d_clen, w.goff, w.gsum, (u64)n, chunk, w.nchunks, d_out);
}
void trc_launch_rca_enc(int nstreams, const uint8_t *d_in, size_t n, uint32_t chunk, const TrcWork &w, uint32_t *d_clen, hipStream_t s)
{
    if (nstreams == 2) launch_rca_enc<2>(d_in, n, chunk, w, d_clen, s); else launch_rca_enc<1>(d_in, n, chunk, w, d_clen, s);
}
void trc_launch_rca_dec(int nstreams, const uint8_t *d_payload, const uint32_t *d_clen, size_t n, uint32_t chunk,
                        const TrcWork &w, uint8_t *d_out, hipStream_t s)
{
    if (nstreams == 2) launch_rca_dec<2>(d_payload, d_clen, n, chunk, w, d_out, s); else launch_rca_dec<1>(d_payload, d_clen, n, chunk, w, d_out, s);
}
