// trc_ans_vlc.hip -- Turbo-VLC integer coders over the adaptive CDF rANS (second half):
//   anscdfuenc16 / anscdfudec16, anscdfuzenc16 / anscdfuzdec16      VN = 1 ("vlc6"), 16-bit elements     anscdf.c:139-253   `turborc -e60/61`
//   anscdfvenc16/32 / anscdfvdec16/32, anscdfvzenc16/32 / anscdfvzdec16/32   VN = 2 ("vlc7")            anscdf.c:255-483   `-e62/63`
// ("z" = zigzag of the delta to the previous element; cdfenc6/7, cdfdec6/7 anscdf_.h:205-230; mnflush :128-138; the
// element split and the mantissa bit string: trc_vlc.h.)  A chunk is far below the reference's block of 4 Mi elements,
// so per chunk the payload is exactly what the reference returns for that slice:
//     [u32 total][st1][st0][u16 rANS words in decode order][mantissa bytes]
// The first symbol of an element is coded on rANS state 1, the second (if any) on state 0; the decoder renormalises
// right after each symbol.  Raw rules (the reference's pointer tests as offsets, B = floor(mantissa bits / 8)): before
// every record words + 30 + B >= len; after the states 4 + words + 8 + 16 + B >= len; at the end total >= len.
//
// Two kernels like the other adaptive rANS coders.  Pass 1 walks the chunk forward: two CDF16 tables per lane in LDS
// (the only LDS use), mantissas to the END of the chunk's slot in scratch2, and for every element two 32-bit record
// slots {cdf_lo << 15 | freq} (second slot 0 when the element has one symbol) -- a uniform 8 B per element, so the
// stack moves in whole 64-byte segments.  Pass 2 pops the slots backwards (second symbol first), words growing down
// from the end of the chunk's region in scratch, and puts the 4-byte header in front of the finished payload.
#include "trc_io.h"
#include "trc_lane_io.h"
#include "trc_nibmodel.h"
#include "trc_vlc.h"
#include "trc_launch.h"

#define VLA_CODE_LDS (TRC_TILE_BYTES + TRC_SRING_BYTES)

// record space of one wave's chunks: 8 bytes per element
template <int ES>
__device__ __forceinline__ WaveChunks vla_record_space(const WaveChunks &wc, u32 stride2)
{
    WaveChunks wr = wc;
    wr.chunk = stride2;                                        // slot of a chunk in scratch2 (records, then room for the mantissa bytes)
    wr.lastlen = (8u * ((wc.lastlen + ES - 1u) / ES) + 15u) & ~15u;
    return wr;
}

// ------------------------------------------------------------------------------ encode, pass 1 ---
template <int ES, int VN, bool ZZ>
__global__ __launch_bounds__(64) void trc_vla_model_kernel(
    const u8 *__restrict__ in, u64 n, u32 chunk, u32 nchunks, u8 *__restrict__ recs, u32 stride2, u32 *__restrict__ aux)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    const u32 lane = threadIdx.x;
    NibModel<2> m; m.init(smem);
    constexpr u32 T = VN == 2 ? 8u : 12u, FIRST = 1u << (VN + 1), VM = (1u << VN) - 1u;

    WaveChunks wc;
    wc.c0 = blockIdx.x * 64u; wc.chunk = chunk; wc.nchunks = nchunks;
    wc.lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    wc.rows = nchunks - wc.c0 < 64u ? nchunks - wc.c0 : 64u;
    const WaveChunks wr = vla_record_space<ES>(wc, stride2);
    const bool alive = lane < wc.rows;
    const u32 c = wc.c0 + lane;
    const u32 len = alive ? wc.len_of(lane) : 0u;

    QuadIn qin; qin.base = in + (u64)wc.c0 * chunk;
    QuadOut qout; qout.base = recs + (u64)wc.c0 * stride2;
    LaneBitsDown bo; bo.start(recs + (u64)(c + 1u) * stride2);
    u32 prev = 0;

    NibTable T0 = m.load(m.table(0)), T1 = T0;                 // both tables in registers (record_r, trc_nibmodel.h); all tables start alike
    auto elem_records = [&](u32 v, bool act, u32 &r0, u32 &r1) {
        u32 x = v;
        if (ZZ) { x = vlc_zigzag_enc(v - prev, ES == 4); prev = act ? v : prev; }
        const bool big = x >= FIRST;
        const u32 f = (31u - (u32)__clz((int)(x | 1u))) - (u32)VN;
        const u32 expo = ((f + 1u) << VN) + ((x >> (f & 31u)) & VM);
        bo.put_if(act && big, f & 31u, x & ((1u << (f & 31u)) - 1u));
        const u32 xs = big ? expo : x;
        const bool two = xs >= T;
        const u32 y0 = two ? ((xs - T) >> 4) + T : xs, y1 = (xs - T) & 15u;
        r0 = m.record_r(T0, m.table(0), y0 & 15u);
        r1 = 0;
        if (act && two) r1 = m.record_r(T1, m.table(1), y1);    // table 1 adapts only where its symbol is coded
    };

    const u32 S = chunk / TRC_SEG;
    qin.issue(wc, 0);
    for (u32 s = 0; s < S; s++) {
        qin.commit();
        if (s + 1 < S) qin.issue(wc, (s + 1) * TRC_SEG);
        uint4 pc0 = qin.read(0), pc1 = qin.read(1), pc2 = qin.read(2), pc3 = qin.read(3);
#pragma nounroll
        for (u32 k = 0; k < 4; k++) {
            const uint4 v = pc0; pc0 = pc1; pc1 = pc2; pc2 = pc3;
            const u32 p0 = s * TRC_SEG + k * 16u;
            if (!__ballot(alive && p0 < len)) continue;         // (uniform: a skipped piece has no live element in any lane)
            const u32 w[4] = { v.x, v.y, v.z, v.w };
            u32 r[16];
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const u32 q0 = p0 + 4u * (u32)d;
                const u32 nb = len > q0 ? (len - q0 < 4u ? len - q0 : 4u) : 0u;   // a partial last element is zero-extended
                const u32 ww = nb >= 4u ? w[d] : (w[d] & ((1u << (8u * nb)) - 1u));
                if (ES == 2) {
                    elem_records(ww & 0xffffu, alive && q0 < len, r[4 * d], r[4 * d + 1]);
                    elem_records(ww >> 16, alive && q0 + 2u < len, r[4 * d + 2], r[4 * d + 3]);
                } else elem_records(ww, alive && q0 < len, r[2 * d], r[2 * d + 1]);
            }
            if (ES == 2) {                                      // 8 elements -> one 64-byte record segment
#pragma unroll
                for (int j = 0; j < 4; j++) qout.put((u32)j, make_uint4(r[4 * j], r[4 * j + 1], r[4 * j + 2], r[4 * j + 3]));
                qout.flush(wr, p0 * 4u);
            } else {                                           // 4 elements -> half a segment; two pieces make one.  Flushed after either
                // half (the odd piece may be skipped as a whole): slots beyond the chunk's last element are never read.
                if (!(k & 1u)) { qout.put(0, make_uint4(r[0], r[1], r[2], r[3])); qout.put(1, make_uint4(r[4], r[5], r[6], r[7])); }
                else           { qout.put(2, make_uint4(r[0], r[1], r[2], r[3])); qout.put(3, make_uint4(r[4], r[5], r[6], r[7])); }
                qout.flush(wr, (p0 & ~31u) * 2u);
            }
        }
    }
    bo.finish(alive);
    if (alive) { aux[2u * c + 1u] = bo.total; }
}

// ------------------------------------------------------------------------------ encode, pass 2 ---
__device__ __forceinline__ void vla_put(u32 &st, u32 rec, StreamOut<true> &so)
{
    const u32 f = rec & 0x7fffu, c0 = rec >> 15;
    const bool emit = st >= (f << 16);
    so.put16_if(emit, st);
    st = emit ? st >> 16 : st;
    st = trc_rans_step(st, f, TRC_PROB_ONE - f, c0);
}

template <int ES>
__global__ __launch_bounds__(64) void trc_vla_code_kernel(
    const u8 *__restrict__ recs, u32 stride2, u64 n, u32 chunk, u32 nchunks,
    u8 *__restrict__ scratch, u32 stride, u32 *__restrict__ aux, u32 *__restrict__ clen, u32 *__restrict__ gsum)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    const u32 lane = threadIdx.x;
    WaveChunks wc;
    wc.c0 = blockIdx.x * 64u; wc.chunk = chunk; wc.nchunks = nchunks;
    wc.lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    wc.rows = nchunks - wc.c0 < 64u ? nchunks - wc.c0 : 64u;
    const WaveChunks wr = vla_record_space<ES>(wc, stride2);
    const bool alive = lane < wc.rows;
    const u32 c = wc.c0 + lane;
    const u32 len = alive ? wc.len_of(lane) : 0u;
    const u32 nel = (len + ES - 1u) / ES;
    const u32 bits = alive ? aux[2u * c + 1u] : 0u, B = bits >> 3;

    TileIn tin; tin.tile = smem; tin.base = recs + (u64)wc.c0 * stride2;
    StreamOut<true> so;
    so.rings = smem + TRC_TILE_BYTES;
    so.scratch = scratch; so.stride = stride; so.c0 = wc.c0; so.wpos = 0; so.nfl = 0;
    u32 st0 = TRC_ANS_LOW, st1 = TRC_ANS_LOW;
    bool ovf = false;

    const u32 S = (8u * (chunk / ES)) / TRC_SEG;               // record segments (8 elements each) of a full chunk
    const u32 top = alive ? (nel - 1u) / 8u : 0u;
    tin.issue(wr, (S - 1u) * TRC_SEG);
    for (u32 s = S - 1u;; s--) {
        tin.commit();
        if (s) tin.issue(wr, (s - 1u) * TRC_SEG);
        const bool act = alive && s <= top && !ovf;
        const u32 hi = (act && s == top) ? nel - 8u * top : 8u;  // elements of the chunk in this segment
        if (act) {
            const uint4 q[4] = { tin.read(0), tin.read(1), tin.read(2), tin.read(3) };
            const u32 *rr = (const u32 *)q;
#pragma unroll
            for (int i = 15; i >= 0; i--) {                    // slot i: element i/2, symbol i%2 (the second symbol is popped first)
                const u32 rec = rr[i];
                if ((u32)(i >> 1) < hi && rec != 0u && !ovf) {
                    if (so.wpos + 30u + B >= len) ovf = true;  // mnflush: ep <= op + 2 + 8 with ep = bp - 8 - words
                    else if (i & 1) vla_put(st0, rec, so);
                    else            vla_put(st1, rec, so);
                }
            }
        }
        so.drain(false, alive);                                // <= 32 new bytes (16 records) per lane
        if (s == 0) break;
    }
    u32 out_len = 0, la = 0;
    if (alive) {
        if (!ovf) {
            so.put16(st0 >> 16); so.put16(st0); so.put16(st1 >> 16); so.put16(st1);      // eceflush st[0], st[1]: st1 lowest
            if (so.wpos + 20u + B >= len) ovf = true;          // op + l >= bp - 8
            la = 4u + so.wpos;
            out_len = la + ((bits + 7u) >> 3);
            if (out_len >= len) ovf = true;                    // op + l >= out_
        }
        if (ovf) out_len = len;
    }
    so.drain(true, alive && !ovf);
    if (alive && !ovf) {
        *(u32_a2 *)(scratch + (u64)(c + 1u) * stride - la) = out_len;       // header in front of the (end-aligned) payload
        aux[2u * c] = la;
    }
    if (alive) clen[c] = out_len;
    const u32 gs = trc_wave_sum(out_len);
    if (lane == 0) gsum[wc.c0 >> 6] = gs;
}

// ------------------------------------------------------------------------------------- decode ---
template <int ES, int VN, bool ZZ>
__global__ __launch_bounds__(64) void trc_vla_dec_kernel(
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
    const bool coded = alive && cl != len && cl >= 12u;        // header + two states

    u32 sa = TRC_ANS_LOW, sb = TRC_ANS_LOW;
    if (coded) { sa = trc_ld32_a2(payload + off + 4u); sb = trc_ld32_a2(payload + off + 8u); }   // sa = encoder state 1 (first symbols)
    LaneIn<2> si; si.prime(payload + off + 12u, coded, trc_sub_sat(cl, 12u));
    const u8 *bend = payload + off + cl;
    u32 bpos = 0, prev = 0;

    // both tables live in registers for the whole chunk (a decoder's table loads and stores are dependent LDS round trips that
    // one wave per SIMD cannot hide; LDS keeps only the K rows of the update)
    NibTable T0 = m.load(m.table(0)), T1 = T0;                 // all tables start alike
    auto get = [&](u32 &s, NibTable &Tb, bool act) -> u32 {    // mndec4: cdf16ansdec, then ecdnorm
        const u32 slot = s & (TRC_PROB_ONE - 1);
        u32 c0, c1;
        const u32 x = trc_nib_find(Tb, slot, c0, c1);
        u32 ns = __umul24(c1 - c0, s >> TRC_PROB_BITS) + slot - c0;
        const u32 w = si.peek16();
        const bool rn = act && ns < TRC_ANS_LOW;
        ns = rn ? (ns << 16) | w : ns;
        si.skip_if(rn);
        s = act ? ns : s;
        m.adapt(Tb, x);
        return x;
    };
    auto get_elem = [&](bool act) -> u32 {
        u32 x = get(sa, T0, act);
        if (act && x >= T) { const u32 z = get(sb, T1, true); x = ((x - T) << 4 | z) + T; }
        if (act && x >= FIRST) {
            u32 f = (x >> VN) - 1u;
            f = f > 30u ? 30u : f;                             // (corrupt input)
            const u32 byteoff = trc_min(bpos >> 3, cl - 8u);
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
static void launch_vla_enc(const uint8_t *d_in, size_t n, uint32_t chunk, const TrcWork &w, uint32_t *d_clen, hipStream_t s)
{
    TRC_LAUNCH_TIMED((trc_vla_model_kernel<ES, VN, ZZ>), dim3(w.ngroups), dim3(64), TRC_NIB2_BYTES, s,
                     d_in, (u64)n, chunk, w.nchunks, w.scratch2, w.stride2, w.aux);
    TRC_LAUNCH_TIMED((trc_vla_code_kernel<ES>), dim3(w.ngroups), dim3(64), VLA_CODE_LDS, s,
                       (const u8 *)w.scratch2, w.stride2, (u64)n, chunk, w.nchunks, w.scratch, w.stride, w.aux, d_clen, w.gsum);
}
template <int ES, int VN, bool ZZ>
static void launch_vla_dec(const uint8_t *d_payload, const uint32_t *d_clen, size_t n, uint32_t chunk,
                           const TrcWork &w, uint8_t *d_out, hipStream_t s)
{
    TRC_LAUNCH_TIMED((trc_vla_dec_kernel<ES, VN, ZZ>), dim3(w.ngroups), dim3(64), TRC_NIB2_BYTES, s,
                     d_payload, d_clen, w.goff, w.gsum, (u64)n, chunk, w.nchunks, d_out);
}
// variant 0 = u (vlc6, 16-bit only), 1 = v (vlc7); zz = zigzag-delta form; elem = 2 or 4 bytes
void trc_launch_vla_enc(int variant, int zz, int elem, const uint8_t *d_in, size_t n, uint32_t chunk, const TrcWork &w, uint32_t *d_clen, hipStream_t s)
{
    if (variant == 0) { if (zz) launch_vla_enc<2, 1, true>(d_in, n, chunk, w, d_clen, s); else launch_vla_enc<2, 1, false>(d_in, n, chunk, w, d_clen, s); }
    else if (elem == 2) { if (zz) launch_vla_enc<2, 2, true>(d_in, n, chunk, w, d_clen, s); else launch_vla_enc<2, 2, false>(d_in, n, chunk, w, d_clen, s); }
    else { if (zz) launch_vla_enc<4, 2, true>(d_in, n, chunk, w, d_clen, s); else launch_vla_enc<4, 2, false>(d_in, n, chunk, w, d_clen, s); }
}
void trc_launch_vla_dec(int variant, int zz, int elem, const uint8_t *d_payload, const uint32_t *d_clen, size_t n, uint32_t chunk,
                        const TrcWork &w, uint8_t *d_out, hipStream_t s)
{
    if (variant == 0) { if (zz) launch_vla_dec<2, 1, true>(d_payload, d_clen, n, chunk, w, d_out, s); else launch_vla_dec<2, 1, false>(d_payload, d_clen, n, chunk, w, d_out, s); }
    else if (elem == 2) { if (zz) launch_vla_dec<2, 2, true>(d_payload, d_clen, n, chunk, w, d_out, s); else launch_vla_dec<2, 2, false>(d_payload, d_clen, n, chunk, w, d_out, s); }
    else { if (zz) launch_vla_dec<4, 2, true>(d_payload, d_clen, n, chunk, w, d_out, s); else launch_vla_dec<4, 2, false>(d_payload, d_clen, n, chunk, w, d_out, s); }
}
