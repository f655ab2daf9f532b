// trc_ans_bit.hip -- bitwise order-0 rANS (codec TRC_ANSB; ansbc / ansbd, anscdf.c:672-731, ecbe/ecbd :659-668;
// `turborc -e66`) -- SURVEY 8f rank 2.
//
// 255-node bit tree of 15-bit probabilities P(bit = 1), all 2^14 at the start (bit 1: p += (2^15-p)>>5, bit 0:
// p -= p>>5).  The reference cuts its input into blocks of 8192 bytes: a forward pass records {bit<<15 | p} per bit,
// a backward pass codes the records on 4 rANS states (last bit first; a byte's bits b0..b7 go to states
// 0,1,2,3,0,1,2,3), block payload [st3][st2][st1][st0][u16 words in decode order]; the decoder renormalises BEFORE
// each bit.  Chunks here are at most one block long (the API rejects larger chunks for this coder), so per chunk the
// payload is exactly what ansbc returns for that slice.  Raw rule: see oracle/trc_oracle.c (totals >= length raw).
//
// Same two-kernel shape as the adaptive CDF rANS: the model pass walks the chunk forward (eight nodes per byte,
// known from the byte: one batch of reads, one of writes; model = 256 x u16 per lane in LDS, [context][lane]) and
// streams 16 B of records per input byte to HBM scratch; the coding pass pops them in reverse.  The divisor of an
// rANS step is the record's probability, different every bit: f32 estimate + exact correction (st < 2^31).
#include "trc_io.h"
#include "trc_lane_io.h"
#include "trc_launch.h"

#define ANSB_MODEL_BYTES (256u * 64u * 2u)                 // [ctx][lane] u16
#define ANSB_CODE_LDS    (TRC_TILE_BYTES + TRC_SRING_BYTES)

// bit 1: p += (2^15 - p) >> 5, bit 0: p -= p >> 5 -- as mask arithmetic (written as ?: the compiler made a divergent if / else of it,
// eight per byte)
__device__ __forceinline__ u32 ansb_adapt(u32 p, u32 bit)
{
    const u32 m = 0u - bit, nm = ~m;                                   // bit 1: m all ones
    const u32 s = ((p ^ m) + (m & (TRC_PROB_ONE + 1u))) >> 5;          // (bit ? 2^15 - p : p) >> 5
    return p + ((s ^ nm) - nm);                                         // bit ? p + s : p - s
}

// ------------------------------------------------------------------------------ encode, pass 1 ---
__global__ __launch_bounds__(64 * TRC_WPG) void trc_ansb_model_kernel(
    const u8 *__restrict__ in, u64 n, u32 chunk, u32 nchunks, u8 *__restrict__ recs)
{
    TRC_QUAD_PROLOGUE(ANSB_MODEL_BYTES);
    u16 *mb = (u16 *)smem + lane;                              // mb[ctx * 64]
    for (u32 i = 0; i < 256; i++) mb[i * 64] = (u16)(TRC_PROB_ONE >> 1);

    WaveChunks wc;
    wc.c0 = grp_ * 64u; wc.chunk = chunk; wc.nchunks = nchunks;
    wc.lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    wc.rows = nchunks - wc.c0 < 64u ? nchunks - wc.c0 : 64u;
    WaveChunks wr = wc;                                        // the same chunks in record space (16 B per byte)
    wr.chunk = 16u * chunk; wr.lastlen = 16u * wc.lastlen;
    const bool alive = lane < wc.rows;
    const u32 len = alive ? wc.len_of(lane) : 0u;

    QuadIn qin; qin.base = in + (u64)wc.c0 * chunk;
    QuadOut qout; qout.base = recs + (u64)wc.c0 * wr.chunk;

    // one byte -> eight 16-bit records, packed two per dword in push order (bit 7 first)
    auto byte_records = [&](u32 x) -> uint4 {
        const u32 path = 0x100u | x;
        u32 r[8];
#pragma unroll
        for (int k = 0; k < 8; k++) r[k] = mb[(path >> (8 - k)) * 64];          // node of bit 7-k
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const u32 bit = (x >> (7 - k)) & 1u;
            mb[(path >> (8 - k)) * 64] = (u16)ansb_adapt(r[k], bit);
            r[k] |= bit << TRC_PROB_BITS;
        }
        return make_uint4(r[0] | r[1] << 16, r[2] | r[3] << 16, r[4] | r[5] << 16, r[6] | r[7] << 16);
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
            const u32 p0 = s * TRC_SEG + k * 16u;
            if (!__ballot(alive && p0 < len)) continue;
#pragma nounroll
            for (u32 d = 0; d < 4; d++) {                      // 4 input bytes -> 32 records = one 64-byte record segment
                const u32 w = v.x; v.x = v.y; v.y = v.z; v.z = v.w;
#pragma unroll
                for (int i = 0; i < 4; i++) qout.put((u32)i, byte_records((w >> (8 * i)) & 255u));
                qout.flush(wr, (p0 + 4u * d) * 16u);           // (records of bytes past a ragged end are never read)
            }
        }
    }
}

// ------------------------------------------------------------------------------ encode, pass 2 ---
__device__ __forceinline__ void ansb_put(u32 &st, u32 rec, StreamOut<true> &so)
{
    const u32 bit = rec >> TRC_PROB_BITS, p0 = rec & (TRC_PROB_ONE - 1);
    const u32 ls = bit ? p0 : TRC_PROB_ONE - p0;                         // ecbe_: the coded interval's width
    const bool emit = st >= (ls << 16);
    so.put16_if(emit, st);
    st = emit ? st >> 16 : st;
    u32 q = (u32)((float)st * __builtin_amdgcn_rcpf((float)ls));          // st/ls within +-1
    u32 r = st - __umul24(q, ls);                                        // q < 2^16+1, ls < 2^15
    if ((int)r < 0) { q--; r += ls; }
    if (r >= ls) { q++; r -= ls; }
    st = st + __umul24(q, TRC_PROB_ONE - ls) + (bit ? 0u : p0);
}

__global__ __launch_bounds__(64 * TRC_WPG) void trc_ansb_code_kernel(
    const u8 *__restrict__ recs, u64 n, u32 chunk, u32 nchunks,
    u8 *__restrict__ scratch, u32 stride, u32 *__restrict__ clen, u32 *__restrict__ gsum)
{
    TRC_QUAD_PROLOGUE(ANSB_CODE_LDS);
    WaveChunks wc;
    wc.c0 = grp_ * 64u; wc.chunk = chunk; wc.nchunks = nchunks;
    wc.lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    wc.rows = nchunks - wc.c0 < 64u ? nchunks - wc.c0 : 64u;
    WaveChunks wr = wc; wr.chunk = 16u * chunk; wr.lastlen = 16u * wc.lastlen;
    const bool alive = lane < wc.rows;
    const u32 c = wc.c0 + lane;
    const u32 len = alive ? wc.len_of(lane) : 0u;

    TileIn tin; tin.tile = smem; tin.base = recs + (u64)wc.c0 * wr.chunk;
    StreamOut<true> so;
    so.rings = smem + TRC_TILE_BYTES;
    so.scratch = scratch; so.stride = stride; so.c0 = wc.c0; so.wpos = 0; so.nfl = 0;
    u32 st[4] = { TRC_ANS_LOW, TRC_ANS_LOW, TRC_ANS_LOW, TRC_ANS_LOW };
    bool ovf = false;

    const u32 S = wr.chunk / TRC_SEG;                          // record segments of a full chunk: 4 input bytes each
    const u32 top = alive ? (len - 1u) / 4u : 0u;              // segment holding the chunk's last byte
    const u32 topbytes = len - 4u * top;                       // bytes of the chunk in it (1..4)
    tin.issue(wr, (S - 1u) * TRC_SEG);
    for (u32 s = S - 1u;; s--) {
        tin.commit();
        if (s) tin.issue(wr, (s - 1u) * TRC_SEG);
        const bool act = alive && s <= top && !ovf;
        const u32 nb = (act && s == top) ? topbytes : 4u;
        if (act) {
            const uint4 q[4] = { tin.read(0), tin.read(1), tin.read(2), tin.read(3) };
            const u32 *rr = (const u32 *)q;                    // rr[2*byte + ...]: 4 dwords per byte, 2 records per dword
#pragma unroll
            for (int i = 31; i >= 0; i--) {                    // record i of the segment: byte i/8, push position i%8
                if ((u32)(i >> 3) < nb) {
                    const u32 rec = (rr[i >> 1] >> (16 * (i & 1))) & 0xffffu;
                    ansb_put(st[(~i) & 3], rec, so);           // last record first on state 0: state = (31 - i) & 3
                }
            }
        }
        so.drain(false, alive);                                // <= 64 new bytes (32 records) per lane
        ovf = ovf || (alive && so.wpos + 16u >= len);          // with the four states it cannot end below len any more
        if (s == 0) break;
    }
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
__global__ __launch_bounds__(64 * TRC_WPG) void trc_ansb_dec_kernel(
    const u8 *__restrict__ payload, const u32 *__restrict__ clen, const u64 *__restrict__ goff, const u32 *__restrict__ gsum,
    u64 n, u32 chunk, u32 nchunks, u8 *__restrict__ out)
{
    TRC_QUAD_PROLOGUE(ANSB_MODEL_BYTES);
    u16 *mb = (u16 *)smem + lane;
    for (u32 i = 0; i < 256; i++) mb[i * 64] = (u16)(TRC_PROB_ONE >> 1);

    WaveChunks wc;
    wc.c0 = grp_ * 64u; wc.chunk = chunk; wc.nchunks = nchunks;
    wc.lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    wc.rows = nchunks - wc.c0 < 64u ? nchunks - wc.c0 : 64u;
    const bool alive = lane < wc.rows;
    const u32 c = wc.c0 + lane;
    const u32 len = alive ? wc.len_of(lane) : 0u;
    const u32 cl = alive ? trc_min(clen[c], len) : 0u;        // a directory entry above the chunk length (corrupt input) reads as raw
    const u32 ex = trc_wave_incl_scan(cl) - cl;
    const u64 off = trc_group_base(goff, gsum, wc.c0 >> 6) + ex;
    const bool coded = alive && cl != len;

    u32 st[4] = { TRC_ANS_LOW, TRC_ANS_LOW, TRC_ANS_LOW, TRC_ANS_LOW };
    if (coded) for (u32 k = 0; k < 4; k++) st[k] = trc_ld32_a2(payload + off + 4u * k);   // decoder st[i] = encoder st[3-i]
    // the stream side: four units of look-ahead in two registers, one 16-byte load per FOUR bits (<= 4 units), trc_lane_io.h LaneLook16
    // (round 4; rounds 2-3 selected every unit out of a 32-byte register window: ten instructions per bit).  No predicate on "this
    // lane is decoding": a lane that is not -- raw chunk, dead lane, past the end of a short last chunk -- runs along on its own
    // registers and model column, nothing of it is stored.
    LaneLook16 sl; sl.prime(payload + off + 16u, trc_sub_sat(cl, 16u));

    auto get_byte = [&](bool) -> u32 {
        u32 ctx = 1;
        u32 p = mb[64];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint4 W = sl.fetch();
            u32 cnt = 0;
#pragma unroll
            for (int jj = 0; jj < 4; jj++) {
                const int j = 4 * h + jj;
                // both children are requested before this bit is known (below the last level the index wraps, values unused)
                const u32 lc = (2u * ctx) & 255u;
                const u32 pl = mb[lc * 64], pr = mb[(lc + 1u) * 64];
                u32 s = st[j & 3];
                if (jj == 0) sl.renorm<0>(s, cnt); else if (jj == 1) sl.renorm<1>(s, cnt);      // ecdnorm before the step
                else if (jj == 2) sl.renorm<2>(s, cnt); else sl.renorm<3>(s, cnt);
                const u32 r = s & (TRC_PROB_ONE - 1), rcx = __umul24(s >> TRC_PROB_BITS, p);   // s >> 15 < 2^17, p < 2^15: the product fits 32 bits
                const u32 m = (u32)((int)(r - p) >> 31);       // ecbd: bit = r < p, as a mask (r, p < 2^15); everything below selects under it
                st[j & 3] = trc_bfi(m, rcx + r, s - rcx - p);
                mb[ctx * 64] = (u16)ansb_adapt(p, m & 1u);
                ctx = ctx * 2 - m;
                p = trc_bfi(m, pr, pl);
            }
            sl.end_group(cnt, W);
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
    trc_wave_copy_raw(__ballot(alive && cl == len && len != 0), off, len, out + (u64)wc.c0 * chunk, chunk, payload);
}

// ------------------------------------------------------------------------------------- launch ---
void trc_launch_ansb_enc(const uint8_t *d_in, size_t n, uint32_t chunk, const TrcWork &w, uint32_t *d_clen, hipStream_t s)
{
    TRC_RAISE_LDS_ONCE(trc_ansb_model_kernel, TRC_WPG * ANSB_MODEL_BYTES);
    TRC_LAUNCH_TIMED(trc_ansb_model_kernel, TRC_QUAD_GRID(w.ngroups), dim3(64 * TRC_WPG), TRC_WPG * (ANSB_MODEL_BYTES), s, d_in, (u64)n, chunk, w.nchunks, w.scratch2);
    TRC_LAUNCH_TIMED(trc_ansb_code_kernel, TRC_QUAD_GRID(w.ngroups), dim3(64 * TRC_WPG), TRC_WPG * (ANSB_CODE_LDS), s,
                       (const u8 *)w.scratch2, (u64)n, chunk, w.nchunks, w.scratch, w.stride, d_clen, w.gsum);
}
void trc_launch_ansb_dec(const uint8_t *d_payload, const uint32_t *d_clen, size_t n, uint32_t chunk,
                         const TrcWork &w, uint8_t *d_out, hipStream_t s)
{
    TRC_RAISE_LDS_ONCE(trc_ansb_dec_kernel, TRC_WPG * ANSB_MODEL_BYTES);
    TRC_LAUNCH_TIMED(trc_ansb_dec_kernel, TRC_QUAD_GRID(w.ngroups), dim3(64 * TRC_WPG), TRC_WPG * (ANSB_MODEL_BYTES), s,
                     d_payload, d_clen, w.goff, w.gsum, (u64)n, chunk, w.nchunks, d_out);
}
