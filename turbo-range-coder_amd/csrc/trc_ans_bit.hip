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
// streams 16 B of records per input byte to HBM scratch; the coding pass pops them in reverse -- four lanes per chunk, one
// per rANS state (round 4), the records laid out per state by the model pass.  The divisor of an rANS step is the record's
// probability, different every bit: f32 estimate of the quotient + exact correction (trc_rans_step, trc_dev.h).
#include "trc_io.h"
#include "trc_lane_io.h"
#include "trc_launch.h"

#define ANSB_MODEL_BYTES (256u * 64u * 2u)                 // [ctx][lane] u16

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
    wr.chunk = 16u * chunk; wr.lastlen = 16u * ((wc.lastlen + 3u) & ~3u);     // whole groups of four bytes: a group's pieces are per STATE, not per byte
    const bool alive = lane < wc.rows;
    const u32 len = alive ? wc.len_of(lane) : 0u;

    QuadIn qin; qin.base = in + (u64)wc.c0 * chunk;
    QuadOut qout; qout.base = recs + (u64)wc.c0 * wr.chunk;

    // one byte -> eight 16-bit records (push order: bit 7 first), packed per rANS state: the last record pushed goes to state 0,
    // dword s = record of push position 7 - s | record of position 3 - s << 16.  The two nodes of a dword adapt together in packed
    // 16-bit arithmetic (bit 1: p += (2^15 - p) >> 5, bit 0: p -= p >> 5 under the halves' masks): 12 operations per pair, 18 as scalars.
    typedef unsigned short trc_us2 __attribute__((ext_vector_type(2)));
    auto byte_records = [&](u32 x) -> uint4 {
        const u32 path = 0x100u | x;
        u32 r[8];
#pragma unroll
        for (int k = 0; k < 8; k++) r[k] = mb[(path >> (8 - k)) * 64];          // node of bit 7-k
        const u32 y = x | x << 12;                             // bit s: the bit of position 7 - s, bit 16 + s: of position 3 - s
        u32 D[4];
#pragma unroll
        for (int sidx = 0; sidx < 4; sidx++) {
            const u32 P = r[7 - sidx] | r[3 - sidx] << 16;
            const u32 B = (y >> sidx) & 0x10001u;
            const u32 M = __umul24(B, 0xffffu), NM = ~M;       // all ones in the halves whose bit is 1
            const trc_us2 t = __builtin_bit_cast(trc_us2, P ^ M) + __builtin_bit_cast(trc_us2, M & 0x80018001u);     // bit ? 2^15 - p : p
            const u32 sft = __builtin_bit_cast(u32, t >> (trc_us2)5);
            const trc_us2 np = __builtin_bit_cast(trc_us2, P) + (__builtin_bit_cast(trc_us2, sft ^ NM) - __builtin_bit_cast(trc_us2, NM));
            const u32 NP = __builtin_bit_cast(u32, np);
            mb[(path >> (1 + sidx)) * 64] = (u16)NP;           // node of position 7 - s
            mb[(path >> (5 + sidx)) * 64] = (u16)(NP >> 16);   // node of position 3 - s
            D[sidx] = P | B << TRC_PROB_BITS;
        }
        return make_uint4(D[0], D[1], D[2], D[3]);
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
                // the coding pass works with four lanes per chunk, lane s on state s = push positions 7 - s and 3 - s of every
                // byte: piece s of the segment holds exactly that lane's records, one dword per byte
                uint4 R[4];
#pragma unroll
                for (int i = 0; i < 4; i++) R[i] = byte_records((w >> (8 * i)) & 255u);
                qout.put(0, make_uint4(R[0].x, R[1].x, R[2].x, R[3].x)); qout.put(1, make_uint4(R[0].y, R[1].y, R[2].y, R[3].y));
                qout.put(2, make_uint4(R[0].z, R[1].z, R[2].z, R[3].z)); qout.put(3, make_uint4(R[0].w, R[1].w, R[2].w, R[3].w));
                qout.flush(wr, (p0 + 4u * d) * 16u);           // (records of bytes past a ragged end are never read)
            }
        }
    }
}

// ------------------------------------------------------------------------------ encode, pass 2 ---
// Four lanes per chunk (round 4; the scheme of trc_ansa_codeq_kernel, trc_ans_adaptive.hip): the four rANS states of ansbc are
// independent chains that share the ORDER of their 16-bit words.  Lanes 4i .. 4i + 3 take chunk i of the wave, lane s state s;
// a step is four consecutive records (descending record order = ascending lane order: the last record pushed is state 0's), the
// quad's emit flags come from one __ballot, a lane's word goes to the shared stream position + 2 x (emits of the lanes before it)
// in the chunk's ring (StreamOut<DOWN, false, QUAD>), a lane with nothing to store writes to its own dummy slot.  The model pass
// leaves every lane's records contiguous (16 B per four input bytes), so they come straight from HBM into registers.
// ecbe_ (anscdf.c:659-668) is a plain rANS step with frequency ls = bit ? p : 2^15 - p and start bit ? 0 : p; its divisor is
// different every bit: f32 estimate + exact correction (st < 2^31).  Raw rule: words + 16 >= len (the reference has no test
// before the end of a block; the words only grow, so a chunk that is there already stops coding).
#define ANSBQ_WAVE_LDS  (16u * TRC_SRING_STRIDE + 256u)        // 16 rings + 16 x 4 dummy slots
#define ANSBQ_LDS(GPW)  (4u * (GPW) * ANSBQ_WAVE_LDS + 64u + 64u)              // + the waves' byte counts + TrcPace's progress counters
// GPW groups of 64 chunks per workgroup: 1, or 4 with TrcPace when the launch is one residency round (trc_ansa_codeq_kernel)
template <int GPW>
__global__ __launch_bounds__(256 * GPW) void trc_ansb_codeq_kernel(
    const u8 *__restrict__ recs, u64 n, u32 chunk, u32 nchunks,
    u8 *__restrict__ scratch, u32 stride, u32 *__restrict__ clen, u32 *__restrict__ gsum)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem_wg_[];
    const u32 wv = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const u32 lane = trc_lane(), s = lane & 3u, ci = lane >> 2;
    u8 *const smem = smem_wg_ + wv * ANSBQ_WAVE_LDS;
    u32 *const wsum = (u32 *)(smem_wg_ + 4u * GPW * ANSBQ_WAVE_LDS);
    TrcPace pace; pace.init(trc_lds_addr(smem_wg_) + 4u * GPW * ANSBQ_WAVE_LDS + 64u, threadIdx.x, wv);
    if (GPW > 1) __syncthreads();
    const u32 cw0 = blockIdx.x * (64u * GPW) + wv * 16u;       // this wave's first chunk
    const u32 c = cw0 + ci;
    const bool alive = c < nchunks;
    const u32 lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    const u32 len = alive ? (c == nchunks - 1u ? lastlen : chunk) : 0u;
    const u8 *rbase = recs + (u64)(alive ? c : 0u) * (16u * (u64)chunk) + 16u * s;

    StreamOut<true, false, true, true> so;
    so.rings = smem;
    so.scratch = scratch; so.stride = stride; so.c0 = cw0; so.wpos = 0; so.nfl = 0;
    u32 st = TRC_ANS_LOW;
    bool ovf = false;
    const u32 sh = lane & ~3u, below = (1u << s) - 1u;
    const u32 ringw = trc_lds_addr(so.rings) + ci * TRC_SRING_STRIDE;
    const u32 dummy = trc_lds_addr(smem) + 16u * TRC_SRING_STRIDE + 16u * ci + 4u * s;

    const u32 T = chunk / 4u;                                  // groups (4 input bytes = 32 records) of a full chunk
    const u32 top = len ? (len - 1u) / 4u : 0u;                // group holding the chunk's last byte
    const u32 topbytes = len - 4u * top;
    uint4 nx = make_uint4(0, 0, 0, 0);
    if (alive && len && T - 1u <= top) nx = trc_ld16_nt(rbase + (size_t)(T - 1u) * 64u);
    for (u32 t = T - 1u;; t--) {
        if (GPW > 1 && !(t & 15u)) pace.step(T - t);           // (every 64 input bytes)
        const uint4 q = nx;
        if (t && alive && len && t - 1u <= top) nx = trc_ld16_nt(rbase + (size_t)(t - 1u) * 64u);
        const bool act = alive && len != 0u && t <= top;
        const u32 nb = act ? (t == top ? topbytes : 4u) : 0u;
        ovf = ovf || (act && so.wpos + 16u >= len);            // with the four states it cannot end below len any more
        const u32 d[4] = { q.x, q.y, q.z, q.w };
        u32 wh = so.wpos >> 1;                                 // 16-bit units appended so far
#pragma unroll
        for (int b = 3; b >= 0; b--) {
            const bool go = (u32)b < nb && !ovf;               // (the same for the four lanes of a chunk)
            const u32 go15 = go ? 15u : 0u;
#pragma unroll
            for (int h = 0; h < 2; h++) {                      // push positions 7 - s, then 3 - s
                const u32 p0 = h ? __builtin_amdgcn_ubfe(d[b], 16, 15) : d[b] & 0x7fffu;
                const u32 m = h ? (u32)((int)d[b] >> 31) : (u32)__builtin_amdgcn_sbfe((int)d[b], 15, 1);   // bit 1: all ones
                const u32 np0 = TRC_PROB_ONE - p0;
                const u32 f = trc_bfi(m, p0, np0), g = trc_bfi(m, np0, p0), c0 = trc_bfi(m, 0u, p0);
                // (the vote is taken on the bare compare and masked with `go` as a value: a vote on `go && compare` makes the compiler
                // rebuild the lane mask through a 0 / 1 vector, two instructions per step)
                const bool ge = st >= (f << 16), emit = go && ge;
                const u32 q4 = (u32)(__ballot(ge) >> sh) & go15;
                const u32 u = wh + (u32)__builtin_popcount(q4 & below);               // this lane's unit: the u-th of the chunk
                const u32 at = ringw + 2u * (~u & (TRC_SRING / 2u - 1u));            // (ring offset of unit u: -(2u + 2) mod the ring)
                trc_lds_write16(emit ? at : dummy, st);
                wh += (u32)__builtin_popcount(q4);
                const u32 s1 = emit ? st >> 16 : st;
                st = go ? trc_rans_step(s1, f, g, c0) : st;
            }
        }
        so.wpos = wh << 1;
        so.drain(false, alive);                                // <= 64 new bytes (32 records) per chunk
        if (t == 0) break;
    }
    u32 out_len = 0;
    if (alive) {
        if (!ovf) {                                            // states 0..3, high half first, each below the one before
            so.put16_at(true, so.wpos + 4u * s, st >> 16); so.put16_at(true, so.wpos + 4u * s + 2u, st);
            so.wpos += 16u;
            if (so.wpos >= len) ovf = true;
        }
        out_len = ovf ? len : so.wpos;
    }
    so.drain(true, alive && !ovf);
    if (alive && s == 0u) clen[c] = out_len;
    const u32 ws = trc_wave_sum(s == 0u ? out_len : 0u);
    if (lane == 0) wsum[wv] = ws;
    __syncthreads();
    if (lane == 0 && !(wv & 3u) && (blockIdx.x * GPW + (wv >> 2)) * 64u < nchunks) gsum[blockIdx.x * GPW + (wv >> 2)] = wsum[wv] + wsum[wv + 1] + wsum[wv + 2] + wsum[wv + 3];
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
    static const int gpw_env = getenv("TRC_CODEQ_GPW") ? atoi(getenv("TRC_CODEQ_GPW")) : 0;     // tuning aid: 1 / 4 force the workgroup shape
    if (gpw_env ? gpw_env == 4 : (w.ngroups >= 512u && w.ngroups <= 4u * 256u)) {
        // (not padded to one workgroup per CU as the anscdf pass is: this pass streams 16 bytes of records per input byte and runs no
        // worse with whatever the dispatcher does -- 1.18-1.31 ms unpadded, 1.29-1.35 padded, profiles/r05_notes.md)
        TRC_RAISE_LDS_ONCE(trc_ansb_codeq_kernel<4>, ANSBQ_LDS(4));
        TRC_LAUNCH_TIMED(trc_ansb_codeq_kernel<4>, dim3((w.ngroups + 3u) / 4u), dim3(1024), ANSBQ_LDS(4), s,
                           (const u8 *)w.scratch2, (u64)n, chunk, w.nchunks, w.scratch, w.stride, d_clen, w.gsum);
    } else
        TRC_LAUNCH_TIMED(trc_ansb_codeq_kernel<1>, dim3(w.ngroups), dim3(256), ANSBQ_LDS(1), s,
                           (const u8 *)w.scratch2, (u64)n, chunk, w.nchunks, w.scratch, w.stride, d_clen, w.gsum);
}
void trc_launch_ansb_dec(const uint8_t *d_payload, const uint32_t *d_clen, size_t n, uint32_t chunk,
                         const TrcWork &w, uint8_t *d_out, hipStream_t s)
{
    TRC_RAISE_LDS_ONCE(trc_ansb_dec_kernel, TRC_WPG * ANSB_MODEL_BYTES);
    TRC_LAUNCH_TIMED(trc_ansb_dec_kernel, TRC_QUAD_GRID(w.ngroups), dim3(64 * TRC_WPG), TRC_WPG * (ANSB_MODEL_BYTES), s,
                     d_payload, d_clen, w.goff, w.gsum, (u64)n, chunk, w.nchunks, d_out);
}
