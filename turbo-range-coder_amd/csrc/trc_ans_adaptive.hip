// trc_ans_adaptive.hip -- the adaptive-CDF rANS coders (CDF16 model, trc_nibmodel.h):
//   TRC_ANSA   anscdfenc / anscdfdec    anscdf.c:567-605  `turborc -e56/57/58`     bytes, 4 interleaved states
//   TRC_ANSA4  anscdf4enc / anscdf4dec  anscdf.c:87-133   `turborc -n -e56/57/58`  nibbles (values 0..15), 2 states
// Per chunk the payload is exactly what the reference function returns for that slice (a chunk is far below the
// reference's 4 MiB block, so it is one block); mnenc4/mnenc8x2/mnflush/mndec4/mndec8x2 anscdf_.h:106-162.
//     ANSA   [u32 st3][u32 st2][u32 st1][u32 st0][u16 renorm words in decode order]     (raw if it does not fit)
//     ANSA4  [u32 st1][u32 st0][u16 renorm words in decode order]
//
// The encoder is inherently two-pass (the model adapts forward, rANS codes backward):
//   pass 1  trc_ansa_model_kernel : walk the chunk forward through the adaptive model and record
//           {cdf_lo << 15 | freq} per nibble.  ANSA: 4 records per byte pair in the reference's push order
//           (x0.hi -> state 3, x0.lo -> 2, x1.hi -> 1, x1.lo -> 0; an odd tail byte pairs with a CODED dummy 0).
//           ANSA4: one record per input value; groups of 4 alternate states 1,0,1,0, the n%4 tail uses state 0.
//           Records stream to HBM scratch as uniform 64-byte segments (8 resp. 4 B per input byte: the reference
//           keeps the same stack on the heap, anscdf.c:112,570).  Model-bound: the model is the only thing in LDS
//           (bytes and records move through in-register quad transposes), four waves per CU.
//   pass 2  trc_ansa_code_kernel  : pop the records in reverse, one rANS step each; words grow downward from the
//           end of the chunk's scratch region.  The divisor changes every symbol, so st/f is an f32 estimate plus
//           exact correction (st < 2^31).
//   Raw rule (mnflush): before EVERY record the reference tests ep <= op + 2 + 4*states; the test is monotone.
// The decoders read the stream through a 16-byte register window per lane (trc_lane_io.h).  ANSA4's decoder takes
// the n%4 tail from the state the ENCODER used (the reference decoder's tail reads the other state and does not
// round-trip: see oracle/trc_oracle.c orc_anscdf4dec).
#include <stdlib.h>
#include "trc_io.h"
#include "trc_lane_io.h"
#include "trc_nibmodel.h"
#include "trc_launch.h"

#ifndef ANSA_BATCH
#define ANSA_BATCH 4                                           // bytes per record_bytes call of the model pass (divides 8)
#endif
#define ANSA_MODEL_LDS(NIB) ((NIB) ? TRC_NIB1_BYTES : TRC_NIB_BYTES)
#define ANSA_CODE_LDS       (TRC_TILE_BYTES + TRC_SRING_BYTES)

// geometry of the record space of one wave's chunks (record bytes per input byte: 8 / 4)
template <bool NIB>
__device__ __forceinline__ WaveChunks ansa_record_space(const WaveChunks &wc)
{
    WaveChunks wr = wc;
    wr.chunk = (NIB ? 4u : 8u) * wc.chunk;
    wr.lastlen = NIB ? ((4u * wc.lastlen + 15u) & ~15u) : 8u * (wc.lastlen + (wc.lastlen & 1u));
    return wr;
}

// PLANAR record space (round 4, the two-wave model pass of ANSA): per block of 16 input bytes 64 B of hi records, then 64 B of lo
// records -- each wave of the model pass writes whole 64-byte segments of its own plane
__device__ __forceinline__ WaveChunks ansa_record_space_planar(const WaveChunks &wc)
{
    WaveChunks wr = wc;
    wr.chunk = 8u * wc.chunk;
    wr.lastlen = 128u * ((wc.lastlen + 15u) / 16u);
    return wr;
}

// ------------------------------------------------------------------------------ encode, pass 1 ---
template <bool NIB>
__global__ __launch_bounds__(64 * (NIB ? TRC_NIB_WPG : TRC_WPG)) void trc_ansa_model_kernel(
    const u8 *__restrict__ in, u64 n, u32 chunk, u32 nchunks, u8 *__restrict__ recs)
{
    TRC_QUAD_PROLOGUE(ANSA_MODEL_LDS(NIB));
    NibModel<NIB ? 1 : 17> m; m.init(smem);

    WaveChunks wc;
    wc.c0 = grp_ * 64u; wc.chunk = chunk; wc.nchunks = nchunks;
    wc.lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    wc.rows = nchunks - wc.c0 < 64u ? nchunks - wc.c0 : 64u;
    const WaveChunks wr = ansa_record_space<NIB>(wc);
    const bool alive = lane < wc.rows;
    const u32 len = alive ? wc.len_of(lane) : 0u;

    QuadIn qin; qin.base = in + (u64)wc.c0 * chunk;
    QuadOut qout; qout.base = recs + (u64)wc.c0 * wr.chunk;
    NibTable T0 = m.load(m.table(0));                          // the hi table (the only table of the nibble coder): registers, see record_bytes

    const u32 S = chunk / TRC_SEG;
    qin.issue(wc, 0);
    for (u32 s = 0; s < S; s++) {
        TRC_PACE_STEP(s + 1u);
        qin.commit();
        if (s + 1 < S) qin.issue(wc, (s + 1) * TRC_SEG);
        uint4 pc0 = qin.read(0), pc1 = qin.read(1), pc2 = qin.read(2), pc3 = qin.read(3);
#pragma nounroll
        for (u32 k = 0; k < 4; k++) {
            const uint4 v = pc0; pc0 = pc1; pc1 = pc2; pc2 = pc3;
            const u32 p0 = s * TRC_SEG + k * 16u;
            if (!__ballot(alive && p0 < len)) continue;
            const u32 w[4] = { v.x, v.y, v.z, v.w };
            if (!NIB) {
#pragma unroll
                for (int h = 0; h < 2; h++) {                  // 8 input bytes -> 16 records = one 64-byte record segment
                    u32 r[16];
#pragma unroll
                    for (int q = 0; q < 8 / ANSA_BATCH; q++) {
                        u32 x[ANSA_BATCH], rr[2 * ANSA_BATCH];
#pragma unroll
                        for (int i = 0; i < ANSA_BATCH; i++) {
                            const int b = q * ANSA_BATCH + i;
                            x[i] = (w[2 * h + (b >> 2)] >> (8 * (b & 3))) & 255u;
                            if (p0 + 8u * (u32)h + (u32)b >= len) x[i] = 0;  // the coded dummy of an odd tail (and unused padding)
                        }
                        m.template record_bytes<ANSA_BATCH>(T0, x, rr);
#pragma unroll
                        for (int i = 0; i < 2 * ANSA_BATCH; i++) r[2 * q * ANSA_BATCH + i] = rr[i];
                    }
#pragma unroll
                    for (int j = 0; j < 4; j++) qout.put((u32)j, make_uint4(r[4 * j], r[4 * j + 1], r[4 * j + 2], r[4 * j + 3]));
                    qout.flush(wr, (p0 + 8u * (u32)h) * 8u);
                }
            } else {
                u32 r[16], x[16];                              // 16 input values -> 16 records
#pragma unroll
                for (int i = 0; i < 16; i++) x[i] = (w[i >> 2] >> (8 * (i & 3))) & 15u;
                m.template record_nibs<16>(T0, x, r);
#pragma unroll
                for (int j = 0; j < 4; j++) qout.put((u32)j, make_uint4(r[4 * j], r[4 * j + 1], r[4 * j + 2], r[4 * j + 3]));
                qout.flush(wr, p0 * 4u);
            }
        }
    }
}

// Pass 1 of ANSA as TWO WAVES per 64 chunks (round 4).  The hi record of a byte depends on the hi table alone, the lo record on
// the lo tables alone (the lo table is SELECTED by the hi nibble, which is input, not model state): wave 0 walks the hi nibbles,
// wave 1 the lo nibbles of the same bytes, on disjoint parts of the same LDS rows, with nothing to tell each other -- no barrier.
// Two waves per SIMD on the LDS footprint of one: each wave's issue gaps (a lone wave issues every ~1.5 quad-cycles) are the
// other's slots.  The record space is PLANAR (above) so that both write whole segments.
__global__ __launch_bounds__(128 * TRC_WPG) void trc_ansa_model2_kernel(
    const u8 *__restrict__ in, u64 n, u32 chunk, u32 nchunks, u8 *__restrict__ recs, const u32 *__restrict__ gate, u32 gate_part)
{
    // a workgroup = 4 hi waves (0-3) + 4 lo waves (4-7): pair k = waves k and k + 4 on SIMD k (trc_dev.h, TRC_WPG)
    extern __shared__ __attribute__((aligned(16))) u8 smem_wg_[];
    const u32 wv_ = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool lo_wave = wv_ >= TRC_WPG;
    const u32 grp_ = blockIdx.x * TRC_WPG + (wv_ & (TRC_WPG - 1u));
    // round 5: the pair keeps one pace (TrcPace, trc_dev.h) -- the hi wave is the older of the two and has less to do per byte; left
    // alone it ends early and the lo wave finishes the chunk at a lone wave's issue rate.  Walking together they also ask for the
    // same input lines at the same time (the second request finds the line in the L2).
    TrcPace pace; pace.init(trc_lds_addr(smem_wg_) + TRC_WPG * ANSA_MODEL_LDS(false), threadIdx.x, wv_);
    __syncthreads();
    if (grp_ >= (nchunks + 63u) / 64u) return;
    u8 *const smem = smem_wg_ + (wv_ & (TRC_WPG - 1u)) * ANSA_MODEL_LDS(false);
    const u32 lane = trc_lane();
    NibModel<17> m;
    if (lo_wave) m.init_part(smem, 1u, 16u); else m.init_part(smem, 0u, 1u);       // each wave its own tables (and the same K)

    WaveChunks wc;
    wc.c0 = grp_ * 64u; wc.chunk = chunk; wc.nchunks = nchunks;
    wc.lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    wc.rows = nchunks - wc.c0 < 64u ? nchunks - wc.c0 : 64u;
    wc.gate = gate; wc.gate_part = gate_part;                  // (host-pointer encodes: the input arrives while the waves code, trc_io.h)
    const WaveChunks wr = ansa_record_space_planar(wc);
    const bool alive = lane < wc.rows;
    const u32 len = alive ? wc.len_of(lane) : 0u;

    QuadIn qin; qin.base = in + (u64)wc.c0 * chunk;
    QuadOut qout; qout.base = recs + (u64)wc.c0 * wr.chunk;
    NibTable T0 = m.load(m.table(0));                          // (hi wave)

    const u32 S = chunk / TRC_SEG;
    // (TRC_ANSA_M2_PAIR: both 64-byte halves of a 128-byte input line requested together, QuadIn::take_fwd -- the pass then reads 200 MB
    // from the fabric for its two walks of 100 MB instead of ~1.8 x that, and runs 6 % SLOWER (encode 0.587 -> 0.623 ms,
    // profiles/r05k_ab.txt): the pass is not bound by its traffic.  Off.)
#ifndef TRC_ANSA_M2_PAIR
    qin.issue(wc, 0);
#else
    qin.start_fwd(wc, S);
#endif
    for (u32 s = 0; s < S; s++) {
#ifndef TRC_ANSA_M2_NOPACE
        pace.step(s + 1u);
#endif
#ifndef TRC_ANSA_M2_PAIR
        qin.commit();
        if (s + 1 < S) qin.issue(wc, (s + 1) * TRC_SEG);
#else
        qin.take_fwd(wc, s, S);
#endif
        uint4 pc0 = qin.read(0), pc1 = qin.read(1), pc2 = qin.read(2), pc3 = qin.read(3);
#pragma nounroll
        for (u32 k = 0; k < 4; k++) {
            const uint4 v = pc0; pc0 = pc1; pc1 = pc2; pc2 = pc3;
            const u32 p0 = s * TRC_SEG + k * 16u;
            if (!__ballot(alive && p0 < len)) continue;
            const u32 w[4] = { v.x, v.y, v.z, v.w };
            u32 r[16];
#pragma unroll
            for (int q = 0; q < 4; q++) {                      // 16 input bytes -> 16 records of this wave's plane
                u32 x[4], rr[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    x[i] = (w[q] >> (8 * i)) & 255u;
                    if (p0 + 4u * (u32)q + (u32)i >= len) x[i] = 0;       // the coded dummy of an odd tail (and unused padding)
                }
                // (lo records one byte at a time: no fix-ups between the bytes of a batch; the other wave hides the LDS round trips)
                if (lo_wave) {
#pragma unroll
                    for (int i = 0; i < 4; i++) { const u32 x1[1] = { x[i] }; u32 r1[1]; m.template record_lo<1>(x1, r1); rr[i] = r1[0]; }
                } else m.template record_hi<4>(T0, x, rr);
#pragma unroll
                for (int i = 0; i < 4; i++) r[4 * q + i] = rr[i];
            }
#pragma unroll
            for (int j = 0; j < 4; j++) qout.put((u32)j, make_uint4(r[4 * j], r[4 * j + 1], r[4 * j + 2], r[4 * j + 3]));
            qout.flush(wr, p0 * 8u + (lo_wave ? 64u : 0u));
        }
    }
}

// ------------------------------------------------------------------------------ encode, pass 2 ---
// one step where `act`, nothing where not (no branch: with 64 lanes some lane is always at a different point of its chunk)
__device__ __forceinline__ void ansa_put(u32 &st, u32 rec, StreamOut<true> &so, bool act = true)
{
    const u32 f = rec & 0x7fffu, c0 = rec >> 15;
    const bool emit = act && st >= (f << 16);
    so.put16_if(emit, st);
    const u32 s1 = emit ? st >> 16 : st;
    st = act ? trc_rans_step(s1, f, TRC_PROB_ONE - f, c0) : st;          // (garbage where !act: f may be 0)
}

template <bool NIB>
__global__ __launch_bounds__(64 * TRC_WPG) void trc_ansa_code_kernel(
    const u8 *__restrict__ recs, u64 n, u32 chunk, u32 nchunks,
    u8 *__restrict__ scratch, u32 stride, u32 *__restrict__ clen, u32 *__restrict__ gsum)
{
    TRC_QUAD_PROLOGUE(ANSA_CODE_LDS);
    WaveChunks wc;
    wc.c0 = grp_ * 64u; wc.chunk = chunk; wc.nchunks = nchunks;
    wc.lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    wc.rows = nchunks - wc.c0 < 64u ? nchunks - wc.c0 : 64u;
    const WaveChunks wr = ansa_record_space<NIB>(wc);
    const bool alive = lane < wc.rows;
    const u32 c = wc.c0 + lane;
    const u32 len = alive ? wc.len_of(lane) : 0u;
    const u32 nrec = NIB ? len : 2u * (len + (len & 1u));      // ANSA: 4 per byte pair
    const u32 body = len & ~3u;                                // ANSA4: records below this alternate states 1,0,1,0
    const u32 room = 2u + 4u * (NIB ? 2u : 4u);                // mnflush: ep <= op + sizeof(io_t) + states * 4  ->  raw

    TileIn tin; tin.tile = smem; tin.base = recs + (u64)wc.c0 * wr.chunk;
    StreamOut<true> so;
    so.rings = smem + TRC_TILE_BYTES;
    so.scratch = scratch; so.stride = stride; so.c0 = wc.c0; so.wpos = 0; so.nfl = 0;
    u32 st[4] = { TRC_ANS_LOW, TRC_ANS_LOW, TRC_ANS_LOW, TRC_ANS_LOW };
    bool ovf = false;

    const u32 S = wr.chunk / TRC_SEG;                          // record segments (16 records each) in a full chunk
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
                const bool can = (u32)i < hi && !ovf;
                ovf = ovf || (can && so.wpos + room >= len);
                const bool go = can && !ovf;
                if (!NIB) ansa_put(st[3 - (i & 3)], rr[i], so, go);           // record index 16*s + i, 16*s is a multiple of 4
                else if (i & 1) ansa_put(st[0], rr[i], so, go);
                else {                                         // even position: state 1 inside the body, state 0 in the tail
                    const bool one = 16u * s + (u32)i < body;
                    u32 cur = one ? st[1] : st[0];
                    ansa_put(cur, rr[i], so, go);
                    st[1] = one ? cur : st[1]; st[0] = one ? st[0] : cur;
                }
            }
        }
        so.drain(false, alive);                                // <= 32 new bytes (16 records) per lane
        if (s == 0) break;
    }
    u32 out_len = 0;
    if (alive) {
        if (!ovf) {
            for (int k = 0; k < (NIB ? 2 : 4); k++) { so.put16(st[k] >> 16); so.put16(st[k]); }
            if (so.wpos >= len) ovf = true;
        }
        out_len = ovf ? len : so.wpos;
    }
    so.drain(true, alive && !ovf);
    if (alive) clen[c] = out_len;
    const u32 gs = trc_wave_sum(out_len);
    if (lane == 0) gsum[wc.c0 >> 6] = gs;
}

// Pass 2 over the PLANAR record space (ANSA behind trc_ansa_model2_kernel): a block of 16 input bytes is two segments, hi records
// and lo records; record 2j of the block is hi[j], record 2j + 1 is lo[j].  Everything else as above.
#define ANSA_CODE_PLANAR_LDS (2u * TRC_TILE_BYTES + TRC_SRING_BYTES)
__global__ __launch_bounds__(64 * TRC_WPG) void trc_ansa_code_planar_kernel(
    const u8 *__restrict__ recs, u64 n, u32 chunk, u32 nchunks,
    u8 *__restrict__ scratch, u32 stride, u32 *__restrict__ clen, u32 *__restrict__ gsum)
{
    TRC_QUAD_PROLOGUE(ANSA_CODE_PLANAR_LDS);
    WaveChunks wc;
    wc.c0 = grp_ * 64u; wc.chunk = chunk; wc.nchunks = nchunks;
    wc.lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    wc.rows = nchunks - wc.c0 < 64u ? nchunks - wc.c0 : 64u;
    const WaveChunks wr = ansa_record_space_planar(wc);
    const bool alive = lane < wc.rows;
    const u32 c = wc.c0 + lane;
    const u32 len = alive ? wc.len_of(lane) : 0u;
    const u32 nrec = 2u * (len + (len & 1u));                  // 4 per byte pair
    const u32 room = 2u + 4u * 4u;                             // mnflush: ep <= op + sizeof(io_t) + states * 4  ->  raw

    TileIn th, tl;
    th.tile = smem; th.base = recs + (u64)wc.c0 * wr.chunk;
    tl.tile = smem + TRC_TILE_BYTES; tl.base = th.base;
    StreamOut<true> so;
    so.rings = smem + 2u * TRC_TILE_BYTES;
    so.scratch = scratch; so.stride = stride; so.c0 = wc.c0; so.wpos = 0; so.nfl = 0;
    u32 st[4] = { TRC_ANS_LOW, TRC_ANS_LOW, TRC_ANS_LOW, TRC_ANS_LOW };
    bool ovf = false;

    const u32 T = wr.chunk / 128u;                             // blocks (32 records each) in a full chunk
    const u32 top = alive && nrec ? (nrec - 1u) / 32u : 0u;
    th.issue(wr, (T - 1u) * 128u); tl.issue(wr, (T - 1u) * 128u + 64u);
    for (u32 t = T - 1u;; t--) {
        th.commit(); tl.commit();
        if (t) { th.issue(wr, (t - 1u) * 128u); tl.issue(wr, (t - 1u) * 128u + 64u); }
        const bool act = alive && nrec != 0u && t <= top && !ovf;
#pragma unroll
        for (int half = 1; half >= 0; half--) {                // records 31..16, then 15..0 of the block
            if (act) {
                const uint4 qh[2] = { th.read(2u * (u32)half), th.read(2u * (u32)half + 1u) };
                const uint4 ql[2] = { tl.read(2u * (u32)half), tl.read(2u * (u32)half + 1u) };
                const u32 *hh = (const u32 *)qh, *ll = (const u32 *)ql;
#pragma unroll
                for (int i = 15; i >= 0; i--) {
                    const bool can = 32u * t + 16u * (u32)half + (u32)i < nrec && !ovf;
                    ovf = ovf || (can && so.wpos + room >= len);
                    const bool go = can && !ovf;
                    ansa_put(st[3 - (i & 3)], (i & 1) ? ll[i >> 1] : hh[i >> 1], so, go);
                }
            }
            so.drain(false, alive);                            // <= 32 new bytes (16 records) per lane
        }
        if (t == 0) break;
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

// Pass 2 with FOUR LANES PER CHUNK (round 4).  The four rANS states of anscdfenc are independent chains that share only the
// order of their 16-bit words (anscdf_.h:106-138: record i steps state 3 - (i & 3), words go out in record order, downward), and
// pass 2 has no model: nothing but the chunk count (65 104 at 100 MB / 1536) kept it at one wave per SIMD, where a wave issues
// every ~1.5 quad-cycles and the chip's other 7/8 lanes idle.  Here lanes 4i .. 4i + 3 take chunk i of the wave, lane s state s:
// a step is four consecutive records (32t + 4u + 3 - s goes to lane s: descending record order = ascending lane order), the
// emit flags of a quad come from one __ballot, a lane's word goes to the shared stream position + 2 x (emits of the lanes
// before it), the overflow rule (before EVERY record: words so far + 18 >= len -> raw; monotone) is evaluated by every lane with
// its own "words so far" and ORed over the quad -- the same decisions as the one-lane walk.  Four times the waves, a quarter
// of the chain each.  A workgroup = 4 waves = one group of 64 chunks (gsum keeps its meaning).
#define ANSQ_ROW        144u                                   // tile row: one block (64 B hi + 64 B lo records) + 16 B (bank spread)
#define ANSQ_TILE       (16u * ANSQ_ROW)
#define ANSQ_WAVE_LDS   (ANSQ_TILE + 16u * TRC_SRING_STRIDE)   // + 16 rings
#define ANSQ_LDS(GPW)   (4u * (GPW) * ANSQ_WAVE_LDS + 64u + 64u)               // + the waves' byte counts + TrcPace's progress counters
// GPW groups of 64 chunks (four waves each) per workgroup: 1, or 4 with TrcPace (round 5) when the launch is one residency round
// of sixteen waves per CU -- the waves of a SIMD then sit in one workgroup and keep each other's pace (trc_dev.h)
template <int GPW>
__global__ __launch_bounds__(256 * GPW) void trc_ansa_codeq_kernel(
    const u8 *__restrict__ recs, u64 n, u32 chunk, u32 nchunks,
    u8 *__restrict__ scratch, u32 stride, u32 *__restrict__ clen, u32 *__restrict__ gsum)
{
    typedef __attribute__((address_space(3))) u32 lds_u32;
    extern __shared__ __attribute__((aligned(16))) u8 smem_wg_[];
    const u32 wv = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const u32 lane = trc_lane(), s = lane & 3u, ci = lane >> 2;
    u8 *const smem = smem_wg_ + wv * ANSQ_WAVE_LDS;
    u32 *const wsum = (u32 *)(smem_wg_ + 4u * GPW * ANSQ_WAVE_LDS);
    TrcPace pace; pace.init(trc_lds_addr(smem_wg_) + 4u * GPW * ANSQ_WAVE_LDS + 64u, threadIdx.x, wv);
    if (GPW > 1) __syncthreads();
    const u32 cw0 = blockIdx.x * (64u * GPW) + wv * 16u;       // this wave's first chunk
    const u32 c = cw0 + ci;
    const bool alive = c < nchunks;
    const u32 lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    const u32 len = alive ? (c == nchunks - 1u ? lastlen : chunk) : 0u;
    const u32 nrec = 2u * (len + (len & 1u));                  // 4 per byte pair: a multiple of 4
    const u32 room = 2u + 4u * 4u;                             // mnflush: ep <= op + sizeof(io_t) + states * 4  ->  raw
    const u8 *rbase = recs + (u64)(alive ? c : 0u) * (8u * (u64)chunk);

    StreamOut<true, false, true, true> so;
    so.rings = smem + ANSQ_TILE;
    so.scratch = scratch; so.stride = stride; so.c0 = cw0; so.wpos = 0; so.nfl = 0;
    u32 st = TRC_ANS_LOW;
    bool ovf = false;
    const u32 sh = lane & ~3u, below = (1u << s) - 1u;
    const u32 tw = trc_lds_addr(smem) + ci * ANSQ_ROW + 16u * s;                      // where this lane's two pieces of a block land
    const u32 tr = trc_lds_addr(smem) + ci * ANSQ_ROW + ((s & 1u) ? 0u : 64u) + (s < 2u ? 4u : 0u);   // its records: plane, then every other dword

    const u32 T = chunk / 16u;                                 // blocks (32 records each) of a full chunk
    const u32 top = nrec ? (nrec - 1u) / 32u : 0u;
    // a lane that has nothing to store this step stores into its own dummy slot (the row's 16 pad bytes): no exec-mask branch
    const u32 ringw = trc_lds_addr(so.rings) + ci * TRC_SRING_STRIDE;
    const u32 dummy = trc_lds_addr(smem) + ci * ANSQ_ROW + 128u + 4u * s;
    uint4 nh = make_uint4(0, 0, 0, 0), nl = nh;
    if (alive && nrec && T - 1u <= top) { nh = trc_ld16_nt(rbase + (T - 1u) * 128u + 16u * s); nl = trc_ld16_nt(rbase + (T - 1u) * 128u + 64u + 16u * s); }
    for (u32 t = T - 1u;; t--) {
        if (GPW > 1 && !(t & 3u)) pace.step(T - t);
        trc_ldsw128(tw, nh); trc_ldsw128(tw + 64u, nl);
        if (t && alive && nrec && t - 1u <= top) { nh = trc_ld16_nt(rbase + (t - 1u) * 128u + 16u * s); nl = trc_ld16_nt(rbase + (t - 1u) * 128u + 64u + 16u * s); }
        const bool act = alive && nrec != 0u && t <= top;
        const u32 left = act ? nrec - 32u * t : 0u;            // records of this chunk from this block's first one on (>= 32: the whole block)
        u32 r[8];
#pragma unroll
        for (int u = 0; u < 8; u++) r[u] = *(const lds_u32 *)(uintptr_t)(tr + 8u * (u32)u);
#pragma unroll
        for (int half = 1; half >= 0; half--) {
            // The reference tests "words so far + 18 >= len -> raw" before EVERY record; the test is monotone in the words, so it
            // fires somewhere iff it fires before the LAST record (record 0: block 0, lane 3's last step).  Tested here exactly there,
            // and -- only to stop a hopeless chunk from running out of its region -- before every 16 records with the quad's common
            // count (also the reference's test, before that half's first record).
            ovf = ovf || (act && 16u * (u32)half < left && so.wpos + room >= len);
#pragma unroll
            for (int uu = 3; uu >= 0; uu--) {
                const int u = 4 * half + uu;
                const bool go = 4u * (u32)u < left && !ovf;    // (records exist in fours: the same for the four lanes)
                const u32 f = r[u] & 0x7fffu, c0 = r[u] >> 15;
                // (the vote is taken on the bare compare and masked with `go` as a value: a vote on `go && compare` makes the compiler
                // rebuild the lane mask through a 0 / 1 vector, two instructions per step)
                const bool ge = st >= (f << 16), emit = go && ge;
                const u32 q4 = (u32)(__ballot(ge) >> sh) & (go ? 15u : 0u);
                const u32 pre = (u32)__builtin_popcount(q4 & below), tot = (u32)__builtin_popcount(q4);
                if (t == 0 && half == 0 && uu == 0)            // before the last record (lane 3's; the others' counts are smaller)
                    ovf = ovf || (((u32)(__ballot(go && so.wpos + 2u * pre + room >= len) >> sh) & 15u) != 0u);
                const u32 at = ringw + 2u * (~((so.wpos >> 1) + pre) & (TRC_SRING / 2u - 1u));      // (ring offset of unit u: -(2u + 2) mod the ring)
                trc_lds_write16(emit ? at : dummy, st);
                so.wpos += 2u * tot;
                const u32 s1 = emit ? st >> 16 : st;
                st = go ? trc_rans_step(s1, f, TRC_PROB_ONE - f, c0) : st;            // (garbage where !go: f may be 0)
            }
            so.drain(false, alive);                            // <= 32 new bytes (16 records) per chunk
        }
        if (t == 0) break;
    }
    u32 out_len = 0;
    if (alive) {
        if (!ovf) {                                            // mnflush: states 0..3, high half first, each below the one before
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
template <bool NIB>
__global__ __launch_bounds__(64 * TRC_WPG) void trc_ansa_dec_kernel(
    const u8 *__restrict__ payload, const u32 *__restrict__ clen, const u64 *__restrict__ goff, const u32 *__restrict__ gsum,
    u64 n, u32 chunk, u32 nchunks, u8 *__restrict__ out, u32 *__restrict__ prog, u32 *__restrict__ prog_host, u32 prog_part)
{
    TRC_QUAD_PROLOGUE(ANSA_MODEL_LDS(NIB));
    NibModel<NIB ? 1 : 17> m; m.init(smem);

    WaveChunks wc;
    wc.c0 = grp_ * 64u; wc.chunk = chunk; wc.nchunks = nchunks;
    wc.lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    wc.rows = nchunks - wc.c0 < 64u ? nchunks - wc.c0 : 64u;
    wc.prog = prog; wc.prog_host = prog_host; wc.prog_part = prog_part;      // (host-pointer decodes: the output leaves while the waves decode, trc_io.h)
    const bool alive = lane < wc.rows;
    const u32 c = wc.c0 + lane;
    const u32 len = alive ? wc.len_of(lane) : 0u;
    const u32 cl = alive ? trc_min(clen[c], len) : 0u;        // a directory entry above the chunk length (corrupt input) reads as raw
    const u32 ex = trc_wave_incl_scan(cl) - cl;
    const u64 off = trc_group_base(goff, gsum, wc.c0 >> 6) + ex;
    const bool coded = alive && cl != len;
    const u32 body = len & ~3u;

    constexpr u32 NST = NIB ? 2u : 4u;
    u32 st[4] = { TRC_ANS_LOW, TRC_ANS_LOW, TRC_ANS_LOW, TRC_ANS_LOW };
    if (coded) for (u32 k = 0; k < NST; k++) st[k] = trc_ld32_a2(payload + off + 4u * k);   // decoder st[i] = encoder st[NST-1-i] (mnfill)
    LaneInWide si; si.prime(payload + off + 4u * NST, coded && NIB, trc_sub_sat(cl, 4u * NST));   // words follow the states
    LaneLook16 sl;                                             // (the byte coder's stream side: trc_lane_io.h)
    if (!NIB) sl.prime(payload + off + 4u * NST, trc_sub_sat(cl, 4u * NST));

    // cdf16ansdec: search + state update + model update; the renorm comes separately (its order is the word order)
    auto get_nibble = [&](u32 &s, u8 *tb, bool act) -> u32 {
        const u32 slot = s & (TRC_PROB_ONE - 1);
        NibTable T = m.load(tb);
        u32 c0, c1;
        const u32 x = trc_nib_find(T, slot, c0, c1);
        s = (NIB ? act : true) ? __umul24(c1 - c0, s >> TRC_PROB_BITS) + slot - c0 : s;      // (byte coder: lanes that are not decoding run along, see trc_rc_adaptive.hip)
        m.adapt(T, x); m.store(tb, T);
        return x;
    };
    // Table 0 (the hi-nibble table of the byte model, the only table of the nibble coder) is used at every step: it lives in
    // registers for the whole chunk and never travels to LDS -- two of the four dependent LDS round trips of a byte (one wave
    // per SIMD: nothing else hides them), as in the range decoders (trc_rc_adaptive.hip)
    NibTable T0 = m.load(m.table(0));
    auto get_hi = [&](u32 &s, bool act) -> u32 {
        const u32 slot = s & (TRC_PROB_ONE - 1);
        u32 c0, c1;
        const u32 x = trc_nib_find(T0, slot, c0, c1);
        s = (NIB ? act : true) ? __umul24(c1 - c0, s >> TRC_PROB_BITS) + slot - c0 : s;
        m.adapt(T0, x);
        return x;
    };
    auto renorm = [&](u32 &s, bool act) {
        const u32 w = si.peek16();
        const bool rn = (NIB ? act : true) && s < TRC_ANS_LOW;
        s = rn ? (s << 16) | w : s;
        si.skip_if(rn);
    };

    if (wc.prog) {                                             // chunks stored raw go first: a part is reported only when ALL its bytes are out
        wc.skip_rows = __ballot(alive && cl == len && len != 0);  // (and the loop's stores of those rows -- zeros -- stay away from them)
        trc_wave_copy_raw(wc.skip_rows, off, len, out + (u64)wc.c0 * chunk, chunk, payload);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");          // (plain stores: written back to memory before this wave reports anything)
    }
    QuadOut qout; qout.base = out + (u64)wc.c0 * chunk;
    u8 *dst = out + (u64)c * chunk;
    const u32 S = chunk / TRC_SEG;
    for (u32 s = 0; s < S; s++) {
        TRC_PACE_STEP(s + 1u);
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
                    if (!NIB) {
#pragma unroll
                        for (int j = 0; j < 2; j++) {          // mndec8x2: two bytes, then four renorms in order st0..st3
                            const bool act = coded && q0 + 2u * (u32)j < len;     // the second byte of an odd tail is the dummy
                            const uint4 W = sl.fetch();        // (<= 8 stream bytes per group: trc_lane_io.h LaneLook16)
                            const u32 h0 = get_hi(st[0], act), l0 = get_nibble(st[1], m.table(1u + h0), act);
                            const u32 h1 = get_hi(st[2], act), l1 = get_nibble(st[3], m.table(1u + h1), act);
                            u32 cnt = 0;
                            sl.renorm<0>(st[0], cnt); sl.renorm<1>(st[1], cnt); sl.renorm<2>(st[2], cnt); sl.renorm<3>(st[3], cnt);
                            sl.end_group(cnt, W);
                            w |= ((h0 << 4 | l0) | (h1 << 4 | l1) << 8) << (16 * j);
                        }
                    } else {
                        const uint4 pre = si.prefetch();
#pragma unroll
                        for (int i = 0; i < 4; i++) {          // mndec4: positions 0,2 of a group <- st[0], 1,3 <- st[1]; tail <- st[1]
                            const u32 pos = q0 + (u32)i;
                            const bool act = coded && pos < len;
                            const bool first = !(i & 1) && pos < body;
                            u32 cur = first ? st[0] : st[1];
                            const u32 x = get_hi(cur, act);
                            renorm(cur, act);
                            st[0] = first ? cur : st[0]; st[1] = first ? st[1] : cur;
                            w |= x << (8 * i);
                        }
                        si.end_step(pre);
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
    if (!wc.prog) trc_wave_copy_raw(__ballot(alive && cl == len && len != 0), off, len, out + (u64)wc.c0 * chunk, chunk, payload);
}

// ---- ANSA's decoder as two waves per 64 chunks (round 4; the scheme of trc_rca_dec_mc_kernel, trc_rc_adaptive.hip) -------
// Wave D owns the four rANS states and the stream and READS tables; wave M owns the tables and adapts them.  Per byte:
//     phase A   D: hi table -> slot search, state update, h -> mailbox      M: the byte before's lo table adapts by its l
//     phase B   D: lo table (h) -> search, state update, l -> mailbox       M: hi table adapts by h; lo table (h) loaded
// one LDS-only barrier behind each phase; the four renormalisations of a byte pair (their order is the word order) stay
// with wave D behind the pair.
#define ANSA_DMC_MBOX   512u
#define ANSA_DMC_LDS    (TRC_NIB_BYTES + ANSA_DMC_MBOX)
__global__ __launch_bounds__(128 * TRC_WPG) void trc_ansa_dec_mc_kernel(
    const u8 *__restrict__ payload, const u32 *__restrict__ clen, const u64 *__restrict__ goff, const u32 *__restrict__ gsum,
    u64 n, u32 chunk, u32 nchunks, u8 *__restrict__ out)
{
    typedef __attribute__((address_space(3))) u32 lds_u32;
    extern __shared__ __attribute__((aligned(16))) u8 smem_wg_[];
    const u32 wv_ = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool model = wv_ >= TRC_WPG;
    const u32 grp_ = blockIdx.x * TRC_WPG + (wv_ & (TRC_WPG - 1u));
    if (grp_ >= (nchunks + 63u) / 64u) return;                 // (both waves of the pair)
    u8 *const smem = smem_wg_ + (wv_ & (TRC_WPG - 1u)) * ANSA_DMC_LDS;
    const u32 lane = trc_lane();
    NibModel<17> m;
    if (model) m.init(smem); else m.attach(smem);
    const u32 mb = trc_lds_addr(smem) + TRC_NIB_BYTES + lane * 4u;     // mailbox: h at +0, l at +256

    WaveChunks wc;
    wc.c0 = grp_ * 64u; wc.chunk = chunk; wc.nchunks = nchunks;
    wc.lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    wc.rows = nchunks - wc.c0 < 64u ? nchunks - wc.c0 : 64u;
    const bool alive = lane < wc.rows;
    const u32 c = wc.c0 + lane;
    const u32 len = alive ? wc.len_of(lane) : 0u;
    const u32 cl = alive ? trc_min(clen[c], len) : 0u;        // a directory entry above the chunk length (corrupt input) reads as raw
    const bool coded = alive && cl != len;
    const u32 S = chunk / TRC_SEG;
    trc_lds_barrier();                                         // the model's initial tables are in place

    if (model) {
        NibTable T0 = m.load(m.table(0)), TL = T0;
        u32 hp = 0;
        bool have = false;
        for (u32 s = 0; s < S; s++) {
#pragma nounroll
            for (u32 k = 0; k < 4; k++) {
                const u32 p0 = s * TRC_SEG + k * 16u;
                if (!__ballot(coded && p0 < len)) continue;
#pragma nounroll
                for (u32 b = 0; b < 16u; b++) {
                    if (have) {                                // phase A: the byte before's lo table
                        const u32 l = *(const lds_u32 *)(uintptr_t)(mb + 256u);
                        m.adapt(TL, l & 15u); m.store(m.table(1u + hp), TL);
                    }
                    have = true;
                    trc_lds_barrier();
                    const u32 h = *(const lds_u32 *)(uintptr_t)mb & 15u;         // phase B
                    m.adapt(T0, h); m.store(m.table(0), T0);
                    TL = m.load(m.table(1u + h)); hp = h;
                    trc_lds_barrier();
                }
            }
        }
        return;
    }

    const u32 ex = trc_wave_incl_scan(cl) - cl;
    const u64 off = trc_group_base(goff, gsum, wc.c0 >> 6) + ex;
    u32 st[4] = { TRC_ANS_LOW, TRC_ANS_LOW, TRC_ANS_LOW, TRC_ANS_LOW };
    if (coded) for (u32 k = 0; k < 4u; k++) st[k] = trc_ld32_a2(payload + off + 4u * k);   // decoder st[i] = encoder st[3-i] (mnfill)
    LaneInWide si; si.prime(payload + off + 16u, coded, trc_sub_sat(cl, 16u));             // words follow the states

    // cdf16ansdec without its model half: search + state update against the table at `tb` as wave M left it
    auto get = [&](u32 &sx, const u8 *tb, bool act) __attribute__((always_inline)) -> u32 {
        const u32 slot = sx & (TRC_PROB_ONE - 1);
        const NibTable T = m.load(tb);
        u32 c0, c1;
        const u32 x = trc_nib_find(T, slot, c0, c1);
        sx = act ? __umul24(c1 - c0, sx >> TRC_PROB_BITS) + slot - c0 : sx;
        return x;
    };
    auto get_byte = [&](u32 &sh, u32 &sl, bool act) __attribute__((always_inline)) -> u32 {
        const u32 h = get(sh, m.table(0), act);
        *(lds_u32 *)(uintptr_t)mb = h;
        trc_lds_barrier();
        const u32 l = get(sl, m.table(1u + h), act);
        *(lds_u32 *)(uintptr_t)(mb + 256u) = l;
        trc_lds_barrier();
        return h << 4 | l;
    };
    auto renorm = [&](u32 &sx, bool act) __attribute__((always_inline)) {
        const u32 w = si.peek16();
        const bool rn = act && sx < TRC_ANS_LOW;
        sx = rn ? (sx << 16) | w : sx;
        si.skip_if(rn);
    };

    QuadOut qout; qout.base = out + (u64)wc.c0 * chunk;
    u8 *dst = out + (u64)c * chunk;
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
                    for (u32 j = 0; j < 2; j++) {              // mndec8x2: two bytes, then four renorms in order st0..st3
                        const bool act = coded && q0 + 2u * j < len;          // the second byte of an odd tail is the dummy
                        const uint4 pre = si.prefetch();       // (<= 8 stream bytes per group: trc_lane_io.h LaneInWide)
                        const u32 x0 = get_byte(st[0], st[1], act);
                        const u32 x1 = get_byte(st[2], st[3], act);
                        renorm(st[0], act); renorm(st[1], act); renorm(st[2], act); renorm(st[3], act);
                        si.end_step(pre);
                        w |= (x0 | x1 << 8) << (16u * j);
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

// ------------------------------------------------------------------------------------- launch ---
// TRC_ANSA_MC=0 selects the one-wave model pass of rounds 1-3 (A/B measurements, tests of both forms)
static bool ansa_mc_enabled()
{
    static const int env = getenv("TRC_ANSA_MC") ? atoi(getenv("TRC_ANSA_MC")) : 1;
    return env != 0;
}
bool trc_ansa_enc_gate_ok() { return ansa_mc_enabled(); }
template <bool NIB>
static void launch_ansa_enc(const uint8_t *d_in, size_t n, uint32_t chunk, const TrcWork &w, uint32_t *d_clen, hipStream_t s)
{
    if (!NIB && ansa_mc_enabled()) {
        TRC_RAISE_LDS_ONCE(trc_ansa_model2_kernel, TRC_WPG * ANSA_MODEL_LDS(false) + 64u);
        TRC_RAISE_LDS_ONCE(trc_ansa_code_planar_kernel, TRC_WPG * ANSA_CODE_PLANAR_LDS);
        TRC_LAUNCH_TIMED(trc_ansa_model2_kernel, TRC_QUAD_GRID(w.ngroups), dim3(128 * TRC_WPG), TRC_WPG * ANSA_MODEL_LDS(false) + 64u, s, d_in, (u64)n, chunk, w.nchunks, w.scratch2, trc_gate_tls.flag, trc_gate_tls.part);
        trc_launch_ansa_code_planar(n, chunk, w, d_clen, s);
        return;
    }
    if (NIB && trc_nib_big(w.ngroups)) {                       // one 12-wave workgroup per CU, pace-keeping (trc_dev.h)
        TRC_RAISE_LDS_ONCE((trc_ansa_model_kernel<NIB>), TRC_LDS_ONE_PER_CU);
        TRC_LAUNCH_TIMED((trc_ansa_model_kernel<NIB>), dim3((w.ngroups + TRC_NIB_WPG - 1u) / TRC_NIB_WPG), dim3(64 * TRC_NIB_WPG), TRC_LDS_ONE_PER_CU, s, d_in, (u64)n, chunk, w.nchunks, w.scratch2);
    } else {
    TRC_RAISE_LDS_ONCE((trc_ansa_model_kernel<NIB>), NIB ? TRC_LDS_ONE_PER_CU : TRC_WPG * ANSA_MODEL_LDS(NIB));   // (one limit for both shapes of the nibble form: the attribute is set once per call site)
    TRC_LAUNCH_TIMED((trc_ansa_model_kernel<NIB>), TRC_QUAD_GRID(w.ngroups), dim3(64 * TRC_WPG), TRC_WPG * (ANSA_MODEL_LDS(NIB)), s, d_in, (u64)n, chunk, w.nchunks, w.scratch2);
    }
    TRC_LAUNCH_TIMED((trc_ansa_code_kernel<NIB>), TRC_QUAD_GRID(w.ngroups), dim3(64 * TRC_WPG), TRC_WPG * (ANSA_CODE_LDS), s,
                       (const u8 *)w.scratch2, (u64)n, chunk, w.nchunks, w.scratch, w.stride, d_clen, w.gsum);
}
// TRC_ANSA_DMC=1 selects the two-wave decoder.  MEASURED AND NOT USED (profiles/r04_notes.md): bit-exact, 0.93 ms against 0.68.
static bool ansa_dmc_enabled()
{
    static const int env = getenv("TRC_ANSA_DMC") ? atoi(getenv("TRC_ANSA_DMC")) : 0;
    return env != 0;
}
bool trc_ansa_dec_prog_ok() { return !ansa_dmc_enabled(); }       // the one-wave decoder (the default) reports its progress
template <bool NIB>
static void launch_ansa_dec(const uint8_t *d_payload, const uint32_t *d_clen, size_t n, uint32_t chunk,
                            const TrcWork &w, uint8_t *d_out, hipStream_t s)
{
    if (!NIB && ansa_dmc_enabled()) {
        TRC_RAISE_LDS_ONCE(trc_ansa_dec_mc_kernel, TRC_WPG * ANSA_DMC_LDS);
        TRC_LAUNCH_TIMED(trc_ansa_dec_mc_kernel, TRC_QUAD_GRID(w.ngroups), dim3(128 * TRC_WPG), TRC_WPG * ANSA_DMC_LDS, s,
                           d_payload, d_clen, w.goff, w.gsum, (u64)n, chunk, w.nchunks, d_out);
        return;
    }
    // (the nibble decoder in the 12-wave shape with TrcPace -- what the nibble range coders gained 8-10 % from -- measured 0.244 -> 0.322 ms:
    // not taken, profiles/r05q_ab.txt)
    TRC_RAISE_LDS_ONCE((trc_ansa_dec_kernel<NIB>), TRC_WPG * ANSA_MODEL_LDS(NIB));
    TRC_LAUNCH_TIMED((trc_ansa_dec_kernel<NIB>), TRC_QUAD_GRID(w.ngroups), dim3(64 * TRC_WPG), TRC_WPG * (ANSA_MODEL_LDS(NIB)), s,
                       d_payload, d_clen, w.goff, w.gsum, (u64)n, chunk, w.nchunks, d_out, NIB ? nullptr : trc_prog_tls.counters, NIB ? nullptr : trc_prog_tls.host_flags, NIB ? 0u : trc_prog_tls.part);
}
// pass 2 over the planar record space (ANSA's two-wave model pass, and the order-1 coder's)
void trc_launch_ansa_code_planar(size_t n, uint32_t chunk, const TrcWork &w, uint32_t *d_clen, hipStream_t s)
{
    static const int codeq = getenv("TRC_ANSA_CODEQ") ? atoi(getenv("TRC_ANSA_CODEQ")) : 1;     // 0: one lane per chunk (rounds 1-3)
    static const int gpw_env = getenv("TRC_CODEQ_GPW") ? atoi(getenv("TRC_CODEQ_GPW")) : 0;     // tuning aid: 1 / 4 force the workgroup shape
    const bool big = gpw_env ? gpw_env == 4 : (w.ngroups >= 512u && w.ngroups <= 4u * 256u);
    if (codeq && big) {
        // (the LDS request is padded beyond half a CU's: one 16-wave workgroup per CU, four waves per SIMD.  At its real 71 KiB two fit,
        // and where the dispatcher doubles up a CU runs eight waves per SIMD while another idles -- the pass was bimodal, 0.20 / 0.30 ms)
        const size_t lds1 = ANSQ_LDS(4) > TRC_LDS_ONE_PER_CU ? ANSQ_LDS(4) : TRC_LDS_ONE_PER_CU;
        TRC_RAISE_LDS_ONCE(trc_ansa_codeq_kernel<4>, lds1);
        TRC_LAUNCH_TIMED(trc_ansa_codeq_kernel<4>, dim3((w.ngroups + 3u) / 4u), dim3(1024), lds1, s,
                           (const u8 *)w.scratch2, (u64)n, chunk, w.nchunks, w.scratch, w.stride, d_clen, w.gsum);
    } else if (codeq)
        TRC_LAUNCH_TIMED(trc_ansa_codeq_kernel<1>, dim3(w.ngroups), dim3(256), ANSQ_LDS(1), s,
                           (const u8 *)w.scratch2, (u64)n, chunk, w.nchunks, w.scratch, w.stride, d_clen, w.gsum);
    else {
        TRC_RAISE_LDS_ONCE(trc_ansa_code_planar_kernel, TRC_WPG * ANSA_CODE_PLANAR_LDS);
        TRC_LAUNCH_TIMED(trc_ansa_code_planar_kernel, TRC_QUAD_GRID(w.ngroups), dim3(64 * TRC_WPG), TRC_WPG * (ANSA_CODE_PLANAR_LDS), s,
                           (const u8 *)w.scratch2, (u64)n, chunk, w.nchunks, w.scratch, w.stride, d_clen, w.gsum);
    }
}
// pass 2 alone (the order-1 coder of trc_ans_o1.hip produces the same record stack with its own pass 1)
void trc_launch_ansa_code(int nibble, size_t n, uint32_t chunk, const TrcWork &w, uint32_t *d_clen, hipStream_t s)
{
    if (nibble)
        TRC_LAUNCH_TIMED((trc_ansa_code_kernel<true>), TRC_QUAD_GRID(w.ngroups), dim3(64 * TRC_WPG), TRC_WPG * (ANSA_CODE_LDS), s,
                           (const u8 *)w.scratch2, (u64)n, chunk, w.nchunks, w.scratch, w.stride, d_clen, w.gsum);
    else
        TRC_LAUNCH_TIMED((trc_ansa_code_kernel<false>), TRC_QUAD_GRID(w.ngroups), dim3(64 * TRC_WPG), TRC_WPG * (ANSA_CODE_LDS), s,
                           (const u8 *)w.scratch2, (u64)n, chunk, w.nchunks, w.scratch, w.stride, d_clen, w.gsum);
}
void trc_launch_ansa_enc(int nibble, const uint8_t *d_in, size_t n, uint32_t chunk, const TrcWork &w, uint32_t *d_clen, hipStream_t s)
{
    if (nibble) launch_ansa_enc<true>(d_in, n, chunk, w, d_clen, s); else launch_ansa_enc<false>(d_in, n, chunk, w, d_clen, s);
}
void trc_launch_ansa_dec(int nibble, const uint8_t *d_payload, const uint32_t *d_clen, size_t n, uint32_t chunk,
                         const TrcWork &w, uint8_t *d_out, hipStream_t s)
{
    if (nibble) launch_ansa_dec<true>(d_payload, d_clen, n, chunk, w, d_out, s); else launch_ansa_dec<false>(d_payload, d_clen, n, chunk, w, d_out, s);
}
