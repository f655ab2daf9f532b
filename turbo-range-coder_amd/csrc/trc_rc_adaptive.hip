// trc_rc_adaptive.hip -- the adaptive-CDF range coders (CDF16 model, trc_nibmodel.h):
//   TRC_RCA    rccdfenc / rccdfdec      rccdf.c:187-211   `turborc -e46`     bytes, one stream
//   TRC_RCAI   rccdfienc / rccdfidec    rccdf.c:213-249   `turborc -e47`     bytes, hi nibbles on stream 0, lo nibbles on stream 1
//   TRC_RCA4   rccdf4enc / rccdf4dec    rccdf.c:250-275   `turborc -n -e46`  nibbles (values 0..15), one table, one stream
//   TRC_RCAI4  rccdf4ienc / rccdf4idec  rccdf.c:277-323   `turborc -n -e47`  nibbles, even positions on stream 0, odd on stream 1
// Per chunk the payload is exactly what the reference function returns for that slice.  Byte coders (cdf8e/cdf4e
// rccdf_.h:28-40; decoders cdf8d/cdf4d :48-76 with the 16-way search cdflget16 turborc_.h:259-304): the hi nibble
// is coded with the "hi" table, which then adapts; the lo nibble with the "lo" table selected by the hi nibble,
// which then adapts.  Two-stream payloads are [u32 len0][stream 0][stream 1].  Incompressibility rules:
//   RCA / RCA4  OVERFLOW (rcutil_.h:130) after every symbol group (monotone: evaluated once per period);
//   RCAI        OVERFLOWI (rccdf.c:46) after every FULL group of 4 bytes, OVERFLOW on the total at the end;
//   RCAI4       OVERFLOW on stream 1 after every pair, on the total at the end.  The reference returns a meaningless
//               length from the in-loop test (it hands op1 to the macro and returns op0-out, rccdf.c:314,322) and never
//               tests stream 0 against stream 1's base out+4+n/2; in both cases this coder stores the chunk raw.
//               BOTH symbols of a pair are coded against the table as it was before the pair (rccdf.c:311-314).
// Range coder core, carry scheme, exact code/range quotient: trc_rc.h.
//
// One lane = one chunk.  These coders are bound by the per-nibble table update, and their occupancy by the model
// (544 B per lane in LDS): the kernels are written so that the model is the ONLY thing in LDS -- input/output bytes
// move in quad-transposed 64-byte segments in registers (QuadIn/QuadOut, trc_io.h), coded streams through 16-byte
// register windows (trc_lane_io.h) -- which lets four waves (one per SIMD) share a CU.  A period is 4 input bytes:
// first the model walks them and leaves {cdf_lo, freq} records in registers (pure packed-16 VALU + LDS, no
// dependence on the coder state), then the range coder consumes the records with predicated, branch-free steps.
#include <stdlib.h>
#include <type_traits>
#include "trc_rc.h"
#include "trc_lane_io.h"
#include "trc_nibmodel.h"
#include "trc_launch.h"

#define RCA_WAVE_LDS(NIB) ((NIB) ? TRC_NIB1_BYTES : TRC_NIB_BYTES)

template <int NS, bool NIB>
__global__ __launch_bounds__(64 * (NIB ? TRC_NIB_WPG : TRC_WPG)) void trc_rca_enc_kernel(
    const u8 *__restrict__ in, u64 n, u32 chunk, u32 nchunks,
    u8 *__restrict__ scratch, u32 stride, u8 *__restrict__ scratch2, u32 stride2,
    u32 *__restrict__ clen, u32 *__restrict__ gsum)
{
    TRC_QUAD_PROLOGUE(RCA_WAVE_LDS(NIB));
    NibModel<NIB ? 1 : 17> m; m.init(smem);

    WaveChunks wc;
    wc.c0 = grp_ * 64u; wc.chunk = chunk; wc.nchunks = nchunks;
    wc.lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    wc.rows = nchunks - wc.c0 < 64u ? nchunks - wc.c0 : 64u;
    const bool alive = lane < wc.rows;
    const u32 c = wc.c0 + lane;
    const u32 len = alive ? wc.len_of(lane) : 0u;
    const int lim = trc_rc_limit(len);
    const u32 off1 = 4u + len / 2u;                            // stream-1 base inside `out` (rccdf.c:215,279)

    QuadIn qin; qin.base = in + (u64)wc.c0 * chunk;
    LaneOutDirect o0, o1;
    o0.start(scratch + (u64)c * stride + (NS == 2 ? 4u : 0u));
    o1.start(NS == 2 ? scratch2 + (u64)c * stride2 : scratch);
    RcEncD e0, e1; e0.start(); e1.start();
    bool ovf = alive && NS == 1 && lim <= 0;
    NibTable T0 = m.load(m.table(0));                          // hi table / the nibble coders' table: registers (record_bytes, trc_nibmodel.h)

    const u32 S = chunk / TRC_SEG;
    qin.issue(wc, 0);
    for (u32 s = 0; s < S; s++) {
        TRC_PACE_STEP(s + 1u);
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
                // ---- model: 4 bytes -> records (no coder state involved)
                u32 rc[8];
                if (!NIB) {
                    const u32 x[4] = { w & 255u, (w >> 8) & 255u, (w >> 16) & 255u, w >> 24 };
                    m.template record_bytes<4>(T0, x, rc);
                } else if (NS == 1) {
                    const u32 x[4] = { w & 15u, (w >> 8) & 15u, (w >> 16) & 15u, (w >> 24) & 15u };
                    u32 r4[4];
                    m.template record_nibs<4>(T0, x, r4);
#pragma unroll
                    for (int i = 0; i < 4; i++) rc[i] = r4[i];
                } else {
#pragma unroll
                    for (int pr = 0; pr < 2; pr++) {           // both symbols against the table before the pair
                        const u32 x0 = (w >> (16 * pr)) & 15u, x1 = (w >> (16 * pr + 8)) & 15u;
                        u8 *tb = m.table(0);
                        u32 a0, a1, b0, b1;
                        m.bounds(tb, x0, a0, a1); m.bounds(tb, x1, b0, b1);
                        NibTable T = m.load(tb); m.adapt(T, x0); m.adapt(T, x1); m.store(tb, T);
                        rc[2 * pr] = (a0 << TRC_PROB_BITS) | (a1 - a0);
                        rc[2 * pr + 1] = (b0 << TRC_PROB_BITS) | (b1 - b0);
                    }
                }
                // ---- range coder: predicated steps; an encoder's remembered word goes out after every second symbol
                if (!NIB) {
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const bool act = run && q0 + (u32)i < len;
                        e0.sym_rec(act, rc[2 * i] >> TRC_PROB_BITS, rc[2 * i] & 0x7fffu);
                        if (NS == 1) { e0.sym_rec(act, rc[2 * i + 1] >> TRC_PROB_BITS, rc[2 * i + 1] & 0x7fffu); e0.flush(o0); }
                        else {
                            e1.sym_rec(act, rc[2 * i + 1] >> TRC_PROB_BITS, rc[2 * i + 1] & 0x7fffu);
                            if (i & 1) { e0.flush(o0); e1.flush(o1); }
                        }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const bool act = run && q0 + (u32)i < len;
                        if (NS == 1 || !(i & 1)) e0.sym_rec(act, rc[i] >> TRC_PROB_BITS, rc[i] & 0x7fffu);
                        else                     e1.sym_rec(act, rc[i] >> TRC_PROB_BITS, rc[i] & 0x7fffu);
                        if (NS == 1 && (i & 1)) e0.flush(o0);
                    }
                    if (NS == 2) { e0.flush(o0); e1.flush(o1); }
                }
                // ---- incompressibility tests (all monotone in the word counts)
                if (NS == 1) ovf = ovf || (run && q0 < len && (int)(4u * e0.cw.nwords) >= lim);
                else if (!NIB) ovf = ovf || (run && q0 + 4u <= len &&
                                             ((int)(off1 + 4u * e1.cw.nwords) >= lim || 4u + 4u * e0.cw.nwords >= off1));
                else ovf = ovf || (run && q0 + 2u <= len && (int)(off1 + 4u * e1.cw.nwords) >= lim);
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
                if ((int)out_len >= lim || (NIB && 4u + o0.wpos > off1)) ovf = true;
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

// ---- the byte coders' encoder as TWO WAVES per 64 chunks (round 4) -------------------------------------------------
// The encoder above already runs in two phases per period of 4 bytes: the model walks the bytes and leaves records, the
// range coder consumes them.  Nothing of the first phase depends on the coder's state.  With the model filling the LDS
// a CU holds four such waves, one per SIMD, and a lone wave issues an instruction every ~1.5 quad-cycles
// (profiles/r02_notes.md): 40 % of the VALU slots.  Here the two phases are two WAVES of one workgroup: wave 0 owns the
// model (LDS rows, K table, the input bytes) and pushes the 8 records of a period into a double-buffered LDS queue
// (2 x 32 B per lane), wave 1 owns the range coder (state, carry logic, output) and pops them one period later; one
// s_barrier per period keeps them a period apart.  Same LDS per 64 chunks + 4 KiB, twice the waves per SIMD: each wave's
// issue gaps are the other's slots.  Payloads are bit-identical by construction (same records, same coder).
#define RCA_MC_QUEUE   (2u * 2u * 64u * 16u)                   // [buffer][half][lane][16 B]
#define RCA_MC_LDS     (TRC_NIB_BYTES + RCA_MC_QUEUE)

template <int NS>
__global__ __launch_bounds__(192 * TRC_WPG) void trc_rca_enc_mc_kernel(
    const u8 *__restrict__ in, u64 n, u32 chunk, u32 nchunks,
    u8 *__restrict__ scratch, u32 stride, u8 *__restrict__ scratch2, u32 stride2,
    u32 *__restrict__ clen, u32 *__restrict__ gsum, const u32 *__restrict__ gate, u32 gate_part)
{
    // a workgroup = 4 hi-model waves (0-3) + 4 lo-model waves (4-7) + 4 coder waves (8-11): set k = waves k, k + 4, k + 8 = group
    // 4 blockIdx + k; the dispatcher deals a workgroup's waves over consecutive SIMDs, so SIMD k of the CU holds set k (trc_dev.h,
    // TRC_WPG).  THREE waves since the ablations (profiles/r04_notes.md): of the two-wave form the model wave alone took 0.56 ms,
    // the coder wave alone 0.41, both 0.60 -- the model wave was the kernel.  Its two halves need nothing from each other (the hi
    // record of a byte depends on the hi table alone, the lo record on the lo tables alone; the lo table is selected by the hi
    // nibble, which is input: NibModel::record_hi / record_lo), so they are two waves; the queue holds 4 hi records, then 4 lo.
    extern __shared__ __attribute__((aligned(16))) u8 smem_wg_[];
    const u32 wv_ = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const u32 role = wv_ / TRC_WPG;                            // 0: hi records, 1: lo records, 2: coder
    const bool coder = role == 2u;
    const u32 grp_ = blockIdx.x * TRC_WPG + (wv_ & (TRC_WPG - 1u));
    if (grp_ >= (nchunks + 63u) / 64u) return;                 // (all waves of the set: a finished wave no longer counts at s_barrier)
    u8 *const smem = smem_wg_ + (wv_ & (TRC_WPG - 1u)) * RCA_MC_LDS;
    const u32 lane = trc_lane();
    NibModel<17> m;
    if (role == 0u) m.init_part(smem, 0u, 1u); else if (role == 1u) m.init_part(smem, 1u, 16u);     // each model wave its own tables (and the same K)
    const u32 qa = trc_lds_addr(smem) + TRC_NIB_BYTES + lane * 16u;

    WaveChunks wc;
    wc.c0 = grp_ * 64u; wc.chunk = chunk; wc.nchunks = nchunks;
    wc.lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    wc.rows = nchunks - wc.c0 < 64u ? nchunks - wc.c0 : 64u;
    wc.gate = gate; wc.gate_part = gate_part;                  // (host-pointer encodes: the input arrives while the waves code, trc_io.h)
    const u32 S = chunk / TRC_SEG;

    if (!coder) {
        // ---- the model waves.  Walk every period of the chunk (a lane past the end of a short last chunk, or one whose
        // chunk the coder has given up on, leaves records nobody codes).
        QuadIn qin; qin.base = in + (u64)wc.c0 * chunk;
        NibTable T0 = m.load(m.table(0));
        u32 buf = 0;
        qin.issue(wc, 0);
        for (u32 s = 0; s < S; s++) {
            qin.commit();
            if (s + 1 < S) qin.issue(wc, (s + 1) * TRC_SEG);
            uint4 pc0 = qin.read(0), pc1 = qin.read(1), pc2 = qin.read(2), pc3 = qin.read(3);
#pragma nounroll
            for (u32 k = 0; k < 4; k++) {
                uint4 v = pc0; pc0 = pc1; pc1 = pc2; pc2 = pc3;
#pragma nounroll
                for (u32 d = 0; d < 4; d++) {
                    const u32 w = v.x; v.x = v.y; v.y = v.z; v.z = v.w;
                    const u32 x[4] = { w & 255u, (w >> 8) & 255u, (w >> 16) & 255u, w >> 24 };
                    u32 rc[4];
                    // the lo records one byte at a time: with two more waves on the SIMD nothing waits for the LDS round trips a batch of four
                    // requests up front, and the batch's fix-ups between bytes of equal hi nibble (8 selects per earlier byte) are gone (0.612 -> 0.583 ms)
                    if (role == 0u) m.template record_hi<4>(T0, x, rc);
                    else {
#pragma unroll
                        for (int i = 0; i < 4; i++) { const u32 x1[1] = { x[i] }; u32 r1[1]; m.template record_lo<1>(x1, r1); rc[i] = r1[0]; }
                    }
                    trc_ldsw128(qa + buf * 2048u + role * 1024u, make_uint4(rc[0], rc[1], rc[2], rc[3]));
                    trc_lds_barrier();
                    buf ^= 1u;
                }
            }
        }
        return;
    }

    // ---- the coder wave: the range coder, one period behind
    const bool alive = lane < wc.rows;
    const u32 c = wc.c0 + lane;
    const u32 len = alive ? wc.len_of(lane) : 0u;
    const int lim = trc_rc_limit(len);
    const u32 off1 = 4u + len / 2u;                            // stream-1 base inside `out` (rccdf.c:215)
    LaneOutDirect o0, o1;
    o0.start(scratch + (u64)c * stride + (NS == 2 ? 4u : 0u));
    o1.start(NS == 2 ? scratch2 + (u64)c * stride2 : scratch);
    RcEncV e0, e1; e0.start(); e1.start();
    bool ovf = alive && NS == 1 && lim <= 0;
    // A lane codes without a predicate (RcEncV::sym<false>).  The one lane of the grid whose chunk is short ends before its wave
    // does: in the period in which some lane's chunk ends (wave-uniform; also the last period of every full chunk) the symbols are
    // predicated, the ending lanes' coder states are set aside (`done`, snapshots) and given back before finish(); what such a lane
    // computes afterwards is observed by nobody (flush() is gated by `run`).
    bool done = false;
    u32 s0a = 0, s0b = 0, s0c = 0, s0d = 0, s0e = 0, s1a = 0, s1b = 0, s1c = 0, s1d = 0, s1e = 0;

    auto code_period = [&](u32 q0, u32 buf) __attribute__((always_inline)) {
        if (!__ballot(alive && !ovf && !done && q0 < len)) return;
        const u32 a = qa + buf * 2048u;
        const uint4 ra = trc_ldsr128(a), rb = trc_ldsr128(a + 1024u);          // four hi records, four lo records
        const u32 rc[8] = { ra.x, rb.x, ra.y, rb.y, ra.z, rb.z, ra.w, rb.w };
        const bool run = alive && !ovf && !done;
        auto body = [&](auto pred) __attribute__((always_inline)) {
            constexpr bool PRED = decltype(pred)::value;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const bool act = run && q0 + (u32)i < len;
                e0.template sym<PRED>(act, rc[2 * i] >> TRC_PROB_BITS, rc[2 * i] & 0x7fffu);
                if (NS == 1) { e0.template sym<PRED>(act, rc[2 * i + 1] >> TRC_PROB_BITS, rc[2 * i + 1] & 0x7fffu); e0.flush(o0, run); }
                else {
                    e1.template sym<PRED>(act, rc[2 * i + 1] >> TRC_PROB_BITS, rc[2 * i + 1] & 0x7fffu);
                    if (i & 1) { e0.flush(o0, run); e1.flush(o1, run); }
                }
            }
        };
        if (__ballot(run && q0 < len && q0 + 4u >= len)) {     // some lane's chunk ends in this period
            body(std::integral_constant<bool, true>());
            const bool now = run && q0 + 4u >= len;
            s0a = now ? e0.rlo : s0a; s0b = now ? e0.rhi : s0b; s0c = now ? e0.llo : s0c; s0d = now ? e0.lhi : s0d; s0e = now ? e0.lx : s0e;
            s1a = now ? e1.rlo : s1a; s1b = now ? e1.rhi : s1b; s1c = now ? e1.llo : s1c; s1d = now ? e1.lhi : s1d; s1e = now ? e1.lx : s1e;
            done = done || now;
        } else body(std::integral_constant<bool, false>());
        if (NS == 1) ovf = ovf || (run && q0 < len && (int)(4u * e0.cw.nwords) >= lim);
        else ovf = ovf || (run && q0 + 4u <= len &&
                           ((int)(off1 + 4u * e1.cw.nwords) >= lim || 4u + 4u * e0.cw.nwords >= off1));
    };
    {
        const u32 P = S * 16u;                                 // periods of a full chunk = barriers of the model wave
        u32 buf = 0;
#pragma nounroll
        for (u32 p = 0; p <= P; p++) {                         // (one call site: the coder's body exists once)
            if (p) code_period((p - 1u) * 4u, buf ^ 1u);
            if (p < P) trc_lds_barrier();
            buf ^= 1u;
        }
    }
    if (done) {                                                // the states as they were when the chunk ended
        e0.rlo = s0a; e0.rhi = s0b; e0.llo = s0c; e0.lhi = s0d; e0.lx = s0e;
        e1.rlo = s1a; e1.rhi = s1b; e1.llo = s1c; e1.lhi = s1d; e1.lx = s1e;
    }
    u32 out_len = 0;
    if (alive) {
        if (!ovf) {
            e0.finish(o0);
            if (NS == 2) {
                e1.finish(o1);
                out_len = 4u + o0.wpos + o1.wpos;
                if ((int)out_len >= lim) ovf = true;
            } else out_len = o0.wpos;
        }
        if (ovf) out_len = len;
    }
    if (NS == 2 && alive && !ovf) *(u32 *)(scratch + (u64)c * stride) = o0.wpos;          // header: len0
    if (alive) clen[c] = out_len;
    const u32 gs = trc_wave_sum(out_len);
    if (lane == 0) gsum[wc.c0 >> 6] = gs;
}

template <int NS, bool NIB>
__global__ __launch_bounds__(64 * (NIB ? TRC_NIB_WPG : TRC_WPG)) void trc_rca_dec_kernel(
    const u8 *__restrict__ payload, const u32 *__restrict__ clen, const u64 *__restrict__ goff, const u32 *__restrict__ gsum,
    u64 n, u32 chunk, u32 nchunks, u8 *__restrict__ out, u32 *__restrict__ prog, u32 *__restrict__ prog_host, u32 prog_part)
{
    TRC_QUAD_PROLOGUE(RCA_WAVE_LDS(NIB));
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

    LaneIn<4> s0, s1;
    const u32 len0 = (NS == 2 && coded) ? trc_min(trc_ld32_a2(payload + off), trc_sub_sat(cl, 4u)) : 0u;   // a corrupt header cannot point outside the chunk's payload
    s0.prime(payload + off + (NS == 2 ? 4u : 0u), coded && NIB, NS == 2 ? len0 : cl);
    s1.prime(payload + off + 4u + len0, NS == 2 && coded && NIB, trc_sub_sat(cl, 4u + len0));
    RcDec d0, d1;
    constexpr bool LOOK = !NIB;                                // the byte coders' stream side: trc_lane_io.h LaneLook32
    LaneLook32 sl, sl1;
    if (LOOK) {
        u32 a, b;
        sl.prime(payload + off + (NS == 2 ? 4u : 0u), NS == 2 ? len0 : cl, a, b); d0.start(a, b);
        if (NS == 2) { sl1.prime(payload + off + 4u + len0, trc_sub_sat(cl, 4u + len0), a, b); d1.start(a, b); }
    } else {
        { const u32 a = s0.peek32(); s0.skip_if(coded); const u32 b = s0.peek32(); s0.skip_if(coded); d0.start(a, b); }
        { const u32 a = s1.peek32(); s1.skip_if(NS == 2 && coded); const u32 b = s1.peek32(); s1.skip_if(NS == 2 && coded); d1.start(a, b); }
    }

    // Table 0 (the hi-nibble table of the byte model, the only table of the nibble coders) is used at every step: it
    // lives in registers for the whole chunk and never travels to LDS, which takes two of the four dependent LDS round
    // trips out of every byte (one wave per SIMD: nothing else hides them).
    NibTable T0 = m.load(m.table(0));
    // a symbol of a pair: `w` is the pair's look-ahead word, the return value's bit 4 says "renormalised"
    // (byte coders: no predication on `act`.  A lane that is not decoding -- raw chunk, dead lane, past the end of a short last
    // chunk -- runs along on its own registers, its own model row and whatever its clamped stream window holds; nothing of it is
    // observable: its bytes are overwritten by the raw copy or never stored.  Four selects and a mask operation per symbol less.)
    auto get0 = [&](RcDec &dq, u32 w, bool act) -> u32 {
        u32 c0, c1;
        const u32 x = trc_nib_search(T0, dq.scaled(), c0, c1);
        // (the renormalisation as a carry-chain mask + bit-selects, hand-written, measured the same as this compare-and-select form: 0.6035 / 0.6025 ms)
        const u32 rf = dq.consume_w(NIB ? act : true, c0, c1, w) ? 16u : 0u;
        m.adapt(T0, x);
        return x | rf;
    };
    auto get = [&](RcDec &dq, u32 w, u8 *tb, bool act) -> u32 {
        NibTable T = m.load(tb);
        u32 c0, c1;
        const u32 x = trc_nib_search(T, dq.scaled(), c0, c1);  // == first i with t[i+1]*r > code, else 15 (cdflget16)
        // (the renormalisation as a carry-chain mask + bit-selects, hand-written, measured the same as this compare-and-select form: 0.6035 / 0.6025 ms)
        const u32 rf = dq.consume_w(NIB ? act : true, c0, c1, w) ? 16u : 0u;
        m.adapt(T, x); m.store(tb, T);
        return x | rf;
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
                    // every stream advances once per PAIR of its symbols (at most one of the two renormalises)
                    // (the window behind the current one is requested by every lane at the start of a pair and taken in at its
                    // end: trc_lane_io.h `prefetch`)
                    if (!NIB && NS == 1) {
#pragma unroll
                        for (int pr = 0; pr < 2; pr++) {
                            // round 4: the stream side once per TWO bytes (<= 8 stream bytes: one word per byte at most, the window
                            // protocol's limit): one prefetch, the words at the position and behind it, one advance -- the second
                            // byte looks ahead at the second word if the first byte took the first
                            const bool acta = coded && q0 + 2u * (u32)pr < len, actb = coded && q0 + 2u * (u32)pr + 1u < len;
                            const uint4 W = sl.fetch();
                            const u32 swa = sl.w0, swb = sl.w1;
                            const u32 ha = get0(d0, swa, acta);
                            const u32 la = get(d0, swa, m.table(1u + (ha & 15u)), acta);
                            const u32 ra = (ha | la) & 16u;
                            const u32 sw2 = ra ? swb : swa;
                            const u32 hb = get0(d0, sw2, actb);
                            const u32 lb = get(d0, sw2, m.table(1u + (hb & 15u)), actb);
                            sl.end_group((ra + ((hb | lb) & 16u)) >> 4, W);
                            w |= (((ha & 15u) << 4 | (la & 15u)) | ((hb & 15u) << 4 | (lb & 15u)) << 8) << (16 * pr);
                        }
                    } else if (!NIB) {
#pragma unroll
                        for (int pr = 0; pr < 2; pr++) {
                            const bool acta = coded && q0 + 2u * (u32)pr < len, actb = coded && q0 + 2u * (u32)pr + 1u < len;
                            const uint4 W0 = sl.fetch(), W1 = sl1.fetch();
                            const u32 w0 = sl.w0, w1 = sl1.w0;
                            const u32 ha = get0(d0, w0, acta);
                            const u32 la = get(d1, w1, m.table(1u + (ha & 15u)), acta);
                            const u32 hb = get0(d0, w0, actb);
                            const u32 lb = get(d1, w1, m.table(1u + (hb & 15u)), actb);
                            sl.end_group1(((ha | hb) & 16u) >> 4, W0); sl1.end_group1(((la | lb) & 16u) >> 4, W1);
                            w |= (((ha & 15u) << 4 | (la & 15u)) | ((hb & 15u) << 4 | (lb & 15u)) << 8) << (16 * pr);
                        }
                    } else if (NS == 1) {
#pragma unroll
                        for (int pr = 0; pr < 2; pr++) {
                            const u32 sw = s0.peek32();         // (a nibble pair is cheap and consumes little: the plain refill measures better here)
                            const u32 a = get0(d0, sw, coded && q0 + 2u * (u32)pr < len);
                            const u32 b = get0(d0, sw, coded && q0 + 2u * (u32)pr + 1u < len);
                            s0.skip_if(((a | b) & 16u) != 0u);
                            w |= ((a & 15u) | (b & 15u) << 8) << (16 * pr);
                        }
                    } else {
#pragma unroll
                        for (int pr = 0; pr < 2; pr++) {       // both symbols searched in the table before the pair
                            const bool act0 = coded && q0 + 2u * (u32)pr < len, act1 = coded && q0 + 2u * (u32)pr + 1u < len;
                            u32 a0, a1, b0, b1;
                            const u32 x0 = trc_nib_search(T0, d0.scaled(), a0, a1), x1 = trc_nib_search(T0, d1.scaled(), b0, b1);
                            d0.consume_if(s0, act0, a0, a1);
                            d1.consume_if(s1, act1, b0, b1);
                            m.adapt(T0, x0); m.adapt(T0, x1);
                            w |= (x0 | x1 << 8) << (16 * pr);
                        }
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

// ---- the byte coders' DECODER as two waves per 64 chunks (round 4) -------------------------------------------------
// A decoder cannot run ahead of its model: the symbol comes out of the table, the table moves with the symbol.  But the
// two halves of that loop are different work -- search + range update on one side (needs the table as it IS), the 24
// packed operations of cdf16upd on the other (needs the symbol) -- and the hi and lo tables of a byte alternate: while
// the lo nibble is searched, the hi table can adapt, and the other way round.  Wave D (decode) owns the range coder and
// the stream and READS tables from the LDS rows; wave M (model) owns the tables and adapts them:
//     phase A   D: load hi table, search, range update, h -> mailbox     M: lo table of the byte before adapts by its l
//     ---- s_barrier ----
//     phase B   D: load lo table (h), search, range update, l -> mailbox M: hi table adapts by h; lo table (h) loaded
//     ---- s_barrier ----
// Two LDS-only barriers per byte; D's path per byte is two searches and two range updates (~110 instructions instead of
// ~200), M's 64 run in its issue gaps on the same SIMD (workgroup = 4 D waves + 4 M waves: pair k on SIMD k).
#define RCA_DMC_MBOX   512u                                    // h and l, one dword per lane each
#define RCA_DMC_LDS    (TRC_NIB_BYTES + RCA_DMC_MBOX)
template <int NS>
__global__ __launch_bounds__(128 * TRC_WPG) void trc_rca_dec_mc_kernel(
    const u8 *__restrict__ payload, const u32 *__restrict__ clen, const u64 *__restrict__ goff, const u32 *__restrict__ gsum,
    u64 n, u32 chunk, u32 nchunks, u8 *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem_wg_[];
    const u32 wv_ = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool model = wv_ >= TRC_WPG;
    const u32 grp_ = blockIdx.x * TRC_WPG + (wv_ & (TRC_WPG - 1u));
    if (grp_ >= (nchunks + 63u) / 64u) return;                 // (both waves of the pair)
    u8 *const smem = smem_wg_ + (wv_ & (TRC_WPG - 1u)) * RCA_DMC_LDS;
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
        // ---- wave M: the same walk (same trip counts: the barriers pair up), adapting what wave D has just decoded
        NibTable T0 = m.load(m.table(0)), TL = T0;
        u32 hp = 0;                                            // the lo table TL holds: 1 + hp (nothing to write back before the first byte)
        bool have = false;
        for (u32 s = 0; s < S; s++) {
#pragma nounroll
            for (u32 k = 0; k < 4; k++) {
                const u32 p0 = s * TRC_SEG + k * 16u;
                if (!__ballot(coded && p0 < len)) continue;
#pragma nounroll
                for (u32 b = 0; b < 16u; b++) {
                    if (have) {                                // phase A: the byte before's lo table
                        const u32 l = *(const __attribute__((address_space(3))) u32 *)(uintptr_t)(mb + 256u);
                        m.adapt(TL, l & 15u); m.store(m.table(1u + hp), TL);
                    }
                    have = true;
                    trc_lds_barrier();
                    const u32 h = *(const __attribute__((address_space(3))) u32 *)(uintptr_t)mb & 15u;   // phase B
                    m.adapt(T0, h); m.store(m.table(0), T0);
                    TL = m.load(m.table(1u + h)); hp = h;
                    trc_lds_barrier();
                }
            }
        }
        return;
    }

    // ---- wave D
    const u32 ex = trc_wave_incl_scan(cl) - cl;
    const u64 off = trc_group_base(goff, gsum, wc.c0 >> 6) + ex;
    LaneIn<4> s0, s1;
    const u32 len0 = (NS == 2 && coded) ? trc_min(trc_ld32_a2(payload + off), trc_sub_sat(cl, 4u)) : 0u;   // a corrupt header cannot point outside the chunk's payload
    s0.prime(payload + off + (NS == 2 ? 4u : 0u), coded, NS == 2 ? len0 : cl);
    s1.prime(payload + off + 4u + len0, NS == 2 && coded, trc_sub_sat(cl, 4u + len0));
    RcDec d0, d1;
    { const u32 a = s0.peek32(); s0.skip_if(coded); const u32 b = s0.peek32(); s0.skip_if(coded); d0.start(a, b); }
    { const u32 a = s1.peek32(); s1.skip_if(NS == 2 && coded); const u32 b = s1.peek32(); s1.skip_if(NS == 2 && coded); d1.start(a, b); }

    // a symbol against the table at `tb` as the model wave left it; bit 4 of the result says "renormalised"
    auto get = [&](RcDec &dq, u32 w, const u8 *tb, bool act) __attribute__((always_inline)) -> u32 {
        const NibTable T = m.load(tb);
        u32 c0, c1;
        const u32 x = trc_nib_search(T, dq.scaled(), c0, c1);
        const bool rn = dq.consume_w(act, c0, c1, w);
        return x | (rn ? 16u : 0u);
    };
    // one byte: phase A (hi), barrier, phase B (lo), barrier
    auto get_byte = [&](RcDec &dh, RcDec &dl, u32 wh, u32 wl, bool act, u32 &rnh, u32 &rnl) __attribute__((always_inline)) -> u32 {
        const u32 h = get(dh, wh, m.table(0), act);
        *(__attribute__((address_space(3))) u32 *)(uintptr_t)mb = h;
        trc_lds_barrier();
        const u32 l = get(dl, wl, m.table(1u + (h & 15u)), act);
        *(__attribute__((address_space(3))) u32 *)(uintptr_t)(mb + 256u) = l;
        trc_lds_barrier();
        rnh = h & 16u; rnl = l & 16u;
        return (h & 15u) << 4 | (l & 15u);
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
                    if (NS == 1) {
#pragma nounroll
                        for (u32 i = 0; i < 4; i++) {          // both symbols of a byte share one look-ahead word (at most one renormalises)
                            const bool act = coded && q0 + i < len;
                            const uint4 pre = s0.prefetch();
                            const u32 sw = s0.peek32();
                            u32 rh, rl;
                            const u32 x = get_byte(d0, d0, sw, sw, act, rh, rl);
                            s0.advance_pre((rh | rl) >> 2, pre);
                            w |= x << (8u * i);
                        }
                    } else {
#pragma nounroll
                        for (u32 pr = 0; pr < 2; pr++) {       // a stream advances once per PAIR of its symbols
                            const bool acta = coded && q0 + 2u * pr < len, actb = coded && q0 + 2u * pr + 1u < len;
                            const uint4 pre0 = s0.prefetch(), pre1 = s1.prefetch();
                            const u32 w0 = s0.peek32(), w1 = s1.peek32();
                            u32 rha, rla, rhb, rlb;
                            const u32 xa = get_byte(d0, d1, w0, w1, acta, rha, rla);
                            const u32 xb = get_byte(d0, d1, w0, w1, actb, rhb, rlb);
                            s0.advance_pre((rha | rhb) >> 2, pre0); s1.advance_pre((rla | rlb) >> 2, pre1);
                            w |= (xa | xb << 8) << (16u * pr);
                        }
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

template <int NS, bool NIB>
static void launch_rca_enc(const uint8_t *d_in, size_t n, uint32_t chunk, const TrcWork &w, uint32_t *d_clen, hipStream_t s)
{
    if (NIB && trc_nib_big(w.ngroups)) {                       // one 12-wave workgroup per CU, pace-keeping (trc_dev.h)
        TRC_RAISE_LDS_ONCE((trc_rca_enc_kernel<NS, NIB>), TRC_LDS_ONE_PER_CU);
        TRC_LAUNCH_TIMED((trc_rca_enc_kernel<NS, NIB>), dim3((w.ngroups + TRC_NIB_WPG - 1u) / TRC_NIB_WPG), dim3(64 * TRC_NIB_WPG), TRC_LDS_ONE_PER_CU, s,
                           d_in, (u64)n, chunk, w.nchunks, w.scratch, w.stride, w.scratch2, w.stride2, d_clen, w.gsum);
        return;
    }
    TRC_RAISE_LDS_ONCE((trc_rca_enc_kernel<NS, NIB>), NIB ? TRC_LDS_ONE_PER_CU : TRC_WPG * RCA_WAVE_LDS(NIB));   // (one limit for both shapes of the nibble form: the attribute is set once per call site)
    TRC_LAUNCH_TIMED((trc_rca_enc_kernel<NS, NIB>), TRC_QUAD_GRID(w.ngroups), dim3(64 * TRC_WPG), TRC_WPG * (RCA_WAVE_LDS(NIB)), s,
                       d_in, (u64)n, chunk, w.nchunks, w.scratch, w.stride, w.scratch2, w.stride2, d_clen, w.gsum);
}
template <int NS, bool NIB>
static void launch_rca_dec(const uint8_t *d_payload, const uint32_t *d_clen, size_t n, uint32_t chunk,
                           const TrcWork &w, uint8_t *d_out, hipStream_t s)
{
    if (NIB && trc_nib_big(w.ngroups)) {
        TRC_RAISE_LDS_ONCE((trc_rca_dec_kernel<NS, NIB>), TRC_LDS_ONE_PER_CU);
        TRC_LAUNCH_TIMED((trc_rca_dec_kernel<NS, NIB>), dim3((w.ngroups + TRC_NIB_WPG - 1u) / TRC_NIB_WPG), dim3(64 * TRC_NIB_WPG), TRC_LDS_ONE_PER_CU, s,
                           d_payload, d_clen, w.goff, w.gsum, (u64)n, chunk, w.nchunks, d_out, (u32 *)nullptr, (u32 *)nullptr, 0u);
        return;
    }
    TRC_RAISE_LDS_ONCE((trc_rca_dec_kernel<NS, NIB>), NIB ? TRC_LDS_ONE_PER_CU : TRC_WPG * RCA_WAVE_LDS(NIB));   // (one limit for both shapes of the nibble form: the attribute is set once per call site)
    TRC_LAUNCH_TIMED((trc_rca_dec_kernel<NS, NIB>), TRC_QUAD_GRID(w.ngroups), dim3(64 * TRC_WPG), TRC_WPG * (RCA_WAVE_LDS(NIB)), s,
                       d_payload, d_clen, w.goff, w.gsum, (u64)n, chunk, w.nchunks, d_out, NIB ? nullptr : trc_prog_tls.counters, NIB ? nullptr : trc_prog_tls.host_flags, NIB ? 0u : trc_prog_tls.part);
}
// TRC_RCA_MC=0 selects the one-wave encoder of rounds 1-3 for the byte coders (A/B measurements, tests of both forms)
static bool rca_mc_enabled()
{
    static const int env = getenv("TRC_RCA_MC") ? atoi(getenv("TRC_RCA_MC")) : 1;
    return env != 0;
}
template <int NS>
static void launch_rca_enc_mc(const uint8_t *d_in, size_t n, uint32_t chunk, const TrcWork &w, uint32_t *d_clen, hipStream_t s)
{
    TRC_RAISE_LDS_ONCE(trc_rca_enc_mc_kernel<NS>, TRC_WPG * RCA_MC_LDS);
    TRC_LAUNCH_TIMED((trc_rca_enc_mc_kernel<NS>), TRC_QUAD_GRID(w.ngroups), dim3(192 * TRC_WPG), TRC_WPG * RCA_MC_LDS, s,
                       d_in, (u64)n, chunk, w.nchunks, w.scratch, w.stride, w.scratch2, w.stride2, d_clen, w.gsum, trc_gate_tls.flag, trc_gate_tls.part);
}
bool trc_rca_enc_gate_ok() { return rca_mc_enabled(); }
void trc_launch_rca_enc(int nstreams, int nibble, const uint8_t *d_in, size_t n, uint32_t chunk, const TrcWork &w, uint32_t *d_clen, hipStream_t s)
{
    if (!nibble && rca_mc_enabled()) {
        if (nstreams == 2) launch_rca_enc_mc<2>(d_in, n, chunk, w, d_clen, s); else launch_rca_enc_mc<1>(d_in, n, chunk, w, d_clen, s);
        return;
    }
    if (nibble) { if (nstreams == 2) launch_rca_enc<2, true>(d_in, n, chunk, w, d_clen, s); else launch_rca_enc<1, true>(d_in, n, chunk, w, d_clen, s); }
    else        { if (nstreams == 2) launch_rca_enc<2, false>(d_in, n, chunk, w, d_clen, s); else launch_rca_enc<1, false>(d_in, n, chunk, w, d_clen, s); }
}
template <int NS>
static void launch_rca_dec_mc(const uint8_t *d_payload, const uint32_t *d_clen, size_t n, uint32_t chunk,
                              const TrcWork &w, uint8_t *d_out, hipStream_t s)
{
    TRC_RAISE_LDS_ONCE(trc_rca_dec_mc_kernel<NS>, TRC_WPG * RCA_DMC_LDS);
    TRC_LAUNCH_TIMED((trc_rca_dec_mc_kernel<NS>), TRC_QUAD_GRID(w.ngroups), dim3(128 * TRC_WPG), TRC_WPG * RCA_DMC_LDS, s,
                       d_payload, d_clen, w.goff, w.gsum, (u64)n, chunk, w.nchunks, d_out);
}
// TRC_RCA_DMC=1 selects the two-wave decoder.  MEASURED AND NOT USED (profiles/r04_notes.md): bit-exact, and 1.5x SLOWER than the
// one-wave decoder (rccdf 100 MB: 1.04 ms against 0.68) -- two barriers and two dependent table loads per byte leave wave D
// waiting longer than the 24 packed operations it hands over took.
static bool rca_dmc_enabled()
{
    static const int env = getenv("TRC_RCA_DMC") ? atoi(getenv("TRC_RCA_DMC")) : 0;
    return env != 0;
}
bool trc_rca_dec_prog_ok() { return !rca_dmc_enabled(); }         // the one-wave decoder (the default) reports its progress
void trc_launch_rca_dec(int nstreams, int nibble, const uint8_t *d_payload, const uint32_t *d_clen, size_t n, uint32_t chunk,
                        const TrcWork &w, uint8_t *d_out, hipStream_t s)
{
    if (!nibble && rca_dmc_enabled()) {
        if (nstreams == 2) launch_rca_dec_mc<2>(d_payload, d_clen, n, chunk, w, d_out, s); else launch_rca_dec_mc<1>(d_payload, d_clen, n, chunk, w, d_out, s);
        return;
    }
    if (nibble) { if (nstreams == 2) launch_rca_dec<2, true>(d_payload, d_clen, n, chunk, w, d_out, s); else launch_rca_dec<1, true>(d_payload, d_clen, n, chunk, w, d_out, s); }
    else        { if (nstreams == 2) launch_rca_dec<2, false>(d_payload, d_clen, n, chunk, w, d_out, s); else launch_rca_dec<1, false>(d_payload, d_clen, n, chunk, w, d_out, s); }
}
