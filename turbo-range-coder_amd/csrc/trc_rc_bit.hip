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
// chunk bytes move through in-register quad transposes (trc_io.h QuadIn/QuadOut), the coded stream through per-lane
// registers (trc_lane_io.h: released words are stored directly; the decoder carries two look-ahead words and makes one
// 16-byte load from its stream position per byte -- end of round 4; rounds 2-4: a 32-byte window), so five waves share a
// CU: one wave per SIMD, where a kernel's time is its instruction count, scalar mask logic, branches and the wait states
// behind a vector-written mask included (profiles/r02_notes.md, r04_notes.md 14).  The two sides:
//   encoder  the eight nodes a byte visits are known from the byte itself ((0x100|x) >> (8-k)) and are all different,
//            so their probabilities are read in one batch, adapted with packed 16-bit arithmetic and written back before
//            the eight coding steps run on registers (state on 32-bit halves, mask selects, one emit per byte);
//   decoder  the path depends on the decoded bits; both children of the current node are requested before the bit is
//            resolved, the bit is the borrow of code - cut and everything that depends on it a bit-select under the mask made
//            from it -- one hand-written block per bit (the compiler turns the C form back into v_cndmask on an SGPR pair).
// Round 4: the encoder is a model wave + a coder wave per 64 chunks (trc_rcb_enc_mc_kernel, the default).
// Round 3, ENCODER only: the deepest tree level (nodes 128..255: half of the model) lives in global memory, one 256-byte row per
// lane, filled by the wave itself at its start -- 16 KiB of model per wave in LDS instead of 32: nine waves per CU where five fit,
// i.e. more than two per SIMD, which is what fills the issue gaps of a one-wave-per-SIMD kernel (round 2's half-model ablation:
// 1.4x).  The encoder knows a byte's level-7 node when it has the byte, so the 2-byte load is issued with the LDS reads and is
// back long before the last bit is coded; the decoder learns that node only from the seventh bit it has just decoded, a memory
// round trip on its critical path per byte, and keeps the whole model in LDS.  Measured: it buys residency, not throughput (the launch code says when it is used).
#include <stdlib.h>
#include "trc_rc.h"
#include "trc_nibmodel.h"
#include "trc_lane_io.h"
#include "trc_launch.h"

#define RCB_A(x) (x)
#define RCB_MODEL_BYTES (256u * 64u * 2u)                  // [ctx][lane] u16
#define RCB_AMASK 0x7fffu
#define RCB_WAVE_LDS    RCB_MODEL_BYTES
#define RCB_ENC_WAVE_LDS(L7G) ((L7G) ? RCB_MODEL_BYTES / 2u : RCB_MODEL_BYTES)

// (a & m) | (b & ~m) as the one instruction it is (from the C form the compiler builds and / and-or pairs, or compares and selects)
__device__ __forceinline__ u32 rcb_bfi(u32 m, u32 a, u32 b)
{
    u32 r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ u32 rcb_adapt(u32 p, u32 bit) { return (p - (((p - (bit << TRC_PROB_BITS)) >> 5) + bit)) & 0xffffu; }

template <bool L7G>
__global__ __launch_bounds__(64 * TRC_WPG) void trc_rcb_enc_kernel(
    const u8 *__restrict__ in, u64 n, u32 chunk, u32 nchunks,
    u8 *__restrict__ scratch, u32 stride, u8 *__restrict__ level7, u32 *__restrict__ clen, u32 *__restrict__ gsum)
{
    TRC_QUAD_PROLOGUE(RCB_ENC_WAVE_LDS(L7G));
    u16 *mb = (u16 *)smem + lane;                              // mb[ctx * 64]
    for (u32 i = 0; i < RCB_ENC_WAVE_LDS(L7G) / 128u; i++) mb[i * 64] = (u16)(TRC_PROB_ONE >> 1);
    u16 *l7 = nullptr;
    if constexpr (L7G) {
    // this wave's level-7 rows: 64 lanes x 256 B, contiguous -- filled with coalesced 16-byte stores (lane l: bytes 16 l + 1024 i),
    // then every lane works on its own row.  The loads below see these stores: same wave, the stores are waited for.
    u8 *l7w = level7 + (u64)grp_ * (64u * 256u);
    {
        const u32 h = (TRC_PROB_ONE >> 1) | (TRC_PROB_ONE >> 1) << 16;
#pragma unroll
        for (u32 i = 0; i < 16u; i++) *(uint4 *)(l7w + lane * 16u + i * 1024u) = make_uint4(h, h, h, h);
        __builtin_amdgcn_s_waitcnt(0x0f70);                    // vmcnt(0) (the compiler does not order the plain stores against the u16 loads of other lanes' bytes)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    }
    l7 = (u16 *)(l7w + lane * 256u);                           // l7[node - 128]
    }

    WaveChunks wc;
    wc.c0 = grp_ * 64u; wc.chunk = chunk; wc.nchunks = nchunks;
    wc.lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    wc.rows = nchunks - wc.c0 < 64u ? nchunks - wc.c0 : 64u;
    const bool alive = lane < wc.rows;
    const u32 c = wc.c0 + lane;
    const u32 len = alive ? wc.len_of(lane) : 0u;
    const int lim = trc_rc_limit(len);

    QuadIn qin; qin.base = in + (u64)wc.c0 * chunk;
    LaneOutDirect so; so.start(scratch + (u64)c * stride);
    // The coder state of rcbe_ (turborc_.h:417-421) as 32-bit halves: range = rhi:rlo, low = lhi:llo.  The reference finds a
    // carry by comparing `low` with its value at the last renormalisation; here the carry-out of every `low +=` is ORed
    // into a lane mask (`cy`, an SGPR pair: SALU work), which is the same event -- between two renormalisations `low`
    // grows by less than the range it had at the first, so it wraps at most once.
    u32 rlo = ~0u, rhi = ~0u, llo = 0, lhi = 0;
    bool cy = false;
    TrcCarry cw; cw.start();
    bool ovf = alive && lim <= 0;
    // `live`: this lane is still coding.  A lane that is not (dead lane, incompressible chunk, past the end of a short
    // last chunk) keeps running the same arithmetic on its own registers and its own model column -- nothing of it is
    // observable, because only live lanes emit words -- so the eight steps of a byte carry no per-lane predication at
    // all.  A lane whose chunk ends before the wave's does (the input's last chunk) flushes at that byte boundary.
    bool live = alive && !ovf;
    u32 out_len = alive ? len : 0u;                            // raw until proven otherwise

    // One byte = 4 x (renormalisation point + two bits), unrolled.  With one wave per SIMD (32 KiB of model per wave) the
    // kernel's time is its instruction count -- scalar mask logic and branches included (profiles/r02_notes.md) -- so:
    //  * everything that does not depend on the coder state is done for the whole byte up front: the eight nodes of the
    //    byte's path ((0x100|x) >> (8-k)) are all different, so their probabilities are read in one batch, adapted two at
    //    a time with packed 16-bit arithmetic (the update p -= ((p - (bit<<15)) >> 5) + bit keeps only 16 bits: in 16-bit
    //    lanes the logical shift of the 32-bit form is an arithmetic one) and written back where they came from;
    //  * a bit step is ~12 VALU operations: cut = (range >> 15) * p as alignbit / shift / 32x32->64 mad / 24-bit mad; bit
    //    selects as and-or under sign masks of the bit (no VCC); low += (bit ? 0 : cut) keeping the carry-out;
    //    range = bit ? cut : range - cut;
    //  * a renormalisation point only SHIFTS the state and leaves behind (mask, word, carry) in registers of its own; the
    //    emit logic (held-back word, carry, store) runs ONCE per byte for the one word a lane has at most -- a lane emits a
    //    word every ~6 bytes.  Lanes with two or more words in one byte (>= 32 bits of range spent on <= 6 bits) are
    //    found by one test per byte and served point by point behind a wave-uniform branch.
    const u32 mcol = trc_lds_addr(smem) + lane * 2u;           // this lane's model column as an LDS byte address
    auto put_byte = [&](u32 x) {
        const u32 t = 0x100u | x;
        u32 ad[8];
        ad[0] = mcol + 128u;
#pragma unroll
        for (int k = 1; k < 8; k++) ad[k] = (((t >> (8 - k)) << 7) & RCB_AMASK) + mcol;
        u32 pr[8];
        u16 *g7 = l7 + ((t >> 1) & 127u);                      // (L7G) the byte's level-7 node
        if constexpr (L7G) pr[7] = *g7; else pr[7] = trc_ldsr16(ad[7]);
#pragma unroll
        for (int k = 0; k < 7; k++) pr[k] = trc_ldsr16(ad[k]);
        u32 P[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            P[j] = pr[2 * j] | pr[2 * j + 1] << 16;
            const u32 B = __builtin_amdgcn_ubfe(x, 7 - 2 * j, 1) | __builtin_amdgcn_ubfe(x, 6 - 2 * j, 1) << 16;    // the pair's two bits, one per half
            const trc_s2 pv = trc_as_s2(P[j]), bv = trc_as_s2(B);
            const trc_s2 np = pv - (((pv - trc_as_s2(B << 15)) >> (trc_s2)5) + bv);
            const u32 NP = trc_as_u32(np);
            trc_ldsw16(ad[2 * j], NP);
            if (L7G && j == 3) *g7 = (u16)(NP >> 16); else trc_ldsw16(ad[2 * j + 1], NP >> 16);
        }
        bool rnj[4], cyj[4];
        u32 pwj[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            {                                                  // renorm before bits 7,5,3,1 only (_RCENORM2)
                const bool rn = rhi == 0u;
                rnj[j] = rn; cyj[j] = rn && cy; pwj[j] = lhi;
                cy = cy && !rn;
                lhi = rn ? llo : lhi; llo = rn ? 0u : llo;
                rhi = rn ? rlo : rhi; rlo = rn ? 0u : rlo;
            }
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int k = 2 * j + h;
                const u32 prob = h ? P[j] >> 16 : P[j] & 0xffffu;
                const u32 m = (u32)__builtin_amdgcn_sbfe((int)x, 7 - k, 1);        // bit 1: all ones
                const u32 slo = __builtin_amdgcn_alignbit(rhi, rlo, TRC_PROB_BITS), shi = rhi >> TRC_PROB_BITS;
                const u64 c64 = (u64)slo * prob;
                const u32 clo = (u32)c64, chi = __umul24(shi, prob) + (u32)(c64 >> 32);    // shi < 2^17, prob < 2^16
                u32 k1, k2;
                llo = __builtin_addc(llo, rcb_bfi(m, 0u, clo), 0u, &k1);           // low += bit ? 0 : cut
                lhi = __builtin_addc(lhi, rcb_bfi(m, 0u, chi), k1, &k2);
                cy = cy || (k2 != 0u);
                u32 b1, b2;
                const u32 tlo = __builtin_subc(rlo, clo, 0u, &b1), thi = __builtin_subc(rhi, chi, b1, &b2);
                rlo = rcb_bfi(m, clo, tlo); rhi = rcb_bfi(m, chi, thi);             // range = bit ? cut : range - cut
            }
        }
        const bool two = (rnj[0] && (rnj[1] || rnj[2] || rnj[3])) || (rnj[1] && (rnj[2] || rnj[3])) || (rnj[2] && rnj[3]);
        if (__ballot(two && live)) {                           // rare: some lane has several words in this byte
#pragma unroll
            for (int j = 0; j < 4; j++) cw.emit_if(so, rnj[j] && live, cyj[j], pwj[j]);
        } else {
            const bool pend = rnj[0] || rnj[1] || rnj[2] || rnj[3];
            const bool pcy = cyj[0] || cyj[1] || cyj[2] || cyj[3];
            const u32 pw = rnj[3] ? pwj[3] : rnj[2] ? pwj[2] : rnj[1] ? pwj[1] : pwj[0];
            cw.emit_if(so, pend && live, pcy, pw);
        }
    };
    // rceflush (turborc_.h:118-128) on the same state, then everything still held back
    auto finish = [&]() {
        u64 low = ((u64)lhi << 32) | llo;
        u64 rg = ((u64)rhi << 32) | rlo;
        bool c0 = cy;
        if (rg < TRC_TOP32) { cw.emit(so, c0, (u32)(low >> 32)); low <<= 32; rg <<= 32; c0 = false; }
        if (rg > ((u64)1 << 33)) {
            const u64 nl = low + TRC_TOP32;
            cw.emit(so, c0 || nl < low, (u32)(nl >> 32));
        } else {
            const u64 nl = low + 1;
            cw.emit(so, c0 || nl < low, (u32)(nl >> 32));
            cw.emit(so, false, (u32)nl);
        }
        cw.release(so);
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
            const u32 p0 = s * TRC_SEG + k * 16u;
            const bool ends = __ballot(live && len - p0 < 16u) != 0;     // a short last chunk ends inside this piece (once per grid)
#pragma nounroll
            for (u32 d = 0; d < 4; d++) {
                const u32 w = v.x; v.x = v.y; v.y = v.z; v.z = v.w;
                const u32 q0 = p0 + d * 4u;
#pragma nounroll
                for (u32 i = 0; i < 4; i++) {
                    if (ends && live && q0 + i == len) {       // decide and flush it at its byte boundary
                        if ((int)(4u * cw.nwords) < lim) { finish(); out_len = so.wpos; }
                        live = false;                          // (else: incompressible, out_len stays the raw length)
                    }
                    put_byte((w >> (8 * i)) & 255u);
                }
                ovf = ovf || (live && (int)(4u * cw.nwords) >= lim);           // OVERFLOW per dword, monotone
                live = live && !ovf;
            }
        }
    }
    if (live) { finish(); out_len = so.wpos; }
    if (alive) clen[c] = out_len;
    const u32 gs = trc_wave_sum(out_len);
    if (lane == 0) gsum[wc.c0 >> 6] = gs;
}

// ---- the encoder as TWO WAVES per 64 chunks (round 4; the scheme of trc_rca_enc_mc_kernel, trc_rc_adaptive.hip) --------
// Nothing of the model side of a byte depends on the coder: the eight nodes are a function of the byte, their
// probabilities a function of the bytes before it.  Wave 0 owns the model (32 KiB of LDS, the input bytes): per byte it
// reads the eight probabilities, adapts them, and pushes four RECORDS {p | bit << 15} x 2 (one per half of a dword: a
// probability has 15 bits) into a double-buffered LDS queue, two bytes = 32 B per lane and period.  Wave 1 owns the range
// coder (state on 32-bit halves, carry logic, output) and pops them one period later; the bits ride in the records, so
// it never sees the input.  One LDS-only s_barrier per period.  36 KiB per workgroup: four per CU, two waves per SIMD.
#define RCB_MC_QUEUE   (2u * 2u * 64u * 16u)                   // [buffer][byte of the period][lane][16 B]
#define RCB_MC_LDS     (RCB_MODEL_BYTES + RCB_MC_QUEUE)

__global__ __launch_bounds__(128 * TRC_WPG) void trc_rcb_enc_mc_kernel(
    const u8 *__restrict__ in, u64 n, u32 chunk, u32 nchunks,
    u8 *__restrict__ scratch, u32 stride, u32 *__restrict__ clen, u32 *__restrict__ gsum, const u32 *__restrict__ gate, u32 gate_part)
{
    // a workgroup = 4 model waves (0-3) + 4 coder waves (4-7): pair k = waves k and k + 4 on SIMD k (trc_dev.h, TRC_WPG)
    extern __shared__ __attribute__((aligned(16))) u8 smem_wg_[];
    const u32 wv_ = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool coder = wv_ >= TRC_WPG;
    const u32 grp_ = blockIdx.x * TRC_WPG + (wv_ & (TRC_WPG - 1u));
    if (grp_ >= (nchunks + 63u) / 64u) return;                 // (both waves of the pair: a finished wave no longer counts at s_barrier)
    u8 *const smem = smem_wg_ + (wv_ & (TRC_WPG - 1u)) * RCB_MC_LDS;
    const u32 lane = trc_lane();
    const u32 qa = trc_lds_addr(smem) + RCB_MODEL_BYTES + lane * 16u;

    WaveChunks wc;
    wc.c0 = grp_ * 64u; wc.chunk = chunk; wc.nchunks = nchunks;
    wc.lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    wc.rows = nchunks - wc.c0 < 64u ? nchunks - wc.c0 : 64u;
    wc.gate = gate; wc.gate_part = gate_part;                  // (host-pointer encodes: the input arrives while the waves code, trc_io.h)
    const u32 S = chunk / TRC_SEG;

    if (!coder) {
        // ---- wave 0: the model
        u16 *mb = (u16 *)smem + lane;                          // mb[ctx * 64]
        for (u32 i = 0; i < RCB_MODEL_BYTES / 128u; i++) mb[i * 64] = (u16)(TRC_PROB_ONE >> 1);
        const u32 mcol = trc_lds_addr(smem) + lane * 2u;
        // the records of one byte: R[j] = {p(node of bit 7-2j) | bit << 15, p(node of bit 6-2j) | bit << 15} before adaptation
        auto model_byte = [&](u32 x, u32 qaddr) __attribute__((always_inline)) {
            const u32 t = 0x100u | x;
            u32 ad[8];
            ad[0] = mcol + 128u;
#pragma unroll
            for (int k = 1; k < 8; k++) ad[k] = (((t >> (8 - k)) << 7) & RCB_AMASK) + mcol;
            u32 pr[8];
#pragma unroll
            for (int k = 0; k < 8; k++) pr[k] = trc_ldsr16(ad[k]);
            u32 R[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const u32 P = pr[2 * j] | pr[2 * j + 1] << 16;
                const u32 B = __builtin_amdgcn_ubfe(x, 7 - 2 * j, 1) | __builtin_amdgcn_ubfe(x, 6 - 2 * j, 1) << 16;
                const trc_s2 pv = trc_as_s2(P), bv = trc_as_s2(B);
                const trc_s2 np = pv - (((pv - trc_as_s2(B << 15)) >> (trc_s2)5) + bv);
                const u32 NP = trc_as_u32(np);
                trc_ldsw16(ad[2 * j], NP);
                trc_ldsw16(ad[2 * j + 1], NP >> 16);
                R[j] = P | B << 15;
            }
            trc_ldsw128(qaddr, make_uint4(R[0], R[1], R[2], R[3]));
        };
        QuadIn qin; qin.base = in + (u64)wc.c0 * chunk;
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
#pragma unroll
                    for (u32 h = 0; h < 2; h++) {
                        const u32 a = qa + buf * 2048u;
                        model_byte((w >> (16 * h)) & 255u, a);
                        model_byte((w >> (16 * h + 8)) & 255u, a + 1024u);
                        trc_lds_barrier();
                        buf ^= 1u;
                    }
                }
            }
        }
        return;
    }

    // ---- wave 1: the range coder (the state and the emit logic of trc_rcb_enc_kernel), one period behind
    const bool alive = lane < wc.rows;
    const u32 c = wc.c0 + lane;
    const u32 len = alive ? wc.len_of(lane) : 0u;
    const int lim = trc_rc_limit(len);
    LaneOutDirect so; so.start(scratch + (u64)c * stride);
    u32 rlo = ~0u, rhi = ~0u, llo = 0, lhi = 0;
    u32 lx = 0;                                                // carry limb of `low` (0 / 1): see code_byte
    TrcCarry cw; cw.start();
    bool ovf = alive && lim <= 0;
    bool live = alive && !ovf;
    u32 out_len = alive ? len : 0u;

    // One byte = 4 x (renormalisation point + two bits).  Bookkeeping on the VECTOR side only (the ablation without the emit logic
    // ran the encoder in 0.91 instead of 1.33 ms: a third of the coder wave was mask algebra in SGPRs -- two / pend / carry flags,
    // each a VALU -> SALU -> VALU round trip -- and one emit per byte; profiles/r04_notes.md):
    //  * the carry of `low +=` is a third limb `lx` (0 / 1) fed by the add's carry-out (v_addc), not an ORed lane mask;
    //  * a renormalisation point shifts the state and pushes one bit into `nrnb` ("there is NO word": complemented, round 5) and one into `cyb` ("it carries
    //    into the words before it"), keeps the word in `pw` (and every point's word in pwj[] for the rare several-words case);
    //  * the emit logic runs ONCE PER PERIOD of two bytes (a lane emits a word every ~6 bytes): popcount(rnb) >= 2 in some lane is
    //    the wave-uniform rare path that replays the points in order.
    auto code_byte = [&](const uint4 rec, u32 &nrnb, u32 &cyb, u32 &pw, u32 (&pwj)[4]) __attribute__((always_inline)) {
        const u32 R[4] = { rec.x, rec.y, rec.z, rec.w };
#pragma unroll
        for (int j = 0; j < 4; j++) {
            {
                // rn = (rhi == 0): t = min(rhi, 1) is its complement as a number, mr = t - 1 its mask; everything that hangs on it is
                // arithmetic or bit-select (hand-written: the compiler's form is a compare into an SGPR pair, ten selects on it and the
                // wait states between them).  No VCC: round 4 took mask and count off a carry chain (v_subrev_co / v_subb / v_addc back
                // to back), a VALU read of VCC straight behind the VALU write of it -- the compiler keeps two wait states there on
                // gfx950 (scripts/check_isa_hazards.py).  The point's flag goes into `nrnb` COMPLEMENTED (bit = "no word here").
                u32 mr, t, lhin;
                pwj[j] = lhi;
                asm("v_min_u32_e32 %1, 1, %7\n\t"
                    "v_add_u32_e32 %0, -1, %1\n\t"                   // mr = -rn
                    "v_lshl_add_u32 %3, %3, 1, %1\n\t"               // nrnb = 2 nrnb + !rn
                    "v_and_b32_e32 %1, %0, %6\n\t"
                    "v_lshl_add_u32 %4, %4, 1, %1\n\t"                // cyb = 2 cyb + (rn ? lx : 0)
                    "v_bfi_b32 %5, %0, %10, %5\n\t"                   // pw = rn ? lhi : pw
                    "v_bfi_b32 %6, %0, 0, %6\n\t"                     // lx = rn ? 0 : lx
                    "v_bfi_b32 %2, %0, %9, %10\n\t"                   // lhi = rn ? llo : lhi
                    "v_bfi_b32 %9, %0, 0, %9\n\t"                     // llo = rn ? 0 : llo
                    "v_bfi_b32 %7, %0, %8, %7\n\t"                    // rhi = rn ? rlo : rhi
                    "v_bfi_b32 %8, %0, 0, %8"                           // rlo = rn ? 0 : rlo
                    : "=&v"(mr), "=&v"(t), "=&v"(lhin), "+v"(nrnb), "+v"(cyb), "+v"(pw), "+v"(lx), "+v"(rhi), "+v"(rlo), "+v"(llo)
                    : "v"(lhi));
                lhi = lhin;
            }
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const u32 prob = h ? __builtin_amdgcn_ubfe(R[j], 16, 15) : R[j] & 0x7fffu;
                const u32 m = h ? (u32)((int)R[j] >> 31) : (u32)__builtin_amdgcn_sbfe((int)R[j], 15, 1);   // bit 1: all ones
                const u32 slo = __builtin_amdgcn_alignbit(rhi, rlo, TRC_PROB_BITS), shi = rhi >> TRC_PROB_BITS;
                const u64 c64 = (u64)slo * prob;
                const u32 clo = (u32)c64, chi = __umul24(shi, prob) + (u32)(c64 >> 32);
                // low += bit ? 0 : cut (carry into lx: at most one between two renormalisations), range = bit ? cut : range - cut;
                // one block.  The two carry chains run on two carry registers (VCC for `low +=`, an SGPR pair for `range -`) and are
                // interleaved with the bit-selects so that two instructions sit between every VALU write of a carry and the VALU read
                // of it -- the wait states the compiler keeps on gfx950 (its own v_add_co / v_addc carry an s_nop 1), here filled
                // with the block's own work: same nine instructions as the back-to-back form of round 4.
                u32 t0, t1, u0, u1;
                u64 sb;
                asm("v_bfi_b32 %[t0], %[m], 0, %[clo]\n\t"
                    "v_add_co_u32_e32 %[llo], vcc, %[llo], %[t0]\n\t"
                    "v_sub_co_u32_e64 %[u0], %[sb], %[rlo], %[clo]\n\t"
                    "v_bfi_b32 %[t1], %[m], 0, %[chi]\n\t"
                    "v_addc_co_u32_e32 %[lhi], vcc, %[lhi], %[t1], vcc\n\t"
                    "v_subb_co_u32_e64 %[u1], %[sb], %[rhi], %[chi], %[sb]\n\t"
                    "v_bfi_b32 %[rlo], %[m], %[clo], %[u0]\n\t"
                    "v_addc_co_u32_e32 %[lx], vcc, 0, %[lx], vcc\n\t"
                    "v_bfi_b32 %[rhi], %[m], %[chi], %[u1]"
                    : [t0] "=&v"(t0), [t1] "=&v"(t1), [u0] "=&v"(u0), [u1] "=&v"(u1), [sb] "=&s"(sb),
                      [llo] "+v"(llo), [lhi] "+v"(lhi), [lx] "+v"(lx), [rlo] "+v"(rlo), [rhi] "+v"(rhi)
                    : [m] "v"(m), [clo] "v"(clo), [chi] "v"(chi) : "vcc");
            }
        }
    };
    // the words of NP renormalisation points (rnb / cyb: point 0 in bit NP - 1), in order
    auto emit_points = [&](u32 rnb, u32 cyb, u32 pw, const u32 (&pa)[4], const u32 (&pb)[4], const int NP) __attribute__((always_inline)) {
        const u32 cnt = (u32)__builtin_popcount(rnb);
        if (__ballot(cnt >= 2u && live)) {                      // rare: some lane has several words in this period
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (j >= NP) break;
                const u32 bit = 1u << (NP - 1 - j);
                cw.emit_if(so, (rnb & bit) != 0u && live, (cyb & bit) != 0u, j < 4 ? pa[j & 3] : pb[j & 3]);
            }
        } else cw.emit_if(so, rnb != 0u && live, cyb != 0u, pw);
    };
    auto finish = [&]() __attribute__((always_inline)) {                                      // rceflush (turborc_.h:118-128), then everything still held back
        u64 low = ((u64)lhi << 32) | llo;
        u64 rg = ((u64)rhi << 32) | rlo;
        bool c0 = lx != 0u;
        if (rg < TRC_TOP32) { cw.emit(so, c0, (u32)(low >> 32)); low <<= 32; rg <<= 32; c0 = false; }
        if (rg > ((u64)1 << 33)) {
            const u64 nl = low + TRC_TOP32;
            cw.emit(so, c0 || nl < low, (u32)(nl >> 32));
        } else {
            const u64 nl = low + 1;
            cw.emit(so, c0 || nl < low, (u32)(nl >> 32));
            cw.emit(so, false, (u32)nl);
        }
        cw.release(so);
    };
    auto code_period = [&](u32 q0, u32 buf) __attribute__((always_inline)) {
        if (!__ballot(live)) return;
        const u32 a = qa + buf * 2048u;
        const uint4 r0 = trc_ldsr128(a), r1 = trc_ldsr128(a + 1024u);
        const bool ends = __ballot(live && len - q0 < 2u) != 0;        // a short last chunk ends inside this period (once per grid)
        u32 nrnb = 0, cyb = 0, pw = 0, pa[4], pb[4] = { 0, 0, 0, 0 };        // nrnb: one bit per point, SET where the point emits nothing
        if (ends && live && q0 == len) { if ((int)(4u * cw.nwords) < lim) { finish(); out_len = so.wpos; } live = false; }
        code_byte(r0, nrnb, cyb, pw, pa);
        if (ends) {                                             // (wave-uniform) the chunk may end between the two bytes: emit what byte 0 left first
            emit_points(~nrnb & 0xfu, cyb, pw, pa, pb, 4);
            nrnb = cyb = 0;
            if (live && q0 + 1u == len) { if ((int)(4u * cw.nwords) < lim) { finish(); out_len = so.wpos; } live = false; }
            code_byte(r1, nrnb, cyb, pw, pa);
            emit_points(~nrnb & 0xfu, cyb, pw, pa, pb, 4);
        } else {
            code_byte(r1, nrnb, cyb, pw, pb);
            emit_points(~nrnb & 0xffu, cyb, pw, pa, pb, 8);
        }
        ovf = ovf || (live && (int)(4u * cw.nwords) >= lim);           // OVERFLOW, monotone
        live = live && !ovf;
    };
    {
        const u32 P = S * 32u;                                 // periods of a full chunk = barriers of the model wave
        u32 buf = 0;
#pragma nounroll
        for (u32 p = 0; p <= P; p++) {                         // (one call site: the coder's body exists once)
            if (p) code_period((p - 1u) * 2u, buf ^ 1u);
            if (p < P) trc_lds_barrier();
            buf ^= 1u;
        }
    }
    if (live) { finish(); out_len = so.wpos; }
    if (alive) clen[c] = out_len;
    const u32 gs = trc_wave_sum(out_len);
    if (lane == 0) gsum[wc.c0 >> 6] = gs;
}

__global__ __launch_bounds__(64 * TRC_WPG) void trc_rcb_dec_kernel(
    const u8 *__restrict__ payload, const u32 *__restrict__ clen, const u64 *__restrict__ goff, const u32 *__restrict__ gsum,
    u64 n, u32 chunk, u32 nchunks, u8 *__restrict__ out, u32 *__restrict__ prog, u32 *__restrict__ prog_host, u32 prog_part)
{
    TRC_QUAD_PROLOGUE(RCB_WAVE_LDS);
    u16 *mb = (u16 *)smem + lane;
    for (u32 i = 0; i < RCB_MODEL_BYTES / 128u; i++) mb[i * 64] = (u16)(TRC_PROB_ONE >> 1);

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

    // rcbd_ (turborc_.h:447-452) on 32-bit halves: range = rhi:rlo, code = chi:clo (rcdinit: two words).
    // The stream side of a byte: ONE 16-byte load from the lane's stream position at the start of the byte (W: the words at rpos,
    // rpos + 4, + 8, + 12; requested ~1300 cycles before its first ordinary use).  Two look-ahead words w0 / w1 are carried in
    // registers: a renormalisation point takes w0 and moves w1 up -- no window select, no position arithmetic at the four points;
    // at the end of the byte the next pair is W[cnt], W[cnt + 1] (two bit-selects each).  A byte that renormalises more than
    // twice (>= 64 bits of range spent on <= 6 bits) takes W.z / W.w behind a wave-uniform test; one that takes more than two
    // words in all reloads behind another.  (Rounds 2-4 kept a 32-byte register window per lane, LaneIn: window refill, boundary
    // test and two three-level selects were ~24 instructions per byte; this is ~11.)
    const u8 *src = payload + off;
    const u32 lim = cl;                                        // no load from beyond this stream offset (corrupt input: re-reads the end)
    u32 rpos = 8u;
    u32 rlo = ~0u, rhi = ~0u, chi, clo, w0, w1;
    { const uint4 W0 = trc_ld16_a2(src); chi = W0.x; clo = W0.y; w0 = W0.z; w1 = W0.w; }
    u32 p1 = (u32)(TRC_PROB_ONE >> 1);                         // probability of node 1, read back at the end of every byte

    const u32 mcol = trc_lds_addr(smem) + lane * 2u;           // this lane's model column as an LDS byte address
    const u32 negm = 0u - mcol;
    auto get_byte = [&](bool act) -> u32 {
        const uint4 W = trc_ld16_a2(src + trc_min(rpos, lim));
        u32 a = mcol + 128u;                                   // LDS address of the current node (row stride 128 B)
        u32 c0 = mcol + 256u;                                  // row 2*ctx: the node's children
        u32 p = p1;
        u32 cnt = 0;                                           // words taken in this byte
        bool many = false;                                     // (wave-uniform) some lane may end the byte with more than two
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (!(k & 1)) {                                    // renorm before bits 7,5,3,1 only
                if (k >= 4) {
                    if (__ballot(cnt >= 2u)) {                 // rare: both look-ahead words are gone -- the window's upper half
                        w0 = cnt == 2u ? W.z : w0; w1 = cnt == 2u ? W.w : w1;      // (cnt == 3: W.w has moved up into w0 already)
                        many = true;
                    }
                }
                // (no predication on `act`: a lane that is not decoding -- raw chunk, dead lane, past the end of a short last chunk -- runs
                // along on its own registers, its own model column and whatever its clamped stream window holds, statistically like any
                // other lane; nothing of it is observable.  An `act &&` here is a mask operation between the compare and five selects.)
                // rn = (rhi == 0): t = min(rhi, 1), mask = t - 1, count -= mask, five bit-selects -- no SGPR mask (v_cmp + v_cndmask needs
                // two wait states between them on gfx950 and the compiler made a branch of some of these) and, since round 5, no
                // VCC either: round 4's v_subrev_co / v_subb / v_addc read VCC straight behind the VALU write of it.
                u32 mr, t;
                asm("v_min_u32_e32 %1, 1, %2\n\t"
                    "v_add_u32_e32 %0, -1, %1\n\t"
                    "v_sub_u32_e32 %7, %7, %0\n\t"
                    "v_bfi_b32 %2, %0, %3, %2\n\t"
                    "v_bfi_b32 %3, %0, 0, %3\n\t"
                    "v_bfi_b32 %4, %0, %5, %4\n\t"
                    "v_bfi_b32 %5, %0, %6, %5\n\t"
                    "v_bfi_b32 %6, %0, %8, %6"
                    : "=&v"(mr), "=&v"(t), "+v"(rhi), "+v"(rlo), "+v"(chi), "+v"(clo), "+v"(w0), "+v"(cnt) : "v"(w1));
            }
            // both children (row c0 = 2*ctx) are requested before this bit is known (not below the last level)
            u32 pl = 0, pr = 0;
            if (k < 7) { pl = trc_ldsr16(RCB_A(c0)); pr = trc_ldsr16(RCB_A(c0) + 128u); }
            const u32 slo = __builtin_amdgcn_alignbit(rhi, rlo, TRC_PROB_BITS), shi = rhi >> TRC_PROB_BITS;
            const u64 c64 = (u64)slo * p;
            const u32 cl = (u32)c64, hh = (u32)(c64 >> 32);          // cut = ch:cl, ch = shi * p + hh (inside the block: it fills a wait state)
            // The bit is the borrow of code - cut.  It stays on the vector side: m = 0 - 0 - borrow (all ones for a 1) leaves the
            // borrow in VCC, the adapted probability p - ((p >> 5) | m & (~0x7fff >> 5)) - bit takes it from there, the child is
            // c0 | (m & 128), the next row 2 * child.  Left to the compiler the borrow became an SGPR mask read by nine selects
            // (v_cndmask), each group behind the wait states a vector-written mask needs on gfx950.
            // The rest of the step sits in the same block: range - cut, the four bit-selects of the state.
            // Round 5, the wait states: a VALU read of VCC / an SGPR pair needs two instructions between it and the VALU write
            // (the compiler keeps an s_nop 1 between its own v_sub_co and v_subb; round 4 ran the chains back to back).  The two
            // subtractions run on two carry registers (code - cut: VCC, range - cut: an SGPR pair) and are interleaved with each
            // other, with the high product and with p >> 5, so that every such pair has its two states filled with the step's own
            // work: the same 15 + 3 instructions per bit as before (scripts/check_isa_hazards.py finds no site in the library).
            u32 m, t5, t6, r5, r6, ch, tt, np, an, cn;
            u64 sb;
            asm("v_sub_co_u32_e32 %[t5], vcc, %[clo], %[cl]\n\t"              // code - cut, low
                "v_mad_u32_u24 %[ch], %[shi], %[p], %[hh]\n\t"
                "v_sub_co_u32_e64 %[r5], %[sb], %[rlo], %[cl]\n\t"            // range - cut, low
                "v_subb_co_u32_e32 %[t6], vcc, %[chi], %[ch], vcc\n\t"
                "v_lshrrev_b32_e32 %[tt], 5, %[p]\n\t"
                "v_subb_co_u32_e64 %[r6], %[sb], %[rhi], %[ch], %[sb]\n\t"
                "v_subb_co_u32_e64 %[m], vcc, 0, 0, vcc\n\t"                  // m = -bit, VCC = bit
                "v_bfi_b32 %[clo], %[m], %[clo], %[t5]\n\t"                   // code = bit ? code : code - cut
                "v_bfi_b32 %[chi], %[m], %[chi], %[t6]\n\t"
                "v_and_or_b32 %[tt], %[m], %[k5], %[tt]\n\t"
                "v_subb_co_u32_e32 %[np], vcc, %[p], %[tt], vcc\n\t"          // p - ((p | m & ~0x7fff) >> 5) - bit
                "v_and_or_b32 %[an], %[m], %[k128], %[c0]\n\t"                // child
                "v_lshl_add_u32 %[cn], %[an], 1, %[negm]\n\t"                 // its children's row
                "v_bfi_b32 %[rlo], %[m], %[cl], %[r5]\n\t"                    // range = bit ? cut : range - cut
                "v_bfi_b32 %[rhi], %[m], %[ch], %[r6]"
                : [t5] "=&v"(t5), [t6] "=&v"(t6), [r5] "=&v"(r5), [r6] "=&v"(r6), [ch] "=&v"(ch), [tt] "=&v"(tt), [m] "=&v"(m), [np] "=&v"(np),
                  [an] "=&v"(an), [cn] "=&v"(cn), [sb] "=&s"(sb), [rlo] "+v"(rlo), [rhi] "+v"(rhi), [clo] "+v"(clo), [chi] "+v"(chi)
                : [cl] "v"(cl), [hh] "v"(hh), [shi] "v"(shi), [p] "v"(p), [k5] "s"(0xffff8000u >> 5), [k128] "s"(128u), [c0] "v"(c0), [negm] "v"(negm) : "vcc");
            trc_ldsw16(RCB_A(a), np);                          // rcb_adapt: p - (t5 + bit), 16 bits kept by the store
            a = an; c0 = cn;                                   // child 2*ctx + bit (bit 7 of c0 is clear) and its children's row
            p = rcb_bfi(m, pr, pl);
        }
        p1 = trc_ldsr16(mcol + 128u);                          // node 1 as the next byte will find it
        rpos += cnt << 2;                                      // the stream moves once per byte
        {
            const u32 m0 = (u32)__builtin_amdgcn_sbfe((int)cnt, 0, 1), m1 = (u32)__builtin_amdgcn_sbfe((int)cnt, 1, 1);
            w0 = rcb_bfi(m1, W.z, rcb_bfi(m0, W.y, W.x)); w1 = rcb_bfi(m1, W.w, rcb_bfi(m0, W.z, W.y));
        }
        if (many && __ballot(cnt > 2u)) {                      // rare: the next pair lies behind W
            const uint4 X = trc_ld16_a2(src + trc_min(rpos, lim));
            w0 = cnt > 2u ? X.x : w0; w1 = cnt > 2u ? X.y : w1;
        }
        return ((a + negm) >> 7) & 255u;
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
        uint4 pc0 = make_uint4(0, 0, 0, 0), pc1 = pc0, pc2 = pc0, pc3 = pc0;
#pragma nounroll
        for (u32 k = 0; k < 4; k++) {
            const u32 p0 = s * TRC_SEG + k * 16u;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (__ballot(coded && p0 < len)) {
#pragma nounroll
                for (u32 d = 0; d < 4; d++) {
                    u32 w = 0;
#pragma nounroll
                    for (u32 i = 0; i < 4; i++) w |= get_byte(true) << (8 * i);
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

bool trc_rcb_dec_prog_ok() { return true; }
bool trc_rcb_enc_gate_ok()
{
    const int env = getenv("TRC_RCB_L7G") ? atoi(getenv("TRC_RCB_L7G")) : -1, env_mc = getenv("TRC_RCB_MC") ? atoi(getenv("TRC_RCB_MC")) : -1;
    return env_mc >= 0 ? env_mc != 0 : (env < 0);              // the model wave + coder wave form (the default) is the one that waits at the gate
}
void trc_launch_rcb_enc(const uint8_t *d_in, size_t n, uint32_t chunk, const TrcWork &w, uint32_t *d_clen, hipStream_t s)
{
    // Level 7 in global memory buys residency (9 waves per CU instead of 5), not throughput: its two scattered 2-byte accesses
    // per byte and lane keep the texture-address unit as busy as the second wave per SIMD keeps the ALUs fed (100 MB, chunk 512:
    // 1.68 -> 1.86-2.0 ms).  It pays where it saves a residency ROUND: more waves than 5 per CU hold, no more than 9 per CU
    // hold (100 MB at chunk 1024: 1526 waves, 2.12 -> 1.30 ms).  TRC_RCB_L7G=0 / 1 force a form.
    static const int env = getenv("TRC_RCB_L7G") ? atoi(getenv("TRC_RCB_L7G")) : -1;
    static const int env_mc = getenv("TRC_RCB_MC") ? atoi(getenv("TRC_RCB_MC")) : -1;
    const bool l7g = env >= 0 ? env != 0 : (w.ngroups > 5u * 256u && w.ngroups <= 9u * 256u);
    // the two-wave form (model wave + coder wave): four workgroups per CU
    const bool mc = env_mc >= 0 ? env_mc != 0 : (env < 0);
    if (mc) {
        TRC_RAISE_LDS_ONCE(trc_rcb_enc_mc_kernel, TRC_WPG * RCB_MC_LDS);
        TRC_LAUNCH_TIMED(trc_rcb_enc_mc_kernel, TRC_QUAD_GRID(w.ngroups), dim3(128 * TRC_WPG), TRC_WPG * RCB_MC_LDS, s,
                           d_in, (u64)n, chunk, w.nchunks, w.scratch, w.stride, d_clen, w.gsum, trc_gate_tls.flag, trc_gate_tls.part);
        return;
    }
    if (l7g) {
        TRC_RAISE_LDS_ONCE(trc_rcb_enc_kernel<true>, TRC_WPG * RCB_ENC_WAVE_LDS(true));
        TRC_LAUNCH_TIMED(trc_rcb_enc_kernel<true>, TRC_QUAD_GRID(w.ngroups), dim3(64 * TRC_WPG), TRC_WPG * (RCB_ENC_WAVE_LDS(true)), s,
                           d_in, (u64)n, chunk, w.nchunks, w.scratch, w.stride, w.scratch2, d_clen, w.gsum);
    } else {
        TRC_RAISE_LDS_ONCE(trc_rcb_enc_kernel<false>, TRC_WPG * RCB_ENC_WAVE_LDS(false));
        TRC_LAUNCH_TIMED(trc_rcb_enc_kernel<false>, TRC_QUAD_GRID(w.ngroups), dim3(64 * TRC_WPG), TRC_WPG * (RCB_ENC_WAVE_LDS(false)), s,
                           d_in, (u64)n, chunk, w.nchunks, w.scratch, w.stride, w.scratch2, d_clen, w.gsum);
    }
}
void trc_launch_rcb_dec(const uint8_t *d_payload, const uint32_t *d_clen, size_t n, uint32_t chunk,
                        const TrcWork &w, uint8_t *d_out, hipStream_t s)
{
    TRC_RAISE_LDS_ONCE(trc_rcb_dec_kernel, TRC_WPG * RCB_WAVE_LDS);
    TRC_LAUNCH_TIMED(trc_rcb_dec_kernel, TRC_QUAD_GRID(w.ngroups), dim3(64 * TRC_WPG), TRC_WPG * (RCB_WAVE_LDS), s,
                       d_payload, d_clen, w.goff, w.gsum, (u64)n, chunk, w.nchunks, d_out, trc_prog_tls.counters, trc_prog_tls.host_flags, trc_prog_tls.part);
}
