// trc_ans_static.hip -- static-CDF rANS, 2 interleaved states per chunk (codec TRC_ANS4S).
//
// Per chunk the emitted bytes are exactly what anscdf4senc (reference anscdf.c:57-73, core
// anscdf_.h:46-48,90-94,101-103) returns for that slice:
//     [u32 state1][u32 state0][u16 renorm words in decode order]          (raw copy if >= len)
// The encoder walks its chunk BACKWARDS (n%4 tail bytes on state 0, then b3->s1 b2->s0 b1->s1
// b0->s0 per 4-byte group) and its words grow DOWNWARD from the end of the chunk's private scratch
// region; the decoder reads forward.  One lane = one chunk, 64 chunks per wave, so a wave64
// carries 128 reference rANS states; the serial dependency is broken across chunks, never
// inside one (that would change the bitstream).
//
// MI355X mapping: symbol tables in LDS (encoder: 256 x 16 B {reciprocal, 2^15-f | shift<<24,
// f<<16, c0}; decoder: 32 KiB slot->symbol LUT + 256 x 8 B {f, -c0}); all HBM traffic in 64-byte
// quad segments through the LDS tiles/rings of trc_io.h.  No MFMA: integer work, bounded by instruction
// issue (measured in round 5: 2.3 cycles per plain VOP2 wave64 operation, 4.2-4.6 for the three-operand / multiply / SDWA /
// DPP / compare kind these loops are made of, profiles/r05_valu_rates.txt) and LDS, not by HBM (DESIGN.md has the arithmetic).
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include "trc_io.h"
#include "trc_gather.h"
#include "trc_launch.h"

#define ENC_WAVE_LDS (TRC_SRING_BYTES)       // input arrives through an in-register quad transpose
#define ENC_PACE_LDS 192u                    // TrcPace's progress counters, behind the symbol table; then the fused gather's words:
#define ENC_FUSE_WSUM   64u                  //   u32[16]  the waves' group sums
#define ENC_FUSE_TICKET 128u                 //   u32      the workgroup's ticket
#define ENC_FUSE_BASE   136u                 //   u64      the workgroup's place in the payload
#ifndef TRC_ENC_FUSED
#define TRC_ENC_FUSED 0                      // 1: the encoder's waves gather their own payload when the launch is one residency round (trc_gather.h) -- measured a wash, off
#endif
#ifndef TRC_ENC_BALANCE
#define TRC_ENC_BALANCE 1                    // workgroups of twelve waves that keep each other's pace when the launch is one residency round
#endif
#define DEC_WAVE_LDS (TRC_SRING_BYTES)       // 8.3 KiB: 12 waves + 34 KiB of tables fit one CU
#ifndef TRC_DEC_LATE_FLUSH
#define TRC_DEC_LATE_FLUSH 1    // decoder: a segment's output stores behind the next period's commit (0: before it, as in round 2)
#endif
#ifndef TRC_ENC_EARLY_COMMIT
#define TRC_ENC_EARLY_COMMIT 1  // encoder: the next input segment lands before the current segment's last drain (0: after it, as in round 2)
#endif
#ifndef TRC_ENC_REP_DEFAULT
#define TRC_ENC_REP_DEFAULT 1
#endif

// ------------------------------------------------------------------------------------- encode ---
// LDS traffic of the symbol loop is written out by hand (inline asm, waits counted by hand: cdna_hip_programming.md 5.7):
//   * the symbol table is REPLICATED REP times inside LDS, entry x of replica r at byte x*16*REP + r*16, and lane l reads
//     replica l & (REP-1).  A ds_read_b128 is served in four groups of 16 lanes and a group finishes in one LDS cycle only
//     if its lanes touch 16 different 16-byte bank groups (MI355X_MICROARCH.md, LDS): with one copy of the table, 16 random
//     symbols take ~3 cycles (PMC round 1: 61 % of the LDS cycles were bank conflicts); with REP = 16 every lane of a
//     group owns its own bank group (the lane sets {0-3,12-15,20-27} / {4-11,16-19,28-31} have distinct l & 15) and the
//     read is conflict free whatever the symbols are; REP = 8 leaves two lanes per bank group.
//   * the compiler split part of these 16-byte reads into pairs of ds_read2_b32 (two LDS instructions, 32-bank rules);
//     the asm form is always one ds_read_b128, and the reads of the NEXT four symbols are in flight while the current
//     four are coded.
//   * renorm words go to the ring speculatively every symbol (the slot of the next unit is always free); the cursor
//     is kept as a halfword counter that a compare's carry decrements (v_subb), ring address = and + shift-add.
// one rANS step (ece, anscdf_.h:90-94): renorm-emit, then st = (st/f)<<15 + st%f + c0.
//   e = { m, (2^15-f) | sh<<24, f<<16, c0' }:  q = umulhi(st, m) >> sh == st / f  for st < 2^31
//   (f == 1 uses m = 2^32-1, sh = 0, c0' = c0 + 2^15-1: umulhi gives st-1, see trc_dir.hip)
//
// The renorm half of a step is one hand-written block (6 instructions, 5 VALU): compare -> VCC; ring address of the next
// unit from the halfword cursor (and, shift-add: also the two wait states a VALU write of VCC needs before a VALU reads
// it as a mask on gfx950); speculative 16-bit store; st = VCC ? st >> 16 : st as ONE v_cndmask with an SDWA source
// select (WORD_1 of st); cursor -= VCC (v_subbrev).  The division half (mul_hi, SDWA shift, mul24, add3) follows in the same
// block (left to the compiler it came with an s_nop per symbol behind the asm block: 1-1.5 % of the kernel).
#ifndef TRC_ENC_PRED_WRITE
#define TRC_ENC_PRED_WRITE 0          // 1: store only in lanes that emit (EXEC = VCC around the ds_write): ablation
#endif
__device__ __forceinline__ void ans_put(u32 &st, const trc_v4u e, u32 rbase, u32 &wn)
{
    u32 t;
#if TRC_ENC_PRED_WRITE
    u64 sv;
    asm volatile("v_cmp_ge_u32_e32 vcc, %0, %4\n\t"
                 "v_and_b32_e32 %2, 63, %1\n\t"
                 "v_lshl_add_u32 %2, %2, 1, %5\n\t"
                 "s_and_saveexec_b64 %3, vcc\n\t"
                 "ds_write_b16 %2, %0\n\t"
                 "s_mov_b64 exec, %3\n\t"
                 "v_cndmask_b32_sdwa %0, %0, %0, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
                 "v_subbrev_co_u32_e32 %1, vcc, 0, %1, vcc"
                 : "+v"(st), "+v"(wn), "=&v"(t), "=&s"(sv) : "v"(e.z), "v"(rbase) : "vcc", "memory");
#else
    // the whole step in one block, division included (mul_hi, SDWA shift by the entry's shift byte, mul24, add3): the compiler puts
    // a wait state behind an asm block whose result the next instruction reads (it cannot see which instruction wrote it),
    // which was an s_nop per symbol between the renorm block and the division -- 1-1.5 % of the kernel at every chunk size
    asm volatile("v_cmp_ge_u32_e32 vcc, %0, %3\n\t"
                 "v_and_b32_e32 %2, 63, %1\n\t"
                 "v_lshl_add_u32 %2, %2, 1, %4\n\t"
                 "ds_write_b16 %2, %0\n\t"
                 "v_cndmask_b32_sdwa %0, %0, %0, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
                 "v_subbrev_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
                 "v_mul_hi_u32 %2, %0, %5\n\t"
                 "v_lshrrev_b32_sdwa %2, %6, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\t"
                 "v_mul_u32_u24_e32 %2, %2, %6\n\t"
                 "v_add3_u32 %0, %0, %7, %2"
                 : "+v"(st), "+v"(wn), "=&v"(t) : "v"(e.z), "v"(rbase), "v"(e.x), "v"(e.y), "v"(e.w) : "vcc", "memory");
    return;
#endif
    const u32 q = __umulhi(st, e.x) >> (e.y >> 24);
    st = st + e.w + __umul24(q, e.y);                         // mul24 ignores the shift byte
}
// table address of byte k of w: (byte << SH) in one SDWA shift (+ the lane's replica offset when the table is replicated)
template <int K>
__device__ __forceinline__ u32 ans_taddr(u32 w, u32 sh, u32 tbase, bool replicated)
{
    u32 a;
    if (K == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(a) : "v"(sh), "v"(w));
    if (K == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(a) : "v"(sh), "v"(w));
    if (K == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(a) : "v"(sh), "v"(w));
    if (K == 3) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(a) : "v"(sh), "v"(w));
    return replicated ? a + tbase : a;
}
// the four table entries of one input dword, requested together
struct EncQuad { trc_v4u e0, e1, e2, e3; };
__device__ __forceinline__ void ans_fetch4(EncQuad &q, u32 w, u32 tbase, int shift)
{
    const u32 sh = (u32)shift;
    q.e3 = trc_lds_read128(ans_taddr<3>(w, sh, tbase, shift != 4));
    q.e2 = trc_lds_read128(ans_taddr<2>(w, sh, tbase, shift != 4));
    q.e1 = trc_lds_read128(ans_taddr<1>(w, sh, tbase, shift != 4));
    q.e0 = trc_lds_read128(ans_taddr<0>(w, sh, tbase, shift != 4));
}
// all of q's registers become valid here: at most `pending` LDS operations were issued after its four reads
#define ANS_WAIT4(q, pending)                                                                                        \
    asm volatile("s_waitcnt lgkmcnt(" #pending ")" : "+v"(q.e0), "+v"(q.e1), "+v"(q.e2), "+v"(q.e3) :: "memory")

#ifdef TRC_ENC_PROF                                              // variant builds only: wall clock (100 MHz) per wave: start, first symbol, last symbol, end
__device__ unsigned long long trc_enc_wall[4 * 4096];
#endif
template <int BLOCK, int REP, bool FUSED>
__global__ __launch_bounds__(BLOCK) void trc_ans4s_enc_kernel(
    const u8 *__restrict__ in, u64 n, u32 chunk, u32 nchunks,
    const uint4 *__restrict__ etab_g, u8 *__restrict__ scratch, u32 stride,
    u32 *__restrict__ clen, u32 *__restrict__ gsum,
    u8 *__restrict__ payload, u64 *__restrict__ total, u64 *__restrict__ goff_out, u8 *__restrict__ sync)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    constexpr u32 TAB = 4096u * REP;
    constexpr int SH = REP == 16 ? 8 : REP == 8 ? 7 : REP == 4 ? 6 : 4;
    static_assert(REP == 1 || REP == 8 || REP == 16, "replica count");
    const u32 tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
#ifdef TRC_ENC_PROF
    const u64 ew0 = wall_clock64();
#endif
    u8 *wbase = smem + TAB + ENC_PACE_LDS + wv * ENC_WAVE_LDS;
    u32 *const fuse_wsum = (u32 *)(smem + TAB + ENC_FUSE_WSUM);
    if (FUSED) {                                                // the workgroup's place in the container: a ticket, not blockIdx (trc_gather.h)
        if (tid == 0) *(u32 *)(smem + TAB + ENC_FUSE_TICKET) = __hip_atomic_fetch_add((u32 *)(sync + TRC_SYNC_TICKET), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < 16u) fuse_wsum[tid] = 0u;
    }
    for (u32 i = tid; i < 256u * REP; i += BLOCK) ((uint4 *)smem)[i] = etab_g[i / REP];
    TrcPace pace; pace.init(trc_lds_addr(smem) + TAB, tid, wv);
    __syncthreads();
    const u32 wg = FUSED ? (u32)__builtin_amdgcn_readfirstlane((int)*(const u32 *)(smem + TAB + ENC_FUSE_TICKET)) : blockIdx.x;

    WaveChunks wc;
    wc.c0 = (wg * (BLOCK / 64) + wv) * 64u; wc.chunk = chunk; wc.nchunks = nchunks;
    wc.lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    if (wc.c0 >= nchunks) return;                               // (a barrier waits for the surviving waves only; wave 0 of a workgroup always has chunks)
    wc.rows = nchunks - wc.c0 < 64u ? nchunks - wc.c0 : 64u;

    const bool alive = lane < wc.rows;
    const u32 c = wc.c0 + lane;
    const u32 len = alive ? wc.len_of(lane) : 0u;
    QuadIn tin; tin.base = in + (u64)wc.c0 * chunk;
    StreamOut<true, false, false, true> so;                   // (write-through drains: trc_io.h)
    so.rings = wbase;
    so.scratch = scratch; so.stride = stride; so.c0 = wc.c0; so.wpos = 0; so.nfl = 0;
    const u32 tbase = (lane & (u32)(REP - 1)) << 4;                              // this lane's replica (table at LDS offset 0)
    const u32 rbase = (u32)(uintptr_t)(so.rings - smem) + trc_raddr(lane, 0);    // this lane's ring, as an LDS byte address
    const uint4 *etab1 = (const uint4 *)smem;                                    // generic view for the ragged tail: replica 0

    const u32 S = chunk / TRC_SEG;
    const u32 top = alive ? (len - 1u) / TRC_SEG : 0u;          // segment holding the chunk's last byte
    const u32 toplen = len - TRC_SEG * top;                     // bytes of the chunk in that segment
    u32 st0 = TRC_ANS_LOW, st1 = TRC_ANS_LOW;
    bool ovf = false;

    // both halves of a 128-byte line are requested together (QuadIn, paired mode)
    if ((S - 1u) & 1u) { tin.issue_slot<1>(wc, S - 1u); tin.issue_slot<0>(wc, S - 2u); } else tin.issue_slot<0>(wc, S - 1u);
    // land segment sn (its registers become `p`) and request the line below it.  The compiler guards the landing with
    // s_waitcnt vmcnt(0), and on gfx950 stores count in vmcnt: done at the top of a segment (round 2) it sat out the latency
    // of the drain stores issued just before; it now runs BEFORE the last drain of the segment above (TRC_ENC_EARLY_COMMIT).
    auto take = [&](u32 sn) {
        if (sn & 1u) tin.commit_slot<1>(); else tin.commit_slot<0>();
        if (!(sn & 1u) && sn >= 1u) {                           // the next line down: in flight during this segment (and the next)
            tin.issue_slot<1>(wc, sn - 1u);
            if (sn >= 2u) tin.issue_slot<0>(wc, sn - 2u);
        }
    };
    take(S - 1u);
#ifdef TRC_ENC_PROF
    const u64 ew1 = wall_clock64();
#endif
    for (u32 s = S - 1u;; s--) {
        if (BLOCK > 256) pace.step(S - s);                     // (workgroups of more than four waves: some share a SIMD)
#if !TRC_ENC_EARLY_COMMIT
        if (s != S - 1u) take(s);
#endif
        bool act = alive && s <= top && !ovf;
        const bool ragged = act && s == top && toplen != TRC_SEG;
        if (ragged) {                                           // last chunk only: byte by byte
            const u32 body = len & ~3u;
            const u8 *mine = in + (u64)c * chunk;               // (one lane in the whole grid: plain byte loads)
            for (u32 pos = len; pos > TRC_SEG * top;) {
                pos--;
                const uint4 e = etab1[(u32)mine[pos] * REP];
                const trc_v4u ev = { e.x, e.y, e.z, e.w };
                u32 wn = ~(so.wpos >> 1);
                if (pos >= body || !(pos & 1u)) ans_put(st0, ev, rbase, wn); else ans_put(st1, ev, rbase, wn);
                so.wpos = (~wn) << 1;                           // (the top segment is a lane's first: <= 126 bytes into an empty ring)
            }
            act = false;
        }
#pragma unroll
        for (int k = 3; k >= 0; k--) {
            if (act) {
                const uint4 v = tin.read((u32)k);
                u32 wn = ~(so.wpos >> 1);                       // halfword cursor: next unit -> ring halfword wn & 63
                EncQuad a, b;
                // four dwords, top first; the reads of dword d-1 fly while dword d is coded.  LDS operations issued
                // after a's reads when a is waited for: 4 (b's reads) [+ 4 ring writes of the dword before]
                ans_fetch4(a, v.w, tbase, SH);
                ans_fetch4(b, v.z, tbase, SH);
                ANS_WAIT4(a, 4);
                ans_put(st1, a.e3, rbase, wn); ans_put(st0, a.e2, rbase, wn); ans_put(st1, a.e1, rbase, wn); ans_put(st0, a.e0, rbase, wn);
                ans_fetch4(a, v.y, tbase, SH);
                ANS_WAIT4(b, 8);
                ans_put(st1, b.e3, rbase, wn); ans_put(st0, b.e2, rbase, wn); ans_put(st1, b.e1, rbase, wn); ans_put(st0, b.e0, rbase, wn);
                ans_fetch4(b, v.x, tbase, SH);
                ANS_WAIT4(a, 8);
                ans_put(st1, a.e3, rbase, wn); ans_put(st0, a.e2, rbase, wn); ans_put(st1, a.e1, rbase, wn); ans_put(st0, a.e0, rbase, wn);
                ANS_WAIT4(b, 4);
                ans_put(st1, b.e3, rbase, wn); ans_put(st0, b.e2, rbase, wn); ans_put(st1, b.e1, rbase, wn); ans_put(st0, b.e0, rbase, wn);
                so.wpos = (~wn) << 1;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the ring writes above are not in the compiler's books
#if TRC_ENC_EARLY_COMMIT
            if (k == 0 && s > 0) take(s - 1u);                  // (this segment's last piece has been read)
#endif
            so.drain(false, alive);                             // <= 32 new bytes per lane since the last drain
            ovf = ovf || (alive && so.wpos + 8u >= len);        // already incompressible: stop coding this chunk
            act = act && !ovf;
        }
        if (s == 0) break;
    }
#ifdef TRC_ENC_PROF
    const u64 ew2 = wall_clock64();
#endif
    u32 out_len = 0;
    if (alive) {
        if (!ovf) {
            // ansflush: state0 then state1 as u32 below the words (hi half first going down)
            so.put16(st0 >> 16); so.put16(st0); so.put16(st1 >> 16); so.put16(st1);
            ovf = so.wpos >= len;
        }
        out_len = ovf ? len : so.wpos;
    }
    so.drain(true, alive && !ovf);
    if (alive) clen[c] = out_len;
    const u32 gs = trc_wave_sum(out_len);
    if (lane == 0) gsum[wc.c0 >> 6] = gs;
    if (FUSED) {
        // the gather, by the waves that wrote the bytes (trc_gather.h)
        __builtin_amdgcn_s_setprio(0);
        if (lane == 0) fuse_wsum[wv] = gs;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's drains (asm stores: not in the compiler's books) have left for the L2
        trc_lds_barrier();
        u64 *const pub = (u64 *)(sync + TRC_SYNC_PUB);
        const u32 nwg = ((nchunks + 63u) / 64u + (u32)(BLOCK / 64) - 1u) / (u32)(BLOCK / 64);
        if (wv == 0) {
            u32 wsum_all = 0;
#pragma unroll
            for (u32 k = 0; k < (u32)(BLOCK / 64); k++) wsum_all += fuse_wsum[k];
            if (lane == 0) trc_sync_store(pub + wg, TRC_SYNC_VALID | wsum_all);
            const u64 wgbase = trc_sync_prefix(pub, wg);
            if (lane == 0) {
                *(u64 *)(smem + TAB + ENC_FUSE_BASE) = wgbase;
                if (wg == nwg - 1u && total) *total = wgbase + wsum_all;
                // the last workgroup to have finished polling leaves the area as it found it: zero
                const u32 done = __hip_atomic_fetch_add((u32 *)(sync + TRC_SYNC_DONE), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (done == nwg - 1u) {
                    for (u32 j = 0; j < nwg; j++) trc_sync_store(pub + j, 0ull);
                    __hip_atomic_store((u32 *)(sync + TRC_SYNC_TICKET), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store((u32 *)(sync + TRC_SYNC_DONE), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        trc_lds_barrier();
        u64 base = *(const u64 *)(smem + TAB + ENC_FUSE_BASE);
        for (u32 k = 0; k < wv; k++) base += fuse_wsum[k];
        if (lane == 0 && goff_out) goff_out[wc.c0 >> 6] = base;  // (kept for a decode of this directory: TrcWork::goff_area)
        const bool raw = out_len == len;
        const u8 *src = raw ? in + (u64)c * chunk : scratch + (u64)(c + 1u) * stride - out_len;
        trc_wave_gather64(payload + base, (u32 *)wbase, (u64 *)(wbase + 512), alive ? out_len : 0u, alive ? src : scratch);
    }
#ifdef TRC_ENC_PROF
    if (lane == 0) { const u32 wid = (wc.c0 >> 6) & 4095u; trc_enc_wall[4 * wid] = ew0; trc_enc_wall[4 * wid + 1] = ew1; trc_enc_wall[4 * wid + 2] = ew2; trc_enc_wall[4 * wid + 3] = wall_clock64(); }
#endif
}

// ------------------------------------------------------------------------------------- decode ---
// cdf16sansdec + ecdnorm (cdf_.h:99-107, anscdf_.h:51-73) for a byte alphabet:
//     x = lut[slot = st & 0x7fff];  st = f * (st >> 15) + slot - c0;  if (st < 2^15) st = st << 16 | next word
// with a 32 KiB slot -> symbol LUT and a 256-entry {f, -c0} table in LDS.
//
// Round 3 (profiles/r03_notes.md has the counters of every step):
//   * symbols are decoded in PAIRS (one per state) on integer LDS addresses: the LUT sits at LDS offset 0, so its read needs no
//     address arithmetic (through generic pointers the compiler added the zero segment base with a v_add per read); the table
//     read folds its base into the offset field;
//   * the two ring units a pair can consume are requested up front from the INTERLEAVED ring (StreamInT<true>: dword d of
//     every lane in row d, conflict-free whatever the cursors are; row 32 mirrors row 0): the dword behind the cursor and the
//     next one from one address, funnel-shifted by the cursor's parity -- the second symbol's word no longer waits for the
//     first symbol's renormalisation compare, cursor arithmetic and LDS round trip;
//   * the renormalisation of a pair is one hand-scheduled block of 9 VALU (the compiler's form of the same selects: 14):
//     compares into VCC / an SGPR pair, candidates `state << 16 | unit` as byte permutes of the state and the 32-bit window,
//     selects, the halfword cursor advanced by the carries (v_addc); the two symbols' instructions interleaved so that the two
//     wait states a VALU-written mask needs on gfx950 are filled with work;
//   * the output segment of 64 symbols leaves (QuadOut::flush) AFTER the next period's commit instead of before it: the
//     compiler guards the commit's use of the refill registers with s_waitcnt vmcnt(0), and on gfx950 stores count in vmcnt --
//     every segment the wave sat out the full latency of the stores it had just issued.
// A variant with ONE table read per symbol (a 4096-entry table over buckets of 8 slots holding both symbols a bucket can
// straddle, flagged buckets taking the two-read route) is bit-exact and 43 % SLOWER (96 against 67 us: 16 more VALU per pair
// make the kernel issue-bound): commit 341b467, numbers in profiles/r03_notes.md.
typedef u32 trc_v2u __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) trc_v2u trc_lds_u64;
typedef __attribute__((address_space(3))) u32 trc_lds_u32;
typedef StreamInT<true> AnsStreamIn;
#define DEC_LDS_LUT   0u          // u8[32768]
#define DEC_LDS_DTAB  32768u      // uint2[256]  { f, -c0 }
#ifndef TRC_DEC_BALANCE
#define TRC_DEC_BALANCE 2       // the waves of a SIMD keep each other's pace (TrcPace, trc_dev.h); 0: off; 1 / 2 / 3: every period / segment / second period
#endif
#define DEC_LDS_PROG  34816u      // u32[16]     progress counters, [SIMD][age]
#define DEC_LDS_WAVES 34880u

// one symbol, any position (the ragged tail of the last chunk)
__device__ __forceinline__ u32 ans_get(u32 &st, AnsStreamIn &si)
{
    const u32 slot = st & (TRC_PROB_ONE - 1);
    const u32 x = *(const trc_lds_u8 *)(uintptr_t)(DEC_LDS_LUT + slot);
    const trc_v2u e = *(const trc_lds_u64 *)(uintptr_t)(DEC_LDS_DTAB + (x << 3));
    st = __umul24(e.x, st >> TRC_PROB_BITS) + e.y + slot;
    const u32 w = si.peek16();
    const bool rn = st < TRC_ANS_LOW;
    st = rn ? (st << 16) | w : st;
    si.rpos += rn ? 2u : 0u;
    return x;
}
// a pair of symbols, s0 first; hc = halfword cursor into the lane's ring (lbase = LDS address of its dword 0).
// sl0 / sl1 travel with the states: the slots (state & 0x7fff) of the NEXT pair are the last two instructions of the
// renormalisation block -- left to the compiler they followed the block, behind the wait state it puts after an asm
// statement whose outputs the next VALU instruction reads (an s_nop per pair).
__device__ __forceinline__ void ans_get_pair(u32 &s0, u32 &s1, u32 &sl0, u32 &sl1, u32 lbase, u32 &hc, u32 &x0, u32 &x1, u32 sel_lo, u32 sel_hi)
{
    u32 a;
    asm("v_bfe_u32 %0, %1, 1, 5\n\tv_lshl_add_u32 %0, %0, 8, %2" : "=&v"(a) : "v"(hc), "v"(lbase));
    const u32 dw0 = *(const trc_lds_u32 *)(uintptr_t)a;
    const u32 dw1 = *(const trc_lds_u32 *)(uintptr_t)(a + 256u);
    x0 = *(const trc_lds_u8 *)(uintptr_t)(DEC_LDS_LUT + sl0);
    x1 = *(const trc_lds_u8 *)(uintptr_t)(DEC_LDS_LUT + sl1);
    const trc_v2u e0 = *(const trc_lds_u64 *)(uintptr_t)(DEC_LDS_DTAB + (x0 << 3));
    const trc_v2u e1 = *(const trc_lds_u64 *)(uintptr_t)(DEC_LDS_DTAB + (x1 << 3));
    u32 t0 = __umul24(e0.x, s0 >> TRC_PROB_BITS) + e0.y + sl0;
    u32 t1 = __umul24(e1.x, s1 >> TRC_PROB_BITS) + e1.y + sl1;
#ifndef TRC_DEC_ALIGNBYTE
#define TRC_DEC_ALIGNBYTE 0                                     // measured: 60.7-61.0 us either way (profiles/r05zi_alignbyte.txt)
#endif
#if TRC_DEC_ALIGNBYTE                                            // v_alignbyte shifts by 8 x (operand & 3): 2 hc is a plain v_add (2.3 cycles per SIMD), hc << 4 a v_lshlrev (4.2-5)
    const u32 w32 = __builtin_amdgcn_alignbyte(dw1, dw0, hc + hc);                       // units hc, hc + 1
#else
    const u32 w32 = __builtin_amdgcn_alignbit(dw1, dw0, hc << 4);                        // units hc, hc + 1 (the shift uses 5 bits: 16 x parity)
#endif
    u32 c0, c1, sl;
    u64 m1, cy;
    asm("v_cmp_gt_u32_e32 vcc, 0x8000, %[t0]\n\t"
        "v_perm_b32 %[c0], %[t0], %[w], %[slo]\n\t"
        "v_cmp_gt_u32_e64 %[m1], %[low], %[t1]\n\t"
        "v_cndmask_b32_e32 %[t0], %[t0], %[c0], vcc\n\t"
        "v_cndmask_b32_e32 %[sl], %[slo], %[shi], vcc\n\t"
        "v_addc_co_u32_e64 %[hc], %[cy], 0, %[hc], vcc\n\t"
        "v_perm_b32 %[c1], %[t1], %[w], %[sl]\n\t"
        "v_cndmask_b32_e64 %[t1], %[t1], %[c1], %[m1]\n\t"
        "v_addc_co_u32_e64 %[hc], %[cy], 0, %[hc], %[m1]\n\t"
        "v_and_b32_e32 %[n0], 0x7fff, %[t0]\n\t"
        "v_and_b32_e32 %[n1], 0x7fff, %[t1]"
        : [t0] "+v"(t0), [t1] "+v"(t1), [hc] "+v"(hc), [c0] "=&v"(c0), [c1] "=&v"(c1), [sl] "=&v"(sl), [m1] "=&s"(m1), [cy] "=&s"(cy),
          [n0] "=&v"(sl0), [n1] "=&v"(sl1)
        : [w] "v"(w32), [low] "s"(TRC_ANS_LOW), [slo] "v"(sel_lo), [shi] "v"(sel_hi) : "vcc");
    s0 = t0; s1 = t1;
}

#ifdef TRC_DEC_PROF                                              // variant builds only (scripts/build_variant.sh prof -DTRC_DEC_PROF): where a decoder wave's cycles go
__device__ unsigned long long trc_dec_prof[8];                  // fill, directory + prime, periods, flushes, symbols, total, waves
__device__ unsigned long long trc_dec_wall[2 * 4096];            // wall clock (100 MHz) per wave: start, end (plain stores: same-address atomics from 3052 waves cost 500 us)
#ifdef TRC_DEC_PROF_LIGHT                                        // wall-clock span of the launch only: no clock reads inside the loop (each one drains the LDS queue)
#define PROF_T(x) const u64 x = 0
#define PROF_ACC(acc, a, b)
#else
#define PROF_T(x) const u64 x = clock64()
#define PROF_ACC(acc, a, b) acc += (b) - (a)
#endif
#else
#define PROF_T(x)
#define PROF_ACC(acc, a, b)
#endif
__global__ __launch_bounds__(896) void trc_ans4s_dec_kernel(
    const u8 *__restrict__ payload, const u32 *__restrict__ clen, const u64 *__restrict__ goff, const u32 *__restrict__ gsum,
    u64 n, u32 chunk, u32 nchunks,
    const u8 *__restrict__ lut_g, const u32 *__restrict__ dtab_g, u8 *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    if (trc_lds_addr(smem) != 0u) __builtin_trap();            // the decoders address LUT / dtab / rings by ABSOLUTE LDS offsets (DEC_LDS_*): a static __shared__ object in front of the dynamic segment must fail loudly, not decode from the wrong tables
    const u32 tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6, BLOCK = blockDim.x;
    u8 *wbase = smem + DEC_LDS_WAVES + wv * DEC_WAVE_LDS;
    PROF_T(pt0);
#ifdef TRC_DEC_PROF
    u64 acc_p = 0, acc_f = 0, acc_s = 0;
    const u64 wall0 = wall_clock64();
#endif
    WaveChunks wc;
    wc.c0 = (blockIdx.x * (BLOCK / 64) + wv) * 64u; wc.chunk = chunk; wc.nchunks = nchunks;
    wc.lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    const bool valid = wc.c0 < nchunks;
    wc.rows = !valid ? 0u : nchunks - wc.c0 < 64u ? nchunks - wc.c0 : 64u;
    const bool alive = lane < wc.rows;
    const u32 c = wc.c0 + lane;
    const u32 len = alive ? wc.len_of(lane) : 0u;
    // Round 5, start-up order.  In-kernel clocks put 8 us of the 66 between the table fill and the first symbol: a chain of
    // dependent round trips (tables -> barrier -> clen -> group base -> states and first ring fill), the last of them a burst of
    // 128 bytes per stream from every wave of the launch at once.  Now: the directory entry and the group base are requested
    // FIRST, together with the tables (nothing of them depends on the tables); the states and the ring fill are requested before
    // the barrier, which waits for LDS only (__syncthreads() would sit out every load in flight).  (The rings' second halves landing
    // one period later -- StreamInT::prime_land(P, 1) behind the first 16 symbols -- was built too: no change, and it does not go
    // with the aligned segments below, whose first half may hold as little as two bytes of the stream.)
    const u32 cl_raw = alive ? clen[c] : 0u;
    const u64 gbase0 = valid ? trc_group_base(goff, gsum, wc.c0 >> 6) : 0ull;
    // the table fill as one batch of loads (a plain copy loop waits for each of its three loads before it issues the next:
    // 75.9 -> 73.6 us for 100 MB in round 2)
    {
        uint2 *dtab = (uint2 *)(smem + DEC_LDS_DTAB);
        if (BLOCK >= 704u) {
            uint4 t0 = make_uint4(0, 0, 0, 0), t1 = t0, t2 = t0; u32 td = 0;
            t0 = ((const uint4 *)lut_g)[tid];
            t1 = ((const uint4 *)lut_g)[tid + BLOCK];
            if (tid + 2u * BLOCK < 2048u) t2 = ((const uint4 *)lut_g)[tid + 2u * BLOCK];
            if (tid < 256u) td = dtab_g[tid];
            ((uint4 *)smem)[tid] = t0; ((uint4 *)smem)[tid + BLOCK] = t1;
            if (tid + 2u * BLOCK < 2048u) ((uint4 *)smem)[tid + 2u * BLOCK] = t2;
            if (tid < 256u) dtab[tid] = make_uint2(td >> 16, 0u - (td & 0xffffu));
        } else {
            for (u32 i = tid; i < 2048; i += BLOCK) ((uint4 *)smem)[i] = ((const uint4 *)lut_g)[i];
            for (u32 i = tid; i < 256; i += BLOCK) { const u32 d = dtab_g[i]; dtab[i] = make_uint2(d >> 16, 0u - (d & 0xffffu)); }
        }
    }
    PROF_T(pt1);
#if TRC_DEC_BALANCE
    TrcPace pace; pace.init(DEC_LDS_PROG, tid, wv);
#endif

    const u32 cl = alive ? trc_min(cl_raw, len) : 0u;         // a directory entry above the chunk length (corrupt input) reads as raw
    const u32 ex = trc_wave_incl_scan(cl) - cl;
    const u64 off = gbase0 + ex;
    const bool coded = alive && cl != len;

    QuadOut tout; tout.base = out + (u64)wc.c0 * chunk;        // output leaves through an in-register quad transpose
    AnsStreamIn si;
    si.rings = wbase;
    // Round 5: the stream is read in segments aligned to 64 bytes of the PAYLOAD, not of the stream: the ring's "stream" starts at the
    // 64-byte boundary below the first word (r0 bytes of it are somebody else's and are skipped by starting the cursor at r0), so
    // every 64-byte refill request is one 64-byte sector of one line.  Stream-relative segments (rounds 1-4) straddled two
    // sectors at 2-byte alignment: the payload was fetched 2.7 times (profiles/r04_pmc_traffic.txt), the first ring fill of a launch
    // -- 128 bytes per stream from every wave at once -- took three sectors per stream where two do.
    si.gbase = payload; si.soff = off + 8; si.lim = trc_sub_sat(cl, 8u);   // words follow the two states
    const u32 r0 = si.align_start(coded);
    u32 sa = 0, sb = 0;
    if (coded) { sa = trc_ld32_a2(payload + off); sb = trc_ld32_a2(payload + off + 4); }   // sa = enc state 1, sb = enc state 0
    AnsStreamIn::Prime P;
    si.prime_issue(coded, P);
    trc_lds_barrier();                                         // the tables are in place (LDS only: the loads above stay in flight)
    if (!valid) return;
    si.prime_land(P, 0, coded);
    si.prime_land(P, 1, coded);                                // (both halves before the first period: of the first one up to 62 bytes are skipped)
    si.rpos = r0;

    const u32 S = chunk / TRC_SEG;
    const u32 body4 = len & ~3u;
    u8 *dst = out + (u64)c * chunk;
    const u32 rbase = (u32)(uintptr_t)(si.rings - smem) + AnsStreamIn::ra(lane, 0);    // this lane's ring, as an LDS byte address
    u32 sel_lo = 0x05040100u, sel_hi = 0x05040302u;            // byte selectors of the pair step's permutes (VGPR operands: VCC takes the one constant-bus slot)
    asm volatile("" : "+v"(sel_lo), "+v"(sel_hi));
    PROF_T(pt2);
    for (u32 s = 0; s < S; s++) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u32 p0 = s * TRC_SEG + (u32)k * 16u;          // chunk offset of this 16-byte piece
#if TRC_DEC_BALANCE
            // (policies measured, profiles/r05_notes.md: every period 64 us, every segment 62-63, every second period 62-63, graded
            // priorities 63, laggards 3 / leader 1: 63; off: 67)
            if (TRC_DEC_BALANCE == 1 || k == 0 || (TRC_DEC_BALANCE == 3 && k == 2)) pace.step(s * 4u + (u32)k + 1u);
#endif
            // period boundary: land the round requested 16 symbols ago, request the next one
            PROF_T(qa);
            si.period(coded && p0 < len, k & 1);
            PROF_T(qb); PROF_ACC(acc_p, qa, qb);
#if TRC_DEC_LATE_FLUSH
            if (k == 0 && s > 0) tout.flush(wc, (s - 1u) * TRC_SEG);   // the segment before: behind this period's commit (header comment)
#endif
            PROF_T(qc); PROF_ACC(acc_f, qb, qc);
            if (coded && p0 + 16u <= len) {
                u32 w[4];
                u32 hc = si.rpos >> 1;
                u32 slb = sb & (TRC_PROB_ONE - 1), sla = sa & (TRC_PROB_ONE - 1);
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    u32 x0, x1, x2, x3;
                    ans_get_pair(sb, sa, slb, sla, rbase, hc, x0, x1, sel_lo, sel_hi);
                    ans_get_pair(sb, sa, slb, sla, rbase, hc, x2, x3, sel_lo, sel_hi);
                    w[d] = (((((x3 << 8) | x2) << 8) | x1) << 8) | x0;
                }
                si.rpos = hc << 1;
                tout.put((u32)k, make_uint4(w[0], w[1], w[2], w[3]));
                PROF_T(qd); PROF_ACC(acc_s, qc, qd);
            } else if (coded && p0 < len) {                    // last chunk's final partial piece
                for (u32 pos = p0; pos < len; pos++)
                    dst[pos] = (u8)((pos >= body4 || !(pos & 1u)) ? ans_get(sb, si) : ans_get(sa, si));
            }
        }
#if !TRC_DEC_LATE_FLUSH
        tout.flush(wc, s * TRC_SEG);
#endif
    }
#if TRC_DEC_LATE_FLUSH
    tout.flush(wc, (S - 1u) * TRC_SEG);
#endif
    // chunks stored raw (clen == len): the whole wave copies them, one after the other
    trc_wave_copy_raw(__ballot(alive && cl == len && len != 0), off, len, out + (u64)wc.c0 * chunk, chunk, payload);
#ifdef TRC_DEC_PROF
    if (lane == 0) {
#ifdef TRC_DEC_PROF_LIGHT
        const u64 pt3 = 0;
#else
        const u64 pt3 = clock64();
#endif
#ifndef TRC_DEC_PROF_LIGHT
        atomicAdd(&trc_dec_prof[0], pt1 - pt0); atomicAdd(&trc_dec_prof[1], pt2 - pt1); atomicAdd(&trc_dec_prof[2], acc_p);
        atomicAdd(&trc_dec_prof[3], acc_f); atomicAdd(&trc_dec_prof[4], acc_s); atomicAdd(&trc_dec_prof[5], pt3 - pt0); atomicAdd(&trc_dec_prof[6], 1ull);
#endif
        const u32 wid = (blockIdx.x * (BLOCK / 64) + wv) & 4095u;
        trc_dec_wall[2 * wid] = wall0; trc_dec_wall[2 * wid + 1] = wall_clock64();
    }
#endif
}

// ---------------------------------------------------------------- decode, two LANES per chunk ---
// Round 3, for calls that cannot fill the chip with one lane per chunk (fewer than 2048 waves: 100 MB at chunk 1024 and up).
// A wave's time is its own instruction count, and with one lane per chunk a lane runs BOTH states of its chunk.  Here lanes 2i
// and 2i + 1 take chunk i of the wave, one state each (even lane: the state of the even-index symbols): twice the waves, each
// with little more than half the instructions per symbol pair.  The two states share the chunk's word stream -- a unit goes to
// whichever symbol renormalises first, in symbol order -- so a step exchanges the renormalisation flags inside the pair (DPP) and
// both lanes advance one shared cursor: the even lane takes the unit behind the cursor, the odd lane that one or the next,
// depending on the even lane's flag.  The stream's ring belongs to the even lane (the odd lane reads its partner's column and never
// asks for a refill); the two lanes' bytes are merged by DPP into identical 16-byte pieces before the output transpose (pair
// mode of QuadOut, as in trc_rcs2p_dec_kernel).
__device__ __forceinline__ u32 ans_get_half(u32 &s, u32 lbase, u32 &hc, u32 b, u32 seld)
{
    u32 a;
    asm("v_bfe_u32 %0, %1, 1, 5\n\tv_lshl_add_u32 %0, %0, 8, %2" : "=&v"(a) : "v"(hc), "v"(lbase));
    const u32 dw0 = *(const trc_lds_u32 *)(uintptr_t)a;
    const u32 dw1 = *(const trc_lds_u32 *)(uintptr_t)(a + 256u);
    const u32 sl = s & (TRC_PROB_ONE - 1);
    const u32 x = *(const trc_lds_u8 *)(uintptr_t)(DEC_LDS_LUT + sl);
    const trc_v2u e = *(const trc_lds_u64 *)(uintptr_t)(DEC_LDS_DTAB + (x << 3));
    const u32 t = __umul24(e.x, s >> TRC_PROB_BITS) + e.y + sl;
    const u32 w32 = __builtin_amdgcn_alignbit(dw1, dw0, hc << 4);        // units hc, hc + 1
    const u32 rn = t < TRC_ANS_LOW ? 1u : 0u;
    const u32 oth = trc_quad_xor1(rn);                                   // the partner's flag
    // even lane (b = 0): the unit behind the cursor; odd lane: that one, or the next if the even lane took it.  The byte
    // selector of `state << 16 | unit` is 0x05040100 for the window's low unit, + 0x0202 for its high unit (seld = 0x0202 * b)
    const u32 sel = 0x05040100u + oth * seld;
    const u32 c = __builtin_amdgcn_perm(t, w32, sel);
    s = rn ? c : t;
    hc += rn + oth;
    return x;
}

__global__ __launch_bounds__(896) void trc_ans4s_dec2_kernel(
    const u8 *__restrict__ payload, const u32 *__restrict__ clen, const u64 *__restrict__ goff, const u32 *__restrict__ gsum,
    u64 n, u32 chunk, u32 nchunks,
    const u8 *__restrict__ lut_g, const u32 *__restrict__ dtab_g, u8 *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    if (trc_lds_addr(smem) != 0u) __builtin_trap();            // the decoders address LUT / dtab / rings by ABSOLUTE LDS offsets (DEC_LDS_*): a static __shared__ object in front of the dynamic segment must fail loudly, not decode from the wrong tables
    const u32 tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6, BLOCK = blockDim.x;
    u8 *wbase = smem + DEC_LDS_WAVES + wv * DEC_WAVE_LDS;
    {
        uint2 *dtab = (uint2 *)(smem + DEC_LDS_DTAB);
        if (BLOCK >= 704u) {
            uint4 t0 = make_uint4(0, 0, 0, 0), t1 = t0, t2 = t0; u32 td = 0;
            t0 = ((const uint4 *)lut_g)[tid];
            t1 = ((const uint4 *)lut_g)[tid + BLOCK];
            if (tid + 2u * BLOCK < 2048u) t2 = ((const uint4 *)lut_g)[tid + 2u * BLOCK];
            if (tid < 256u) td = dtab_g[tid];
            ((uint4 *)smem)[tid] = t0; ((uint4 *)smem)[tid + BLOCK] = t1;
            if (tid + 2u * BLOCK < 2048u) ((uint4 *)smem)[tid + 2u * BLOCK] = t2;
            if (tid < 256u) dtab[tid] = make_uint2(td >> 16, 0u - (td & 0xffffu));
        } else {
            for (u32 i = tid; i < 2048; i += BLOCK) ((uint4 *)smem)[i] = ((const uint4 *)lut_g)[i];
            for (u32 i = tid; i < 256; i += BLOCK) { const u32 d = dtab_g[i]; dtab[i] = make_uint2(d >> 16, 0u - (d & 0xffffu)); }
        }
    }
    __syncthreads();

    WaveChunks wc;
    wc.c0 = (blockIdx.x * (BLOCK / 64) + wv) * 32u; wc.chunk = chunk; wc.nchunks = nchunks;
    wc.lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    if (wc.c0 >= nchunks) return;
    wc.rows = nchunks - wc.c0 < 32u ? nchunks - wc.c0 : 32u;
    // the directory of the whole group of 64 chunks (payload offsets are per group): lane L reads chunk (group base + L), the
    // pair takes its chunk's numbers from there
    const u32 g0 = wc.c0 & ~63u, cg = g0 + lane;
    const u32 lenL = cg < nchunks ? ((cg == nchunks - 1u) ? wc.lastlen : chunk) : 0u;
    const u32 clL = cg < nchunks ? trc_min(clen[cg], lenL) : 0u;   // a directory entry above the chunk length (corrupt input) reads as raw
    const u32 exL = trc_wave_incl_scan(clL) - clL;
    const u64 gbase = trc_group_base(goff, gsum, wc.c0 >> 6);
    const u32 ci = lane >> 1, b = lane & 1u, srcl = (wc.c0 & 32u) + ci;
    const u32 cl = (u32)__shfl((int)clL, (int)srcl, 64), ex = (u32)__shfl((int)exL, (int)srcl, 64), len = (u32)__shfl((int)lenL, (int)srcl, 64);
    const bool alive = ci < wc.rows;
    const u32 c = wc.c0 + ci;
    const u64 off = gbase + ex;
    const bool coded = alive && cl != len;
    const bool owner = coded && b == 0u;                       // the even lane owns the chunk's stream (ring, refills)

    QuadOut tout; tout.base = out + (u64)wc.c0 * chunk;
    AnsStreamIn si;
    si.rings = wbase;
    si.gbase = payload; si.soff = off + 8; si.lim = trc_sub_sat(cl, 8u);   // words follow the two states
    u32 st = 0;
    if (coded) st = trc_ld32_a2(payload + off + (b ? 0u : 4u));             // [enc state 1][enc state 0]: the even lane runs state 0
    si.prime(owner);

    const u32 S = chunk / TRC_SEG;
    const u32 body4 = len & ~3u;
    u8 *dst = out + (u64)c * chunk;
    const u32 rbase = (u32)(uintptr_t)(si.rings - smem) + AnsStreamIn::ra(lane & ~1u, 0);   // the PAIR's ring (the even lane's column)
    const u32 seld = b ? 0x0202u : 0u, sh0 = 8u * b;
    u32 hc = 0;                                                // the pair's halfword cursor (both lanes keep it)
    for (u32 s = 0; s < S; s++) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u32 p0 = s * TRC_SEG + (u32)k * 16u;          // chunk offset of this 16-byte piece
            si.rpos = hc << 1;
            si.period(owner && p0 < len, k & 1);
            if (k == 0 && s > 0) tout.flush(wc, (s - 1u) * TRC_SEG, true);   // the segment before, behind this period's commit
            const bool full = coded && p0 + 16u <= len;
            u32 w[4];
#pragma unroll
            for (int d = 0; d < 4; d++) {                       // a dword = two pairs of symbols: this lane's bytes are b and b + 2
                u32 m = 0;
                if (full) {
                    const u32 xa = ans_get_half(st, rbase, hc, b, seld);
                    const u32 xb = ans_get_half(st, rbase, hc, b, seld);
                    m = (xa | (xb << 16)) << sh0;
                }
                w[d] = m | trc_quad_xor1(m);                    // (every lane: the exchange must not sit in a branch)
            }
            if (!full && coded && p0 < len) {                   // last chunk's final partial piece: symbol by symbol, both lanes walking it
                for (u32 pos = p0; pos < len; pos++) {
                    const bool mine = ((pos < body4 && (pos & 1u)) ? 1u : 0u) == b;        // odd positions of whole groups: state 1; the tail: state 0
                    u32 rn = 0;
                    if (mine) {
                        const u32 sl = st & (TRC_PROB_ONE - 1);
                        const u32 x = *(const trc_lds_u8 *)(uintptr_t)(DEC_LDS_LUT + sl);
                        const trc_v2u e = *(const trc_lds_u64 *)(uintptr_t)(DEC_LDS_DTAB + (x << 3));
                        const u32 t = __umul24(e.x, st >> TRC_PROB_BITS) + e.y + sl;
                        const u32 a = rbase + (((hc >> 1) & 31u) << 8) + ((hc & 1u) << 1);
                        const u32 unit = trc_ldsr16(a);
                        rn = t < TRC_ANS_LOW ? 1u : 0u;
                        st = rn ? (t << 16) | unit : t;
                        dst[pos] = (u8)x;
                    }
                    hc += rn + trc_quad_xor1(rn);
                }
            }
            tout.put((u32)k, make_uint4(w[0], w[1], w[2], w[3]));
        }
    }
    tout.flush(wc, (S - 1u) * TRC_SEG, true);
    // raw chunks: lanes 0..31 carry their chunks' numbers for the wave copy
    {
        const u32 from = (lane & 31u) << 1;
        const u32 olo = (u32)__shfl((int)(u32)off, (int)from, 64), ohi = (u32)__shfl((int)(u32)(off >> 32), (int)from, 64);
        const u32 l2 = (u32)__shfl((int)len, (int)from, 64), c2 = (u32)__shfl((int)cl, (int)from, 64);
        const bool raw = lane < wc.rows && c2 == l2 && l2 != 0u;
        trc_wave_copy_raw(__ballot(raw), ((u64)ohi << 32) | olo, l2, out + (u64)wc.c0 * chunk, chunk, payload);
    }
}

// ------------------------------------------------------------------------------------- launch ---
// Encoder launch shape.  LDS per workgroup = REP x 4 KiB of symbol table + 8.3 KiB per wave; a CU holds 160 KiB.
//   REP 1 (default): 4 waves share one 4 KiB table, 37 KiB per workgroup -> 16 waves per CU
//   REP 8 (TRC_ENC_REP=8, 32 KiB, 12 waves per workgroup): kept as a measuring aid -- with the hand-issued b128 reads the
//   replicated table is NOT faster (100 MB, chunk 512: 64 vs 63 us; REP 16 at 11 waves per CU, chunk 576: 67 vs 68 us;
//   profiles/r02_notes.md): the bank conflicts round 1's PMC showed were not what the kernel waited for.
template <int REP>
static void ans4s_enc_launch(u32 wpb, bool fused, const uint8_t *d_in, size_t n, uint32_t chunk, const TrcWork &w, uint32_t *d_clen,
                             uint8_t *d_payload, uint64_t *d_total, hipStream_t s)
{
    const uint4 *etab = (const uint4 *)(w.tables + TRC_TAB_ENC);
    const u32 nwaves = w.ngroups;
    const size_t sm = 4096u * REP + ENC_PACE_LDS + wpb * ENC_WAVE_LDS;
    u8 *const sync = w.tables + TRC_TAB_SYNC;
#define TRC_ENC_CASE(W, F)                                                                                              \
    TRC_RAISE_LDS_ONCE((trc_ans4s_enc_kernel<64 * W, REP, F>), 4096u * REP + ENC_PACE_LDS + W * ENC_WAVE_LDS);          \
    TRC_LAUNCH_TIMED((trc_ans4s_enc_kernel<64 * W, REP, F>), dim3((nwaves + W - 1) / W), dim3(64 * W), sm, s,           \
                     d_in, (u64)n, chunk, w.nchunks, etab, w.scratch, w.stride, d_clen, w.gsum, d_payload, d_total, w.goff_area, sync);
    switch (wpb) {
    case 1: TRC_ENC_CASE(1, false) break;
    case 4: TRC_ENC_CASE(4, false) break;
    case 12: if (fused) { TRC_ENC_CASE(12, true) } else { TRC_ENC_CASE(12, false) } break;
    default: break;
    }
#undef TRC_ENC_CASE
}
// returns true when the encoder's waves have gathered the payload themselves (trc_gather.h): no gather launch behind it
bool trc_launch_ans4s_enc(const uint8_t *d_in, size_t n, uint32_t chunk, const TrcWork &w,
                          uint32_t *d_clen, uint8_t *d_payload, uint64_t *d_total, hipStream_t s)
{
    const u32 nwaves = w.ngroups;
    static const int env_rep = getenv("TRC_ENC_REP") ? atoi(getenv("TRC_ENC_REP")) : 0;     // tuning aids
    static const int env_wpb = getenv("TRC_ENC_WPB") ? atoi(getenv("TRC_ENC_WPB")) : 0;
    static const int env_fused = getenv("TRC_ENC_FUSED") ? atoi(getenv("TRC_ENC_FUSED")) : TRC_ENC_FUSED;
    int rep = env_rep ? env_rep : TRC_ENC_REP_DEFAULT;
    u32 wpb = rep == 8 ? 12u : 4u;
    // one residency round (at most twelve waves per CU): one workgroup of twelve waves per CU, whose waves keep each other's pace
    // (TrcPace: they share SIMDs by construction); more than a round: workgroups of four, sixteen waves per CU, new ones moving in
    // as old ones end
    if (TRC_ENC_BALANCE && nwaves <= 12u * 256u) wpb = 12u;
    if (nwaves < 2048) { rep = 1; wpb = 1; }                 // few waves: one per workgroup so they spread over all CUs (12.4 KiB -> 12 per CU)
    if (env_wpb == 1 || env_wpb == 4 || env_wpb == 12) wpb = (u32)env_wpb;      // (tuning aid / tests: forces the shape whatever the size)
    // the twelve-wave workgroups of a one-round launch gather their own payload (no scan kernel in the way: w.goff == NULL)
    const bool fused = env_fused && wpb == 12u && !w.goff && (nwaves + 11u) / 12u <= TRC_SYNC_MAX_WG;
    if (fused) (void)hipMemsetAsync(w.tables + TRC_TAB_SYNC, 0, TRC_SYNC_BYTES, s);      // (not "left zero by the last launch": ADVICE r5)
    if (rep == 8) ans4s_enc_launch<8>(wpb, fused, d_in, n, chunk, w, d_clen, d_payload, d_total, s);
    else ans4s_enc_launch<1>(wpb, fused, d_in, n, chunk, w, d_clen, d_payload, d_total, s);
#ifdef TRC_ENC_PROF
    {
        static int calls = 0;
        if (++calls % 64 == 0) {
            static unsigned long long wl[4 * 4096];
            (void)hipDeviceSynchronize();
            (void)hipMemcpyFromSymbol(wl, HIP_SYMBOL(trc_enc_wall), sizeof wl);
            const unsigned nw = nwaves < 4096u ? nwaves : 4096u;
            unsigned long long t0 = ~0ull;
            for (unsigned i = 0; i < nw; i++) if (wl[4 * i] && wl[4 * i] < t0) t0 = wl[4 * i];
            double s1[3] = { 0, 0, 0 }, s2[3] = { 0, 0, 0 }, s3[3] = { 0, 0, 0 }, mx[3] = { 0, 0, 0 }; unsigned cnt[3] = { 0, 0, 0 }; double smax = 0;
            for (unsigned i = 0; i < nw; i++) {
                const unsigned age = wpb >= 12 ? (i % wpb) / 4u : 0; if (age > 2 || !wl[4 * i]) continue;
                const double a = (wl[4 * i] - t0) * 0.01, b = (wl[4 * i + 1] - t0) * 0.01, c2 = (wl[4 * i + 2] - t0) * 0.01, d = (wl[4 * i + 3] - t0) * 0.01;
                s1[age] += b; s2[age] += c2; s3[age] += d; cnt[age]++; if (d > mx[age]) mx[age] = d; if (a > smax) smax = a;
            }
            fprintf(stderr, "[enc wall, us since the first wave's start] wpb %u, last start %.2f; by age on the SIMD: loop starts / loop ends / wave ends (mean; max end):", wpb, smax);
            for (int a = 0; a < 3; a++) if (cnt[a]) fprintf(stderr, "  %.1f / %.1f / %.1f; %.1f", s1[a] / cnt[a], s2[a] / cnt[a], s3[a] / cnt[a], mx[a]);
            fprintf(stderr, "\n");
        }
    }
#endif
    return fused;
}

void trc_launch_ans4s_dec(const uint8_t *d_payload, const uint32_t *d_clen, size_t n, uint32_t chunk,
                          const TrcWork &w, uint8_t *d_out, hipStream_t s)
{
    const u8 *lut = w.tables + TRC_TAB_LUT;
    const u32 *dtab = (const u32 *)(w.tables + TRC_TAB_DEC);
    const u32 nwaves = w.ngroups;
    static const int env_pair = getenv("TRC_ANS_PAIR") ? atoi(getenv("TRC_ANS_PAIR")) : -1;      // tuning aid: 0 / 1 force the form
    if (env_pair == 1 || (env_pair != 0 && nwaves < 2048u)) {  // too few chunks to fill the chip with one lane each: two lanes per chunk
        const u32 nw2 = (w.nchunks + 31u) / 32u;
        TRC_RAISE_LDS_ONCE(trc_ans4s_dec2_kernel, DEC_LDS_WAVES + 14 * DEC_WAVE_LDS);
        u32 wpb2 = (nw2 + 255u) / 256u;
        wpb2 = wpb2 < 1u ? 1u : wpb2 > 14u ? 14u : wpb2;
        TRC_LAUNCH_TIMED(trc_ans4s_dec2_kernel, dim3((nw2 + wpb2 - 1) / wpb2), dim3(64 * wpb2), DEC_LDS_WAVES + wpb2 * DEC_WAVE_LDS, s,
                           d_payload, d_clen, w.goff, w.gsum, (u64)n, chunk, w.nchunks, lut, dtab, d_out);
        return;
    }
    TRC_RAISE_LDS_ONCE(trc_ans4s_dec_kernel, DEC_LDS_WAVES + 14 * DEC_WAVE_LDS);
    // the 34 KiB of tables are per workgroup, so waves share a workgroup -- but no more than it takes to
    // give every one of the 256 CUs a workgroup (1526 waves: 6 per workgroup -> 255 workgroups); up to 14 fit
    u32 wpb = (nwaves + 255u) / 256u;
    wpb = wpb < 1u ? 1u : wpb > 14u ? 14u : wpb;               // 34 KiB + 14 x 8.3 KiB = 150 KiB of the CU's 160
    if (const char *e = getenv("TRC_DEC_WPB")) { const u32 v = (u32)atoi(e); if (v >= 1 && v <= 14) wpb = v; }   // tuning aid
    const size_t sm = DEC_LDS_WAVES + wpb * DEC_WAVE_LDS;
#ifdef TRC_DEC_PROF
#endif
    TRC_LAUNCH_TIMED(trc_ans4s_dec_kernel, dim3((nwaves + wpb - 1) / wpb), dim3(64 * wpb), sm, s,
                       d_payload, d_clen, w.goff, w.gsum, (u64)n, chunk, w.nchunks, lut, dtab, d_out);
#ifdef TRC_DEC_PROF
    {
        static int calls = 0;
        if (++calls % 64 == 0) {
            unsigned long long h[8];
            (void)hipDeviceSynchronize();
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(trc_dec_prof), sizeof h);
            const double wv = (double)h[6];
            fprintf(stderr, "[dec prof] waves %.0f  cycles per wave: fill %.0f  dir+prime %.0f  periods %.0f  flushes %.0f  symbols %.0f  total %.0f\n",
                    wv, h[0] / wv, h[1] / wv, h[2] / wv, h[3] / wv, h[4] / wv, h[5] / wv);
            memset(h, 0, sizeof h);
            (void)hipMemcpyToSymbol(HIP_SYMBOL(trc_dec_prof), h, sizeof h);
            static unsigned long long wl[2 * 4096];
            (void)hipMemcpyFromSymbol(wl, HIP_SYMBOL(trc_dec_wall), sizeof wl);
            const unsigned nw = nwaves < 4096u ? nwaves : 4096u;
            unsigned long long t0 = ~0ull;
            for (unsigned i = 0; i < nw; i++) if (wl[2 * i] && wl[2 * i] < t0) t0 = wl[2 * i];
            // end times by the wave's place in its workgroup (waves k, k + 4, k + 8 of a workgroup share SIMD k: age order on the SIMD)
            double sum[3] = { 0, 0, 0 }, mx[3] = { 0, 0, 0 }, mn[3] = { 1e9, 1e9, 1e9 }; unsigned cnt[3] = { 0, 0, 0 };
            double smax = 0;
            for (unsigned i = 0; i < nw; i++) {
                const unsigned age = (i % wpb) / 4u; if (age > 2 || !wl[2 * i]) continue;
                const double e = (wl[2 * i + 1] - t0) * 0.01, st = (wl[2 * i] - t0) * 0.01;
                sum[age] += e; cnt[age]++; if (e > mx[age]) mx[age] = e; if (e < mn[age]) mn[age] = e; if (st > smax) smax = st;
            }
            fprintf(stderr, "[dec wall, us since the first wave's start] last start %.2f; wave ends by age on the SIMD (min / mean / max): first %.1f / %.1f / %.1f, second %.1f / %.1f / %.1f, third %.1f / %.1f / %.1f\n",
                    smax, mn[0], sum[0] / (cnt[0] ? cnt[0] : 1), mx[0], mn[1], sum[1] / (cnt[1] ? cnt[1] : 1), mx[1], mn[2], sum[2] / (cnt[2] ? cnt[2] : 1), mx[2]);
        }
    }
#endif
}
