// trc_ans_o1.hip -- order-1 adaptive-CDF byte rANS (codec TRC_ANSO1; anscdf1enc / anscdf1dec, anscdf.c:607-645,
// mnenc8x2x / mndec8x2x anscdf_.h:121-126,164-174; `turborc -e64`) -- SURVEY 8f rank 2.
//
// The coder of trc_ans_adaptive.hip with the CDF16 tables selected by the previous byte: hi table by ctx, lo table
// by (ctx, hi nibble); 256 x 17 tables x 32 B = 136 KiB per chunk, context 0 at the start of a chunk, an odd tail
// pairs with a coded dummy 0.  Payload layout, record format, raw rule and the backward coding pass are those of
// TRC_ANSA (the second pass IS trc_ansa_code_kernel<false>).
//
// 136 KiB per chunk rules LDS out (one chunk per CU would leave 256 lanes busy on the whole chip).  The models live
// in HBM instead -- 288 GB per GPU hold two million of them -- one contiguous block per chunk, each lane working on
// its own: a table is two 16-byte global accesses per lane, the update runs on registers (K table in LDS,
// trc_nibmodel.h), the symbol bounds come out of the register copy by select trees.
//
// Round 3.  PMC arithmetic says these kernels are bound by the texture-address unit: every table access is 64 scattered
// 16-byte lane accesses, four wave-instructions per nibble (load x 2, store x 2).  So the accesses themselves were cut:
//   * each lane keeps the hi table and the lo table it used last IN REGISTERS, with their identities, and goes to memory
//     only when the next nibble needs a different table (write the old one back, fetch the new one) -- loads and stores
//     run under the lanes' own predicates, so a wave-instruction costs the unit only the lanes that miss.  In run-heavy
//     data (what an order-1 coder is for: BWT output) most bytes repeat their predecessor and both tables hit;
//   * FIRST TOUCH: a table that has not been written in this call is not read either -- it IS the initial table
//     (cdf[j] = j << 11).  256 "context seen" bits per lane live in LDS; the 16 "lo table seen" bits of a context ride in
//     entry 0 of its hi table, which the model keeps at 0 (K[x][0] = 0) and which is masked out while the table is in
//     registers.  The 136 KiB-per-chunk fill kernel of rounds 1-2 (3.3 GB per call at 100 MB / chunk 4096, twice per
//     step) is gone, and so is the first read of every table.
#include <stdlib.h>
#include "trc_io.h"
#include "trc_lane_io.h"
#include "trc_nibmodel.h"
#include "trc_launch.h"

#define O1_MODEL_BYTES TRC_O1_MODEL_BYTES
#ifndef TRC_O1_DEC_DEFAULT_ROWS                               // decoder form: 1 = eight lanes per chunk; 16 / 64 = one lane per chunk, that many chunks per wave
#define TRC_O1_DEC_DEFAULT_ROWS(ngroups) ((ngroups) <= 128u ? 1 : 4)
#endif                    // 139264 per chunk

__device__ __forceinline__ NibTable o1_load(const u8 *tb)
{
    NibTable T;
    const uint4 a = *(const uint4 *)tb, b = *(const uint4 *)(tb + 16);
    T.d[0] = a.x; T.d[1] = a.y; T.d[2] = a.z; T.d[3] = a.w; T.d[4] = b.x; T.d[5] = b.y; T.d[6] = b.z; T.d[7] = b.w;
    return T;
}
__device__ __forceinline__ void o1_store(u8 *tb, const NibTable &T)
{
    *(uint4 *)tb = make_uint4(T.d[0], T.d[1], T.d[2], T.d[3]);
    *(uint4 *)(tb + 16) = make_uint4(T.d[4], T.d[5], T.d[6], T.d[7]);
}
// entry pair (t[x], t[x+1]) out of the register copy (entry 16 = 32768)
__device__ __forceinline__ void o1_bounds(const NibTable &T, u32 x, u32 &c0, u32 &c1)
{
    // (as bit-selects under sign-extended bits of x -- no runs of VOP2 v_cndmask -- the model pass measured 3.65 -> 3.90 ms: selects kept)
    const bool b2 = x & 8u, b1 = x & 4u, b0 = x & 2u;
    const u32 e0 = b2 ? T.d[4] : T.d[0], e1 = b2 ? T.d[5] : T.d[1], e2 = b2 ? T.d[6] : T.d[2],
              e3 = b2 ? T.d[7] : T.d[3], e4 = b2 ? TRC_PROB_ONE : T.d[4];
    const u32 f0 = b1 ? e2 : e0, f1 = b1 ? e3 : e1, f2 = b1 ? e4 : e2;
    const u32 g0 = b0 ? f1 : f0, g1 = b0 ? f2 : f1;
    const bool odd = x & 1u;
    c0 = odd ? g0 >> 16 : g0 & 0xffffu;
    c1 = odd ? g1 & 0xffffu : g0 >> 16;
}
__device__ __forceinline__ void o1_adapt(NibTable &T, const u8 *kb, u32 x)
{
    const uint4 a = *(const uint4 *)(kb + x * 32u), b = *(const uint4 *)(kb + x * 32u + 16);
    const u32 K[8] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const trc_s2 v = trc_as_s2(T.d[k]);
        T.d[k] = trc_as_u32(v + ((trc_as_s2(K[k]) - v) >> (trc_s2)7));
    }
}
__device__ __forceinline__ void o1_init_k(u8 *kb)
{
    const u32 lane = trc_lane();
#pragma unroll
    for (u32 j = 0; j < 2; j++) {
        const u32 idx = lane * 2u + j, x = idx >> 3, k = idx & 7u;
        const u32 e0 = 2u * k, e1 = 2u * k + 1u;
        ((u32 *)kb)[idx] = trc_pk(10u * e0 + (e0 > x ? 32736u : 0u), 10u * e1 + (e1 > x ? 32736u : 0u));
    }
    __syncthreads();
}

// The per-lane table cache (header comment).  Tables are named by their index in the chunk's model block: hi table of
// context c = 17 c, lo table (c, h) = 17 c + 1 + h.
#define O1_SEEN_BYTES (64u * 32u)                            // 256 context bits per lane
struct O1Cache {
    NibTable H, L;
    u32 hid, lid;        // what H / L hold (~0u: nothing yet)
    u32 hmask;           // lo tables of H's context written so far (entry 0 of the hi table in memory)
    u8 *mine;            // this lane's model block
    u32 seen;            // LDS byte address of this lane's context bits
    __device__ __forceinline__ static NibTable fresh()
    {
        NibTable T;
#pragma unroll
        for (u32 k = 0; k < 8; k++) T.d[k] = trc_pk((2 * k) << 11, (2 * k + 1) << 11);
        return T;
    }
    // every lane of the wave must call (one wave per workgroup: the barrier orders the zeroing before the first read)
    __device__ __forceinline__ void init(u8 *model_block, u8 *seen_lds)
    {
        typedef __attribute__((address_space(3))) u32 lds_u32;
        mine = model_block; hid = lid = ~0u; hmask = 0; H = fresh(); L = fresh();
        seen = trc_lds_addr(seen_lds) + trc_lane() * 32u;
#pragma unroll
        for (u32 k = 0; k < 8; k++) *(lds_u32 *)(uintptr_t)(seen + 4u * k) = 0u;
        __syncthreads();
    }
    // make H the hi table of context cx (only lanes with `on`)
    __device__ __forceinline__ void need_hi(bool on, u32 cx)
    {
        typedef __attribute__((address_space(3))) u32 lds_u32;
        const u32 id = cx * 17u;
        if (on && id != hid) {
            if (hid != ~0u) { NibTable W = H; W.d[0] |= hmask; o1_store(mine + (size_t)hid * 32u, W); }
            const u32 a = seen + ((cx >> 5) << 2), bits = *(const lds_u32 *)(uintptr_t)a, bit = 1u << (cx & 31u);
            if (bits & bit) { H = o1_load(mine + (size_t)id * 32u); hmask = H.d[0] & 0xffffu; H.d[0] &= 0xffff0000u; }
            else { H = fresh(); hmask = 0; *(lds_u32 *)(uintptr_t)a = bits | bit; }
            hid = id;
        }
    }
    // make L the lo table (cx, h); H must be the hi table of cx
    __device__ __forceinline__ void need_lo(bool on, u32 cx, u32 h)
    {
        const u32 id = cx * 17u + 1u + h;
        if (on && id != lid) {
            if (lid != ~0u) o1_store(mine + (size_t)lid * 32u, L);
            if ((hmask >> h) & 1u) L = o1_load(mine + (size_t)id * 32u);
            else { L = fresh(); hmask |= 1u << h; }
            lid = id;
        }
    }
};

// ------------------------------------------------------------------------------ encode, pass 1 ---
__global__ __launch_bounds__(64) void trc_o1_model_kernel(
    const u8 *__restrict__ in, u64 n, u32 chunk, u32 nchunks, u8 *__restrict__ model, u8 *__restrict__ recs)
{
    __shared__ __attribute__((aligned(16))) u8 kb[TRC_NIBK_BYTES];
    __shared__ __attribute__((aligned(16))) u8 seen[O1_SEEN_BYTES];
    const u32 lane = threadIdx.x;
    o1_init_k(kb);

    WaveChunks wc;
    wc.c0 = blockIdx.x * 64u; wc.chunk = chunk; wc.nchunks = nchunks;
    wc.lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    wc.rows = nchunks - wc.c0 < 64u ? nchunks - wc.c0 : 64u;
    WaveChunks wr = wc;
    wr.chunk = 8u * chunk; wr.lastlen = 8u * (wc.lastlen + (wc.lastlen & 1u));
    const bool alive = lane < wc.rows;
    const u32 len = alive ? wc.len_of(lane) : 0u;
    const u32 plen = len + (len & 1u);                         // bytes coded, dummy included
    O1Cache tc;
    tc.init(model + (u64)(wc.c0 + (alive ? lane : 0u)) * O1_MODEL_BYTES, seen);   // dead lanes alias lane 0's model but never touch it

    QuadIn qin; qin.base = in + (u64)wc.c0 * chunk;
    QuadOut qout; qout.base = recs + (u64)wc.c0 * wr.chunk;
    u32 cx = 0;

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
            if (!__ballot(alive && p0 < len)) continue;
            const u32 w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
            for (int h = 0; h < 2; h++) {                      // 8 input bytes -> 16 records = one 64-byte record segment
                u32 r[16];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const u32 pos = p0 + 8u * (u32)h + (u32)i;
                    u32 x = (w[2 * h + (i >> 2)] >> (8 * (i & 3))) & 255u;
                    if (pos >= len) x = 0;                     // the coded dummy of an odd tail (and unused padding)
                    r[2 * i] = r[2 * i + 1] = 0;
                    const bool on = alive && pos < plen;
                    tc.need_hi(on, cx); tc.need_lo(on, cx, x >> 4);
                    if (on) {
                        u32 a0, a1, b0, b1;
                        o1_bounds(tc.H, x >> 4, a0, a1); o1_bounds(tc.L, x & 15u, b0, b1);
                        o1_adapt(tc.H, kb, x >> 4); o1_adapt(tc.L, kb, x & 15u);
                        r[2 * i] = (a0 << TRC_PROB_BITS) | (a1 - a0);
                        r[2 * i + 1] = (b0 << TRC_PROB_BITS) | (b1 - b0);
                        cx = x;
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; j++) qout.put((u32)j, make_uint4(r[4 * j], r[4 * j + 1], r[4 * j + 2], r[4 * j + 3]));
                qout.flush(wr, (p0 + 8u * (u32)h) * 8u);
            }
        }
    }
}

// Pass 1 as TWO WAVES per 64 chunks (round 4; the scheme of trc_ansa_model2_kernel, trc_ans_adaptive.hip).  The hi record of a byte
// needs the hi table of its context alone, the lo record the lo table of (context, hi nibble) alone -- and both selectors are INPUT
// (the byte before, this byte's hi nibble), not model state.  So a "hi" wave and a "lo" wave walk the same bytes, each with ONE
// cached table per lane and its own first-touch bits (hi: 256 contexts per lane; lo: 4096 tables per lane, [word][lane] in LDS),
// on disjoint tables of the chunk's model block, with nothing to tell each other: no barrier.  A lane's chain is one table round
// trip per byte instead of two, and the two chains of a chunk run side by side.  Records go to the PLANAR record space (64 B of
// hi records, 64 B of lo records per 16 input bytes), which the four-lanes-per-chunk coding pass reads.  Workgroup = W hi waves
// + W lo waves; launched with W = 1 (see trc_launch_anso1_model).
#define O1M2_KB      0u
#define O1M2_HSEEN   TRC_NIBK_BYTES                             // u32[8][64]
#define O1M2_LSEEN   (TRC_NIBK_BYTES + 2048u)                   // u32[128][64]
// End of round 4: between a lane's one table in registers and its model block in HBM sits a small write-back cache in LDS --
// O1C_E entries per lane, direct-mapped by a hash of the table's index, the tag in the low half of dword 0 (entry 0 of a CDF16
// table is always 0).  On data whose statistics drift (what an order-1 coder is for) a lane works on a handful of contexts at a
// time: a register miss was a round trip to HBM on 61 % (hi wave) / 70 % (lo wave) of the bytes; with eight entries it is one
// on 9 % / 21 % (simulation over `drift_bytes`; i.i.d. bytes gain nothing: 95 % -> 67 % / 89 %).  A miss evicts the slot's
// occupant to the model block and fills from it (first-touch rule unchanged); nothing is flushed at the end -- the model of a
// chunk is not needed after its records.
#define O1C_E        8u
#define O1C_BYTES    (O1C_E * 2u * 64u * 16u)                   // [entry][half][lane] 16 B
#define O1M2_CACHE   (TRC_NIBK_BYTES + 2048u + 32768u)          // the hi wave's cache, then the lo wave's
#define O1M2_LDS     (TRC_NIBK_BYTES + 2048u + 32768u + 2u * O1C_BYTES)
__device__ __forceinline__ u32 o1c_slot(u32 id) { return (id ^ (id >> 4) ^ (id >> 8)) & (O1C_E - 1u); }
template <u32 W>                                               // pairs per workgroup
__global__ __launch_bounds__(128 * W) void trc_o1_model2_kernel(
    const u8 *__restrict__ in, u64 n, u32 chunk, u32 nchunks, u8 *__restrict__ model, u8 *__restrict__ recs)
{
    typedef __attribute__((address_space(3))) u32 lds_u32;
    extern __shared__ __attribute__((aligned(16))) u8 smem_wg_[];
    const u32 wv_ = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool lo_wave = wv_ >= W;
    const u32 grp_ = blockIdx.x * W + (wv_ & (W - 1u));
    if (grp_ >= (nchunks + 63u) / 64u) return;
    u8 *const smem = smem_wg_ + (wv_ & (W - 1u)) * O1M2_LDS;
    const u32 lane = trc_lane();
    u8 *const kb = smem + O1M2_KB;
    {                                                          // both waves write the same K; each zeroes its own first-touch bits
#pragma unroll
        for (u32 j = 0; j < 2; j++) {
            const u32 idx = lane * 2u + j, x = idx >> 3, k = idx & 7u;
            const u32 e0 = 2u * k, e1 = 2u * k + 1u;
            ((u32 *)kb)[idx] = trc_pk(10u * e0 + (e0 > x ? 32736u : 0u), 10u * e1 + (e1 > x ? 32736u : 0u));
        }
        u32 *z = (u32 *)(smem + (lo_wave ? O1M2_LSEEN : O1M2_HSEEN));
        for (u32 i = lane; i < (lo_wave ? 128u * 64u : 8u * 64u); i += 64u) z[i] = 0u;
        uint4 *zc = (uint4 *)(smem + O1M2_CACHE + (lo_wave ? O1C_BYTES : 0u));
        for (u32 i = lane; i < O1C_BYTES / 16u; i += 64u) zc[i] = make_uint4(0, 0, 0, 0);        // (tag 0: empty)
        trc_wave_lds_fence();
    }
    u8 *const cache = smem + O1M2_CACHE + (lo_wave ? O1C_BYTES : 0u) + lane * 16u;             // this lane's column: entry e at + e * 2048, second half + 1024
    const u32 seen = trc_lds_addr(smem) + (lo_wave ? O1M2_LSEEN : O1M2_HSEEN) + lane * 4u;      // word w of this lane at seen + 256 w

    WaveChunks wc;
    wc.c0 = grp_ * 64u; wc.chunk = chunk; wc.nchunks = nchunks;
    wc.lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    wc.rows = nchunks - wc.c0 < 64u ? nchunks - wc.c0 : 64u;
    WaveChunks wr = wc;
    wr.chunk = 8u * chunk; wr.lastlen = 128u * ((wc.lastlen + 15u) / 16u);      // planar record space
    const bool alive = lane < wc.rows;
    const u32 len = alive ? wc.len_of(lane) : 0u;
    const u32 plen = len + (len & 1u);                         // bytes coded, dummy included
    u8 *const mine = model + (u64)(wc.c0 + (alive ? lane : 0u)) * O1_MODEL_BYTES;   // dead lanes alias lane 0's model but never touch it

    NibTable Tc = O1Cache::fresh();                            // the one table this lane holds, and which (~0u: none yet)
    u32 tid_c = ~0u;
    // make Tc table `id` of the model block (first-touch bit `bit` of word `wd`), only where `on`
    auto need = [&](bool on, u32 id, u32 fb) __attribute__((always_inline)) {
        if (on && id != tid_c) {
            if (tid_c != ~0u) {                                // the table in hand goes back to its slot (tag = index + 1)
                u8 *sl = cache + o1c_slot(tid_c) * 2048u;
                *(uint4 *)sl = make_uint4(Tc.d[0] | (tid_c + 1u), Tc.d[1], Tc.d[2], Tc.d[3]);
                *(uint4 *)(sl + 1024u) = make_uint4(Tc.d[4], Tc.d[5], Tc.d[6], Tc.d[7]);
            }
            const u8 *sl = cache + o1c_slot(id) * 2048u;
            const uint4 c0 = *(const uint4 *)sl, c1 = *(const uint4 *)(sl + 1024u);
            const u32 tag = c0.x & 0xffffu;
            if (tag == id + 1u) {
                Tc.d[0] = c0.x & 0xffff0000u; Tc.d[1] = c0.y; Tc.d[2] = c0.z; Tc.d[3] = c0.w;
                Tc.d[4] = c1.x; Tc.d[5] = c1.y; Tc.d[6] = c1.z; Tc.d[7] = c1.w;
            } else {
                if (tag) {                                     // the slot's occupant leaves for the model block
                    u8 *tb = mine + (size_t)(tag - 1u) * 32u;
                    *(uint4 *)tb = make_uint4(c0.x & 0xffff0000u, c0.y, c0.z, c0.w);
                    *(uint4 *)(tb + 16) = c1;
                }
                const u32 a = seen + ((fb >> 5) << 8), bits = *(const lds_u32 *)(uintptr_t)a, bit = 1u << (fb & 31u);
                if (bits & bit) Tc = o1_load(mine + (size_t)id * 32u);
                else { Tc = O1Cache::fresh(); *(lds_u32 *)(uintptr_t)a = bits | bit; }
            }
            tid_c = id;
        }
    };

    QuadIn qin; qin.base = in + (u64)wc.c0 * chunk;
    QuadOut qout; qout.base = recs + (u64)wc.c0 * wr.chunk;
    u32 cx = 0;
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
            if (!__ballot(alive && p0 < len)) continue;
            const u32 w[4] = { v.x, v.y, v.z, v.w };
            u32 r[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {                     // 16 input bytes -> 16 records of this wave's plane
                const u32 pos = p0 + (u32)i;
                u32 x = (w[i >> 2] >> (8 * (i & 3))) & 255u;
                if (pos >= len) x = 0;                         // the coded dummy of an odd tail (and unused padding)
                r[i] = 0;
                const bool on = alive && pos < plen;
                const u32 h = x >> 4, sym = lo_wave ? x & 15u : h;
                if (lo_wave) need(on, cx * 17u + 1u + h, cx * 16u + h); else need(on, cx * 17u, cx);
                if (on) {
                    u32 a0, a1;
                    o1_bounds(Tc, sym, a0, a1);
                    o1_adapt(Tc, kb, sym);
                    r[i] = (a0 << TRC_PROB_BITS) | (a1 - a0);
                    cx = x;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) qout.put((u32)j, make_uint4(r[4 * j], r[4 * j + 1], r[4 * j + 2], r[4 * j + 3]));
            qout.flush(wr, p0 * 8u + (lo_wave ? 64u : 0u));
        }
    }
}

// ------------------------------------------------------------------- encode, pass 1 by CHAINS ---
// End of round 5.  The model passes above are chains of table round trips because they walk a chunk in POSITION order, every byte
// touching the table of its own context.  But the encoder knows the whole input: the record of a byte depends only on the state of
// ITS table, i.e. on the earlier bytes of the chunk that used the same table, in order.  The 4352 tables of a chunk are independent
// adaptive chains -- hi table c over the bytes that follow byte value c, lo table (c, h) over those whose hi nibble is h as well --
// and a chain that starts from the initial table needs no model memory: its one table lives in the registers of the lane walking it.
//   A  trc_o1_sort_kernel (a wave per chunk): a stable counting sort of the chunk's positions by table -- histogram by LDS atomics,
//      one scan over the 4352 bins, then row by row (64 positions) the rank of a position among the lanes with the same key (the
//      key's bits as ballots: deterministic, no returning atomics).  Writes the chunk's STREAM -- u32 entries `pos | nibble << 12`, every
//      chain contiguous and starting on a multiple of four entries (a WINDOW, 16 bytes), hi chains first; HEAD on a chain's first
//      entry (the up to three slots behind its last one hold whatever memory held: walked like entries, their records never read)
//      -- for every 128 entries the first chain start at or behind them (`nh`), the chunk's long chains, and for every position
//      where its two records will lie in that order (`perm`).
//   B  trc_o1_walk_kernel (a workgroup per G chunks, G <= 96: round 6, below): the unit of work is "the chains that START in entries
//      [128 j, 128 j + 128) of chunk i", taken off a list in LDS -- first the units of the group's long chains, longest first (the
//      tail of this kernel is its longest chain: drift100m has 16 hi chains per chunk, the longest 1400 entries on average, up to
//      a whole chunk's 4096), then every other unit a chain starts in.  A lane walks from the unit's first chain start to the first chain start behind
//      the unit (both in `nh`: no lane ever looks at a slot another lane writes) -- bounds, adapt, record -- so every chain is
//      walked by exactly one lane, whole.  All lanes busy whatever the statistics.  The lanes advance in lockstep, a ROUND = four
//      windows per lane: at its top the next round's windows are asked for (`global_load_lds_dwordx4` into the other half of an
//      eight-row ring, one request per slot whoever asks) and the last round's records go back IN PLACE of their entries; the one
//      `s_waitcnt vmcnt(0)` of the loop then finds everything a round old.  With compiler-placed waits every entry load also waited
//      for the previous record store (6.6 ms for this kernel).  The stores: a lane's four windows of a round are one 64-byte sector
//      (window k of a stream only in slot k mod 4), and records and addresses are transposed across the quad so that one
//      instruction's four lanes write one lane's sector -- as 16-byte pieces of 64 different lines they cost the memory system a
//      sector each (1.43 ms, 0.99 without the stores; 1.09 so).  profiles/r05_notes.md.
//   C  trc_o1_place_kernel (a workgroup per chunk): the chunk's records in stream order through LDS into the planar record space
//      (`perm`), which the coding pass reads as before (written straight to their positions by B they were 2 x 10^8 four-byte
//      writes scattered over 800 MB: 5.5 ms).
// Everything lives where the chunk's order-1 model block would (w.model, 139264 B per chunk; 101 824 used at most).  Positions are
// 12 bits: chunks up to 4096 bytes (what trc_round_chunk gives this coder); longer chunks take the kernels above.
#define O1S_STREAM   0u                                         // u32[21248 + 4]: 2 x 4096 entries + at most 3 unused slots behind each of 4352 chains
#define O1S_PERM     85008u                                     // u32[4096]: position -> (stream index of its hi record | lo << 16)
#define O1S_NH       101392u                                    // u16[176]: the first chain start at or behind entry 128 j (none: the stream's length)
#define O1S_LEN      101744u                                    // u32: entries of the stream (a multiple of 4)
#define O1S_BIG      101760u                                    // u32[16]: the chunk's long chains, length << 8 | unit (0: none)
#define O1S_BIGMIN   256u
#define O1S_SEGS     168u                                       // units of 128 entries
#define O1S_NHN      176u
#define O1S_HEAD     0x10000u
#define O1S_BINS     4352u
#define O1S_MARKS    (4u * O1S_SEGS * 4u)                       // one bit per stream slot
#define O1S_SORT_LDS (O1S_BINS * 2u + O1S_MARKS)                // packed u16 bins, one bit per stream slot: a chain starts here
__device__ __forceinline__ u32 o1s_up4(u32 v) { return (v + 3u) & ~3u; }
#ifndef O1S_EU_ATTR
#define O1S_EU_ATTR __attribute__((amdgpu_waves_per_eu(3, 3)))
#endif
// Round 6: the chunk's bytes are no longer staged in LDS (4 KiB of a workgroup's 15.7: ten workgroups per CU) -- a row of 64 positions is one
// coalesced byte load, asked for a row ahead, and a position's context is its left neighbour's byte (DPP wave_shr:1, the row before's last byte
// carried in) -- so that three waves share a SIMD: the kernel waited 45 % of its wave-cycles on LDS round trips at two (r06_pmc_o1.txt).
__global__ __launch_bounds__(64) O1S_EU_ATTR void trc_o1_sort_kernel(
    const u8 *__restrict__ in, u64 n, u32 chunk, u32 nchunks, u8 *__restrict__ model)
{
    __shared__ __attribute__((aligned(16))) u8 smem_[O1S_SORT_LDS];
    __shared__ u32 big_[17];                                    // [16]: how many
    const u32 lane = threadIdx.x;
    if (lane < 17u) big_[lane] = 0u;
    u8 *const bins = smem_;
    u32 *const hbits = (u32 *)(bins + O1S_BINS * 2u);
    const u32 c = blockIdx.x;
    const u32 lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    const u32 len = c == nchunks - 1u ? lastlen : chunk, plen = len + (len & 1u);
    u8 *const blk = model + (u64)c * O1_MODEL_BYTES;
    const u8 *src = in + (u64)c * chunk;
    for (u32 i = lane; i < (O1S_BINS * 2u + O1S_MARKS) / 16u; i += 64u) ((uint4 *)bins)[i] = make_uint4(0, 0, 0, 0);
    trc_wave_lds_fence();
    const u32 R = (plen + 63u) >> 6;
    // a row's bytes: x = the byte at the lane's position (0 behind the chunk's end: the coded dummy), pv = the byte before it
    auto row_load = [&](u32 r) -> u32 { const u32 p = r * 64u + lane; return p < len ? (u32)src[p] : 0u; };
    auto left_of = [&](u32 x, u32 &carry) -> u32 {
        const u32 pv = (u32)__builtin_amdgcn_update_dpp((int)carry, (int)x, 0x138, 0xf, 0xf, false);   // wave_shr:1 -- lane 0 keeps `carry`
        carry = (u32)__builtin_amdgcn_readlane((int)x, 63);
        return pv;
    };
    {
        u32 xn = row_load(0), carry = 0;
        for (u32 r = 0; r < R; r++) {
            const u32 pos = r * 64u + lane;
            const u32 x = xn;
            if (r + 1u < R) xn = row_load(r + 1u);
            const u32 pv = left_of(x, carry);
            if (pos < plen) {
                const u32 kh = pv, kl = 256u + (pv << 4) + (x >> 4);
                atomicAdd((u32 *)(bins + (kh >> 1) * 4u), 1u << ((kh & 1u) * 16u));
                atomicAdd((u32 *)(bins + (kl >> 1) * 4u), 1u << ((kl & 1u) * 16u));
            }
        }
    }
    trc_wave_lds_fence();
    u32 *const ent = (u32 *)(blk + O1S_STREAM);
    {                                                           // counts -> window-aligned starts (68 bins per lane); a non-empty bin marks its ends
        u32 cw[34];
#pragma unroll
        for (int i = 0; i < 34; i++) cw[i] = ((const u32 *)(bins + lane * 136u))[i];
        u32 tot = 0;
#pragma unroll
        for (int i = 0; i < 34; i++) tot += o1s_up4(cw[i] & 0xffffu) + o1s_up4(cw[i] >> 16);
        const u32 incl = trc_wave_incl_scan(tot);
        u32 start = incl - tot;
#pragma unroll
        for (int i = 0; i < 34; i++) {
            const u32 a = cw[i] & 0xffffu, b = cw[i] >> 16;
            const u32 sa = start, sb = start + o1s_up4(a);
            if (a) atomicOr(&hbits[sa >> 5], 1u << (sa & 31u));
            if (b) atomicOr(&hbits[sb >> 5], 1u << (sb & 31u));
            if (a >= O1S_BIGMIN) { const u32 k = atomicAdd(&big_[16], 1u); if (k < 16u) big_[k] = a << 8 | sa >> 7; }
            if (b >= O1S_BIGMIN) { const u32 k = atomicAdd(&big_[16], 1u); if (k < 16u) big_[k] = b << 8 | sb >> 7; }
            start = sb + o1s_up4(b);
            ((u32 *)(bins + lane * 136u))[i] = (sa | (a ? 0x8000u : 0u)) | (sb | (b ? 0x8000u : 0u)) << 16;   // bit 15: the chain has not begun (starts are below 21 252)
        }
        const u32 L = (u32)__builtin_amdgcn_readlane((int)incl, 63);
        if (lane == 0u) *(u32 *)(blk + O1S_LEN) = L;
        trc_wave_lds_fence();
        if (lane < 16u) ((u32 *)(blk + O1S_BIG))[lane] = big_[lane];
        // nh[j] = the first chain start at or behind entry 128 j: three units per lane, then a suffix minimum over the lanes
        u32 f[3];
#pragma unroll
        for (u32 q = 0; q < 3u; q++) {
            const u32 j = lane * 3u + q;
            f[q] = L;
            if (j < O1S_SEGS) {
#pragma unroll
                for (int v = 3; v >= 0; v--) { const u32 hb = hbits[4u * j + (u32)v]; if (hb) f[q] = j * 128u + 32u * (u32)v + (u32)__builtin_ctz(hb); }
            }
        }
        f[1] = trc_min(f[1], f[2]); f[0] = trc_min(f[0], f[1]);
        u32 m = f[0];                                           // suffix minimum of the lanes' own minima
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u32 o = (u32)__shfl_down((int)m, d, 64); if (lane + (u32)d < 64u) m = trc_min(m, o); }
        const u32 above = (u32)__shfl_down((int)m, 1, 64);      // ... of the lanes above this one
        const u32 up = lane < 63u ? above : L;
        u16 *const nh = (u16 *)(blk + O1S_NH);
        const u32 n2 = trc_min(f[2], up), n1 = trc_min(f[1], up), n0 = trc_min(f[0], up);
        if (lane * 3u + 0u < O1S_NHN) nh[lane * 3u + 0u] = (u16)n0;
        if (lane * 3u + 1u < O1S_NHN) nh[lane * 3u + 1u] = (u16)n1;
        if (lane * 3u + 2u < O1S_NHN) nh[lane * 3u + 2u] = (u16)n2;
    }
    u32 *const perm = (u32 *)(blk + O1S_PERM);
    // The lanes of this row with the same hi key / the same lo key as mine: the key's bits as ballots, lane order = position order, so a
    // lane's rank among them is a count of lower bits (deterministic, no atomics).  Round 6: a bit's ballot B and the lane's own bit s
    // (0 / -1) give "differs from me in this bit" as B ^ s, the differences are ORed three at a time -- five instructions per bit where
    // the selects between B and ~B took eight -- and a chain's first entry is whoever finds bit 15 of its bin still set (was: two reads
    // of the marks).  (The masks from LDS instead -- every lane ORing its bit into the word of its context and of its hi nibble, one
    // read back -- is SLOWER, 0.80 -> 1.00 ms: same-address 64-bit atomics serialise, and a row's lanes share a handful of contexts.)
    u32 xn = row_load(0), carry = 0;
    for (u32 r = 0; r < R; r++) {
        const u32 pos = r * 64u + lane;
        const bool valid = pos < plen;
        const u32 x = xn;
        if (r + 1u < R) xn = row_load(r + 1u);
        const u32 pv = left_of(x, carry);
        const u64 vm = __ballot(valid);
        u32 dh0 = 0, dh1 = 0, dl0 = 0, dl1 = 0;                 // lanes that differ from this one in some bit of pv / of x >> 4
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const u32 sb = (u32)__builtin_amdgcn_sbfe((int)pv, (u32)b, 1u);
            const u64 B = __ballot(sb != 0u);
            dh0 |= (u32)B ^ sb; dh1 |= (u32)(B >> 32) ^ sb;
        }
#pragma unroll
        for (int b = 4; b < 8; b++) {
            const u32 sb = (u32)__builtin_amdgcn_sbfe((int)x, (u32)b, 1u);
            const u64 B = __ballot(sb != 0u);
            dl0 |= (u32)B ^ sb; dl1 |= (u32)(B >> 32) ^ sb;
        }
        if (valid) {
            const u32 mh0 = (u32)vm & ~dh0, mh1 = (u32)(vm >> 32) & ~dh1, ml0 = mh0 & ~dl0, ml1 = mh1 & ~dl1;
            const u32 kh = pv, kl = 256u + (pv << 4) + (x >> 4);
            u16 *const bh = (u16 *)(bins + kh * 2u), *const bl = (u16 *)(bins + kl * 2u);
            const u32 sh = *bh, sl = *bl;
            const u32 rh = __builtin_amdgcn_mbcnt_hi(mh1, __builtin_amdgcn_mbcnt_lo(mh0, 0u));
            const u32 rl = __builtin_amdgcn_mbcnt_hi(ml1, __builtin_amdgcn_mbcnt_lo(ml0, 0u));
            const u32 nh = (u32)__popc(mh0) + (u32)__popc(mh1), nl = (u32)__popc(ml0) + (u32)__popc(ml1);
            const u32 ih = (sh & 0x7fffu) + rh, il = (sl & 0x7fffu) + rl;
            ent[ih] = pos | (x >> 4) << 12 | (rh == 0u ? (sh >> 15) << 16 : 0u);
            ent[il] = pos | (x & 15u) << 12 | (rl == 0u ? (sl >> 15) << 16 : 0u);
            perm[pos] = ih | il << 16;
            if (rh == nh - 1u) *bh = (u16)((sh & 0x7fffu) + nh);   // the key's last lane moves the bin on (after every lane of the row has read it)
            if (rl == nl - 1u) *bl = (u16)((sl & 0x7fffu) + nl);
        }
    }
}

// Round 6: FOUR waves per workgroup, ONE workgroup per CU (its LDS request is padded past half a CU's), G chunks per workgroup -- a wave has
// its SIMD to itself.  Per-wave clocks (-DTRC_O1W_PROF, profiles/r06_o1_walk_prof.txt) showed what the kernel's time is: its longest
// chain (drift100m: 259 rounds, a chain as long as its chunk) times a wave's time per round, and a wave alone on its SIMD makes a round in
// 3.0 us where two sharing one take 4.1-5.8 each.  G = 96 puts 100 MB (24 414 chunks) on 255 workgroups with 244 rounds of work per lane --
// every wave ends with the longest chain.  The long chains' order of length comes from a counting sort over 80 length classes (the rank
// by comparison of every pair took 27 us of the prologue at G = 32 and 150 at G = 96).
#ifndef O1W_WAVES
#define O1W_WAVES    4u
#endif
#ifndef O1W_GROUP_MAX
#define O1W_GROUP_MAX 96u                                       // chunks per workgroup, at most (the launch picks G: trc_launch_anso1_model)
#endif
#define O1W_NCLS     80u                                        // length classes of the long chains (64 entries wide, longest first)
#define O1W_HIST     (TRC_NIBK_BYTES + 32u)                     // u32[NCLS]
#define O1W_NH       (TRC_NIBK_BYTES + 32u + O1W_NCLS * 4u)
#define O1W_BIG(G)   (O1W_NH + (G) * O1S_NHN * 2u)              // u32[G x 16] as read, then the same by length class
#define O1W_TAKEN(G) (O1W_BIG(G) + 2u * (G) * 64u)              // one bit per unit: on the list already (a long chain's unit)
#define O1W_ULIST(G) (O1W_TAKEN(G) + (G) * 32u)                 // u16[G x (16 + SEGS)]: the units to walk, chunk << 8 | unit (0xffff: none)
#define O1W_STAGE(G) (O1W_ULIST(G) + (G) * (16u + O1S_SEGS) * 2u)
#define O1W_LDS(G)   (O1W_STAGE(G) + O1W_WAVES * 8u * 1024u)
#ifdef TRC_O1W_PROF                                              // variant builds only: wall clocks (100 MHz) and rounds per wave
__device__ unsigned long long trc_o1w_prof[4 * 4096];
#endif
__global__ __launch_bounds__(64 * O1W_WAVES) void trc_o1_walk_kernel(u32 nchunks, u32 G, u8 *__restrict__ model)
{
#ifdef TRC_O1W_PROF
    const u64 pw0 = wall_clock64(); u64 pw1 = 0, prounds = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) u8 smem_wg_[];
    u8 *const kb = smem_wg_;
    u32 *const ctr = (u32 *)(smem_wg_ + TRC_NIBK_BYTES);        // [0] the next list entry, [1] long chains, [4 + w] wave w's share of the other units
    u32 *const hist = (u32 *)(smem_wg_ + O1W_HIST);
    const u16 *const nh = (const u16 *)(smem_wg_ + O1W_NH);
    u32 *const big = (u32 *)(smem_wg_ + O1W_BIG(G)), *const bigs = big + G * 16u;
    u32 *const taken = (u32 *)(smem_wg_ + O1W_TAKEN(G));        // [chunk][8]
    u16 *const ulist = (u16 *)(smem_wg_ + O1W_ULIST(G));
    const u32 wv = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = trc_lane();
    const u32 stage_a = (u32)__builtin_amdgcn_readfirstlane((int)trc_lds_addr(smem_wg_ + O1W_STAGE(G) + wv * 8192u));
    const u32 c0 = blockIdx.x * G;
    const u32 rows = nchunks - c0 < G ? nchunks - c0 : G;
    u8 *const gbase = model + (u64)c0 * O1_MODEL_BYTES;
    if (threadIdx.x < 8u) ctr[threadIdx.x] = 0u;
    for (u32 i = threadIdx.x; i < O1W_NCLS; i += 64u * O1W_WAVES) hist[i] = 0u;
    for (u32 i = threadIdx.x; i < G * O1S_NHN / 8u; i += 64u * O1W_WAVES) {               // the group's unit directory
        const u32 ci = i / (O1S_NHN / 8u), q = i % (O1S_NHN / 8u);
        ((uint4 *)(smem_wg_ + O1W_NH))[i] = ci < rows ? *(const uint4 *)(gbase + (size_t)ci * O1_MODEL_BYTES + O1S_NH + q * 16u)
                                                      : make_uint4(0, 0, 0, 0);
    }
    for (u32 i = threadIdx.x; i < G * 8u; i += 64u * O1W_WAVES) taken[i] = 0u;
    o1_init_k(kb);                                              // (every wave writes the same K; ends with the barrier)
    // its long chains (length << 16 | chunk << 8 | unit), longest first: counted by length class, scanned, placed
    auto cls_of = [](u32 v) -> u32 { const u32 q = v >> 22; return O1W_NCLS - 1u - (q < O1W_NCLS ? q : O1W_NCLS - 1u); };
    for (u32 i = threadIdx.x; i < G * 16u; i += 64u * O1W_WAVES) {
        const u32 ci = i >> 4;
        const u32 v = ci < rows ? ((const u32 *)(gbase + (size_t)ci * O1_MODEL_BYTES + O1S_BIG))[i & 15u] : 0u;
        const u32 it = v ? (v >> 8) << 16 | ci << 8 | (v & 255u) : 0u;
        big[i] = it;
        if (it) atomicAdd(&hist[cls_of(it)], 1u);
    }
    __syncthreads();
    if (wv == 0u) {                                             // class counts -> class starts (two classes per lane)
        const u32 h0 = lane < O1W_NCLS / 2u ? hist[2u * lane] : 0u, h1 = lane < O1W_NCLS / 2u ? hist[2u * lane + 1u] : 0u;
        const u32 incl = trc_wave_incl_scan(h0 + h1);
        if (lane < O1W_NCLS / 2u) { hist[2u * lane] = incl - h0 - h1; hist[2u * lane + 1u] = incl - h1; }
        if (lane == 63u) ctr[1] = incl;
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < G * 16u; i += 64u * O1W_WAVES) {
        const u32 it = big[i];
        if (it) bigs[atomicAdd(&hist[cls_of(it)], 1u)] = it;
    }
    __syncthreads();
    // The list of units to walk (round 6; until then every lane asked for "the next unit" in each of a round's four slots and learnt by
    // three dependent LDS round trips whether a chain started in it and nobody had it -- and half of a chunk's 168 units lie behind the
    // end of its stream): the long chains' units in order of length (a unit two long chains start in: once), then every other unit a
    // chain starts in, every wave compacting its share of them (counted first: no atomics).
    const u32 nbig = ctr[1];
    for (u32 i = threadIdx.x; i < nbig; i += 64u * O1W_WAVES) {
        const u32 it = bigs[i], ci = (it >> 8) & 255u, j = it & 255u, bit = 1u << (j & 31u);
        ulist[i] = (atomicOr(&taken[ci * 8u + (j >> 5)], bit) & bit) ? (u16)0xffffu : (u16)(it & 0xffffu);
    }
    __syncthreads();
    {
        const u32 U = G * O1S_SEGS, per = ((U + 64u * O1W_WAVES - 1u) / (64u * O1W_WAVES)) * 64u;   // units per wave, in rows of 64
        const u32 u0 = wv * per;
        auto is_put = [&](u32 i) -> bool {
            const u32 ci = i / O1S_SEGS, j = i % O1S_SEGS;
            return i < U && ci < rows && nh[ci * O1S_NHN + j] < j * 128u + 128u && !((taken[ci * 8u + (j >> 5)] >> (j & 31u)) & 1u);
        };
        u32 cnt = 0;
        for (u32 i0 = 0; i0 < per; i0 += 64u) cnt += (u32)__popcll(__ballot(is_put(u0 + i0 + lane)));
        if (lane == 0u) ctr[4u + wv] = cnt;
        __syncthreads();
        u32 at = nbig;
        for (u32 k = 0; k < wv; k++) at += ctr[4u + k];
        for (u32 i0 = 0; i0 < per; i0 += 64u) {
            const u32 i = u0 + i0 + lane;
            const bool put = is_put(i);
            const u64 m = __ballot(put);
            if (put) ulist[at + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u))] = (u16)((i / O1S_SEGS) << 8 | (i % O1S_SEGS));
            at += (u32)__popcll(m);
        }
    }
    __syncthreads();
    u32 nunits = nbig;
    for (u32 k = 0; k < O1W_WAVES; k++) nunits += ctr[4u + k];

    NibTable T = O1Cache::fresh();
    bool active = false, done = false, spare = false;
    u32 widx = 0, uend = 0;                                     // the window to ask for next, the end of the unit
    u32 sp_ci = 0, sp_us = 0, sp_ue = 0;                        // the unit in reserve
    u8 *sbase = gbase;
    uint4 rec[4];
    u8 *recp[4], *reqp[4];                                      // where the records of the windows walked / asked for in this round go
    bool recv[4], reqv[4];                                      // ... if anywhere (flags, not null pointers: the stores stay global_store)
#pragma unroll
    for (u32 w = 0; w < 4u; w++) { recp[w] = gbase; reqp[w] = gbase; recv[w] = false; reqv[w] = false; rec[w] = make_uint4(0, 0, 0, 0); }
    u32 buf = 0;
    bool more = true;
#ifdef TRC_O1W_PROF
    pw1 = wall_clock64();
#endif
    while (more) {
#ifdef TRC_O1W_PROF
        prounds++;
#endif
        // A round = four windows per lane.  Everything asked for a round ago -- the windows of this round, the record stores of the
        // last -- is a round old here: the one wait of the loop finds it done.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // This round's windows (asked for a round ago) are read here, and the K rows (K[sym], 32 bytes per entry) of a window are asked for
        // a window ahead: a wave alone on its SIMD has nobody to hide its LDS round trips behind (0.79 -> 0.73 ms).
        uint4 W[4];
#pragma unroll
        for (u32 w = 0; w < 4u; w++) W[w] = trc_ldsr128(stage_a + buf * 4096u + w * 1024u + lane * 16u);
        const u32 kb_a = trc_lds_addr(kb);
        uint4 Ka[2][4], Kb[2][4];
        auto ask = [&](u32 w, u32 b) __attribute__((always_inline)) {
            const u32 e[4] = { W[w].x, W[w].y, W[w].z, W[w].w };
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const u32 a = kb_a + ((e[i] >> 7) & 0x1e0u);       // 32 x the symbol (bits 12..15)
                Ka[b][i] = trc_ldsr128(a); Kb[b][i] = trc_ldsr128(a + 16u);
            }
        };
        ask(0, 0);
        {   // The four windows of a lane are usually 64 consecutive bytes.  Stored lane by lane they are 16-byte pieces of 64 different
            // lines per instruction, and a piece costs the memory system what a whole sector does (this kernel: 1.43 ms, 0.99 without
            // its stores).  Transposed across the quad -- records AND addresses -- instruction j has the quad's four lanes write the
            // four windows of lane 4i + j: one 64-byte request where they are consecutive, still right where they are not.
            u32 m[4][4], ad[4][4];
#pragma unroll
            for (int w = 0; w < 4; w++) {
                m[w][0] = rec[w].x; m[w][1] = rec[w].y; m[w][2] = rec[w].z; m[w][3] = rec[w].w;
                ad[w][0] = (u32)(uintptr_t)recp[w]; ad[w][1] = (u32)((uintptr_t)recp[w] >> 32); ad[w][2] = recv[w] ? 1u : 0u; ad[w][3] = 0u;
            }
            trc_quad_transpose(m); trc_quad_transpose(ad);
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (ad[j][2]) trc_gst128((void *)((uintptr_t)ad[j][1] << 32 | ad[j][0]), make_uint4(m[j][0], m[j][1], m[j][2], m[j][3]));
        }
#pragma unroll
        for (u32 w = 0; w < 4u; w++) { recp[w] = reqp[w]; recv[w] = reqv[w]; }
        const u32 wr = stage_a + (buf ^ 1u) * 4096u;
        buf ^= 1u;
        if (!spare && !done && (!active || uend - widx <= 32u)) {   // a unit in reserve when this one is within two rounds of its end (not earlier: whoever is free first takes the next longest)
            const u32 g = atomicAdd(&ctr[0], 1u);
            if (g >= nunits) done = true;
            else {
                const u32 u = ulist[g];
                if (u != 0xffffu) {
                    sp_ci = u >> 8;
                    sp_us = nh[sp_ci * O1S_NHN + (u & 255u)]; sp_ue = nh[sp_ci * O1S_NHN + (u & 255u) + 1u];
                    spare = true;
                }
            }
        }
#pragma unroll
        for (u32 w = 0; w < 4u; w++) {                          // ask for the next round's windows
            if (active && widx >= uend) active = false;
            if (!active && spare) {                             // on to the unit in reserve
                sbase = gbase + (size_t)sp_ci * O1_MODEL_BYTES + O1S_STREAM;
                widx = sp_us; uend = sp_ue; active = true; spare = false;
            }
            // window k of a stream only in slot k mod 4 of a round: a lane's four windows of a round are then ONE 64-byte sector
            // (a unit that starts elsewhere idles up to three slots first)
            const bool go = active && ((widx >> 2) & 3u) == w;
            u8 *const src = go ? sbase + (size_t)widx * 4u : gbase;
            reqp[w] = src; reqv[w] = go;
            const u32 row = (u32)__builtin_amdgcn_readfirstlane((int)(wr + w * 1024u));
            u32 keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(row) : "memory");
            if (go) widx += 4u;
        }
        {   // walk this round's windows
#pragma unroll
            for (u32 w = 0; w < 4u; w++) {
                if (w + 1u < 4u) ask(w + 1u, (w + 1u) & 1u);
                const u32 e[4] = { W[w].x, W[w].y, W[w].z, W[w].w };
                u32 r[4];
                if (e[0] & O1S_HEAD) T = O1Cache::fresh();      // (chains start on windows; behind a chain's last entry the table is garbage until then: nobody reads it)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const u32 sym = (e[i] >> 12) & 15u;
                    u32 a0, a1;
                    o1_bounds(T, sym, a0, a1);
                    const uint4 ka = Ka[w & 1u][i], kc = Kb[w & 1u][i];
                    const u32 K[8] = { ka.x, ka.y, ka.z, ka.w, kc.x, kc.y, kc.z, kc.w };
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const trc_s2 v = trc_as_s2(T.d[k]);
                        T.d[k] = trc_as_u32(v + ((trc_as_s2(K[k]) - v) >> (trc_s2)7));
                    }
                    r[i] = (a0 << TRC_PROB_BITS) | (a1 - a0);
                }
                rec[w] = make_uint4(r[0], r[1], r[2], r[3]);
            }
        }
        bool pend = !done || active || spare;
#pragma unroll
        for (u32 w = 0; w < 4u; w++) pend = pend || recv[w] || reqv[w];
        more = __ballot(pend) != 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // (the last round's requests write LDS: not behind the wave's end)
#ifdef TRC_O1W_PROF
    if (lane == 0u) { const u32 wid = (blockIdx.x * O1W_WAVES + wv) & 4095u; trc_o1w_prof[4 * wid] = pw0; trc_o1w_prof[4 * wid + 1] = pw1; trc_o1w_prof[4 * wid + 2] = wall_clock64(); trc_o1w_prof[4 * wid + 3] = prounds; }
#endif
}

#define O1P_SLOTS    9216u
__global__ __launch_bounds__(256) void trc_o1_place_kernel(u64 n, u32 chunk, u32 nchunks, const u8 *__restrict__ model, u8 *__restrict__ recs)
{
    __shared__ __attribute__((aligned(16))) u32 sr[O1P_SLOTS];
    const u32 c = blockIdx.x, t = threadIdx.x;
    const u32 lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    const u32 len = c == nchunks - 1u ? lastlen : chunk, plen = len + (len & 1u);
    const u8 *blk = model + (u64)c * O1_MODEL_BYTES;
    const u32 L = *(const u32 *)(blk + O1S_LEN);
    const u32 *st = (const u32 *)(blk + O1S_STREAM);
    const bool staged = L <= O1P_SLOTS;                         // (many short chains -- incompressible input -- leave more unused slots: gather from memory then)
    u8 *const rb = recs + (u64)c * (8u * (u64)chunk);
    const u32 *perm = (const u32 *)(blk + O1S_PERM);
    // (late round 5) a thread's sixteen `perm` words are asked for at once, together with the staging loads: as a loop of load -> two
    // LDS reads -> two stores the workgroup walked sixteen dependent round trips, four workgroups per CU, 24 rounds of them
    u32 pmv[16];
#pragma unroll
    for (u32 k = 0; k < 16u; k++) { const u32 p = t + 256u * k; pmv[k] = p < plen ? perm[p] : 0u; }
    if (staged) for (u32 i = t * 4u; i < L; i += 1024u) *(uint4 *)(sr + i) = *(const uint4 *)(st + i);
    __syncthreads();
#pragma unroll
    for (u32 k = 0; k < 16u; k++) {
        const u32 p = t + 256u * k;
        if (p >= plen) break;
        const u32 pm = pmv[k];
        u8 *dst = rb + (p >> 4) * 128u + (p & 15u) * 4u;
        u32 rh, rl;
        if (staged) { rh = sr[pm & 0xffffu]; rl = sr[pm >> 16]; }
        else { rh = trc_gld32(st + (pm & 0xffffu)); rl = trc_gld32(st + (pm >> 16)); }
        *(u32 *)dst = rh;
        *(u32 *)(dst + 64u) = rl;
    }
}

// ------------------------------------------------------------------------------------- decode ---
// Round 5: R chunks per wave (lanes 0 .. R - 1; 64 / R waves per group of 64 chunks).  These kernels are chains of table round trips:
// a wave makes one per nibble as long as ANY of its lanes needs a table it does not hold, so with 64 chunks per wave (some lane always
// misses) a 4096-byte chunk is 8192 round trips of ~0.8 us, and at 100 MB / 4096 there are 382 such waves -- 0.37 per SIMD, the chip
// idle.  With R = 16 there are four times the waves: 5.88 -> 5.36 ms, NOT the factor the idle SIMDs promised -- a wave of 16 lanes
// still has some lane missing at almost every nibble (a lane misses on 61-70 % of them), so every wave still walks the same
// chain of two dependent ~0.7 us round trips per byte; only its other costs shrink.  R = 8: 6.8 ms (every wave-instruction then
// serves 8 lanes: issue-bound).  profiles/r05_notes.md.
template <u32 R>
__global__ __launch_bounds__(64) void trc_o1_dec_kernel(
    const u8 *__restrict__ payload, const u32 *__restrict__ clen, const u64 *__restrict__ goff, const u32 *__restrict__ gsum,
    u64 n, u32 chunk, u32 nchunks, u8 *__restrict__ model, u8 *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) u8 kb[TRC_NIBK_BYTES];
    __shared__ __attribute__((aligned(16))) u8 seen[O1_SEEN_BYTES];
    const u32 lane = threadIdx.x;
    o1_init_k(kb);

    constexpr u32 WPG = 64u / R;                               // waves per group of 64 chunks
    const u32 g0 = (blockIdx.x / WPG) * 64u, wq = blockIdx.x % WPG;
    WaveChunks wc;
    wc.c0 = g0 + wq * R; wc.chunk = chunk; wc.nchunks = nchunks;
    wc.lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    wc.rows = wc.c0 >= nchunks ? 0u : nchunks - wc.c0 < R ? nchunks - wc.c0 : R;
    // the directory of the whole group (payload offsets are per group of 64 chunks): lane L reads chunk g0 + L, the wave's own R
    // chunks take their numbers from lanes wq R ..
    const u32 cg = g0 + lane;
    const u32 lenL = cg < nchunks ? ((cg == nchunks - 1u) ? wc.lastlen : chunk) : 0u;
    const u32 clL = cg < nchunks ? trc_min(clen[cg], lenL) : 0u;   // a directory entry above the chunk length (corrupt input) reads as raw
    const u32 exL = trc_wave_incl_scan(clL) - clL;
    const u32 srcl = (wq * R + lane) & 63u;
    const u32 cl_s = (u32)__shfl((int)clL, (int)srcl, 64), ex_s = (u32)__shfl((int)exL, (int)srcl, 64), len_s = (u32)__shfl((int)lenL, (int)srcl, 64);
    const bool alive = lane < wc.rows;
    const u32 c = wc.c0 + lane;
    const u32 len = alive ? len_s : 0u;
    const u32 cl = alive ? cl_s : 0u;
    const u32 ex = ex_s;
    const u64 off = trc_group_base(goff, gsum, g0 >> 6) + ex;
    const bool coded = alive && cl != len;
    O1Cache tc;
    tc.init(model + (u64)(wc.c0 + (alive ? lane : 0u)) * O1_MODEL_BYTES, seen);

    u32 st[4] = { TRC_ANS_LOW, TRC_ANS_LOW, TRC_ANS_LOW, TRC_ANS_LOW };
    if (coded) for (u32 k = 0; k < 4; k++) st[k] = trc_ld32_a2(payload + off + 4u * k);   // decoder st[i] = encoder st[3-i] (mnfill)
    LaneIn<2> si; si.prime(payload + off + 16u, coded, trc_sub_sat(cl, 16u));
    u32 cx = 0;

    // cdf16ansdec on a cached table (only lanes with act touch memory, and only for a table they do not hold)
    auto get_nibble = [&](u32 &s, NibTable &T) -> u32 {
        const u32 slot = s & (TRC_PROB_ONE - 1);
        u32 c0, c1;
        const u32 x = trc_nib_find<true>(T, slot, c0, c1);    // (bit-select form: trc_nibmodel.h)
        s = __umul24(c1 - c0, s >> TRC_PROB_BITS) + slot - c0;
        o1_adapt(T, kb, x);
        return x;
    };
    auto get_byte = [&](bool act, u32 &sh, u32 &sl) -> u32 {    // context cx -> byte, which becomes the context
        tc.need_hi(act, cx);
        u32 h = 0, l = 0;
        if (act) h = get_nibble(sh, tc.H);
        tc.need_lo(act, cx, h);
        if (act) { l = get_nibble(sl, tc.L); cx = h << 4 | l; }
        return h << 4 | l;
    };
    auto renorm = [&](u32 &s, bool act) {
        const u32 w = si.peek16();
        const bool rn = act && s < TRC_ANS_LOW;
        s = rn ? (s << 16) | w : s;
        si.skip_if(rn);
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
#pragma unroll
                    for (int j = 0; j < 2; j++) {              // mndec8x2x: two bytes, then four renorms in order st0..st3
                        const bool act = coded && q0 + 2u * (u32)j < len;     // the second byte of an odd tail is the dummy
                        {
                            const u32 x0 = get_byte(act, st[0], st[1]);
                            const u32 x1 = get_byte(act, st[2], st[3]);
                            w |= (x0 | x1 << 8) << (16 * j);
                        }
                        renorm(st[0], act); renorm(st[1], act); renorm(st[2], act); renorm(st[3], act);
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

// ---------------------------------------------------------------------------- decode, by rows ---
// Round 5 (late): EIGHT LANES PER CHUNK.  What the one-lane-per-chunk decoder above spends (timing ablation, 100 MB at chunk 4096,
// profiles/r05_notes.md): 2.3-2.9 ms on its own instruction chain -- a nibble's search, select trees and 8-register update are ~100
// instructions of a lone wave -- and 2.4-3.4 ms on table round trips, every one of them 64 scattered 16-byte lane accesses through the
// texture-address unit.  Here a CDF16 table is spread over the eight lanes of a ROW (lane e holds entries 2e | 2e+1 as one packed
// dword -- the memory format is unchanged, lane e moves dword e), a wave decodes eight chunks, and everything a chunk's decoder
// keeps (states, stream position, context, table identities) is held redundantly by the row's lanes:
//   * a table move is ONE coalesced 32-byte access per row (four bytes per lane);
//   * the symbol: t = T - (slot + 1) in both halves (v_pk_sub_u16; entries and slots are below 2^15, so a half's sign bit says
//     "entry <= slot"), two ballots, a popcount of the row's eight bits each: x = count - 1 (the entries are increasing);
//   * the bounds: the dword holding (t[x], t[x+1]) is T of lane x / 2 when x is even, (T.hi of that lane, T.lo of the next) when
//     odd -- every lane prepares both forms (a DPP row shift, entry 16 = 2^15 behind lane 7), picks by x's parity (x is
//     row-uniform) and ONE ds_bpermute fetches it;
//   * the update: K[x][i] = 10 i + (i > x ? 32736 : 0), and "i > x" is the complement of the search's sign bits; three packed
//     16-bit operations, as cdf16upd does (cdf_.h, the wrapping arithmetic of trc_nibmodel.h).
// ~50 instructions per nibble for eight chunks (107 per byte with the renormalisations and the output).  First touch without the hi table's entry 0: 256 "context seen" bits and 256 x 16
// "lo table seen" bits per chunk in LDS (544 B per chunk, 4.25 KiB per wave).
#define O1R_LANES 8u
#define O1R_CHUNKS (64u / O1R_LANES)                          // chunks per wave
#define O1R_SEEN_BYTES 544u                                   // u16 lo-seen[256], u32 hi-seen[8]
__global__ __launch_bounds__(64) void trc_o1_dec_rows_kernel(
    const u8 *__restrict__ payload, const u32 *__restrict__ clen, const u64 *__restrict__ goff, const u32 *__restrict__ gsum,
    u64 n, u32 chunk, u32 nchunks, u8 *__restrict__ model, u8 *__restrict__ out)
{
    typedef __attribute__((address_space(3))) u32 lds_u32;
    typedef __attribute__((address_space(3))) u16 lds_u16;
    __shared__ __attribute__((aligned(16))) u8 seen_s[O1R_CHUNKS * O1R_SEEN_BYTES];
    const u32 lane = threadIdx.x, e = lane & (O1R_LANES - 1u), row = lane / O1R_LANES;
    for (u32 i = lane; i < O1R_CHUNKS * O1R_SEEN_BYTES / 4u; i += 64u) ((u32 *)seen_s)[i] = 0u;
    __syncthreads();

    constexpr u32 WPG = 64u / O1R_CHUNKS;                      // waves per group of 64 chunks
    const u32 g0 = (blockIdx.x / WPG) * 64u, wq = blockIdx.x % WPG;
    const u32 lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    // the directory of the whole group (payload offsets are per group of 64 chunks): lane L reads chunk g0 + L
    const u32 cg = g0 + lane;
    const u32 lenL = cg < nchunks ? ((cg == nchunks - 1u) ? lastlen : chunk) : 0u;
    const u32 clL = cg < nchunks ? trc_min(clen[cg], lenL) : 0u;   // a directory entry above the chunk length (corrupt input) reads as raw
    const u32 exL = trc_wave_incl_scan(clL) - clL;
    const u64 gbase = trc_group_base(goff, gsum, g0 >> 6);
    // lanes 0 .. 7 as "chunk wq * 8 + lane of the group" (the raw copy at the end), then every lane as its row's chunk
    const u32 srcl = (wq * O1R_CHUNKS + lane) & 63u;
    const u32 cl_w = (u32)__shfl((int)clL, (int)srcl, 64), ex_w = (u32)__shfl((int)exL, (int)srcl, 64), len_w = (u32)__shfl((int)lenL, (int)srcl, 64);
    const u32 cl = (u32)__shfl((int)cl_w, (int)row, 64), ex = (u32)__shfl((int)ex_w, (int)row, 64), len = (u32)__shfl((int)len_w, (int)row, 64);
    const u32 c0w = g0 + wq * O1R_CHUNKS, c = c0w + row;
    const bool alive = c < nchunks;
    const u64 off = gbase + ex;
    const bool coded = alive && cl != len;

    // this lane's dword of table `id` of the row's chunk: mbase[moff + 32 id] (a wave-uniform base and a 32-bit offset: one
    // address instruction per access).  Rows past the last chunk compute offsets beyond the model space and never use them.
    u8 *const mbase = model + (u64)(c0w < nchunks ? c0w : 0u) * O1_MODEL_BYTES;
    const u32 moff = row * O1_MODEL_BYTES + e * 4u;
#ifndef TRC_O1_SWIZZLE
#define TRC_O1_SWIZZLE 1
#endif
    // Table `id` lies in slot id ^ (chunk & 127) of the chunk's block (4352 = 34 x 128 slots: the low seven bits permute inside their
    // 128).  The blocks are 34 x 4096 bytes apart, so without it table `id` of EVERY chunk sits at the same address modulo 4096 -- the
    // same L2 channel -- and the tables of the few contexts most bytes follow are hit by all the rows of the chip at once.
    const u32 swz = TRC_O1_SWIZZLE ? (c & 127u) << 5 : 0u;     // (as a byte offset)
    const u32 seen = trc_lds_addr(seen_s) + row * O1R_SEEN_BYTES;
    const u32 fresh = trc_pk((2u * e) << 11, (2u * e + 1u) << 11);
    const u32 kbase1 = trc_pk(20u * e, 20u * e + 10u) + 0x7fe07fe0u;   // K of a lane none of whose entries is <= slot; every such entry takes 32736 off
    const u32 rowb4 = (lane & 56u) << 2;
    const bool last_lane = e == O1R_LANES - 1u;
    // The row starts out HOLDING the hi table of context 0 and the lo table (0, 0), both fresh, both marked "seen": the first byte's
    // context is 0, so H is right as it stands, and L is either right (first hi nibble 0) or goes to memory untouched -- a swap never
    // has to ask whether there is a table to put back.
    u32 H = fresh, L = fresh;
    u32 hoff = (0u ^ swz) + moff, loff = (32u ^ swz) + moff;   // where the row's current tables live: table id at ((32 id) ^ swz) + moff
    if (e == 0u) { *(lds_u32 *)(uintptr_t)(seen + 512u) = 1u; *(lds_u16 *)(uintptr_t)seen = (u16)1u; }
    trc_wave_lds_fence();

    u32 st0 = TRC_ANS_LOW, st1 = TRC_ANS_LOW, st2 = TRC_ANS_LOW, st3 = TRC_ANS_LOW;
    if (coded) {                                               // decoder st[i] = encoder st[3-i] (mnfill)
        st0 = trc_ld32_a2(payload + off); st1 = trc_ld32_a2(payload + off + 4u);
        st2 = trc_ld32_a2(payload + off + 8u); st3 = trc_ld32_a2(payload + off + 12u);
    }
    const u8 *const sbase = payload + gbase;                   // the words follow the four states: sbase[soff + position]
    const u32 soff = ex + 16u;
    const u32 lim = trc_sub_sat(cl, 16u);
    const u32 lenc = coded ? len : 0u;
    u32 rpos = 0, cx = 0;

    // cdf16ansdec on the row's table T (anscdf_.h:164-174) in two halves: the symbol; then the state update and the table update.
    // Round 6: between the two the NEXT table's dword is asked for -- the lo table's address exists as soon as the hi nibble does, the next
    // byte's hi table's as soon as the lo nibble does -- so its round trip runs under the ~25 instructions of the update instead of
    // after them (the timing ablation of round 5 put 0.8 of the 3.6 ms on exposed round trips).  No table is stored between the early
    // load and the swap that uses it, and a load of a table that turns out not to be needed is dropped.
    struct Nib { u32 slot, f, cc; };
    auto find = [&](u32 s, u32 T, Nib &q) -> u32 {
        q.slot = s & (TRC_PROB_ONE - 1u);
        const u32 t = trc_as_u32(trc_as_s2(T) - trc_as_s2(__umul24(q.slot, 0x10001u) + 0x10001u));
        q.f = (t >> 15) & 0x10001u;                            // per half: entry <= slot
        u32 cnt = (u32)__popc(q.f);                            // the row's sum, in every lane: xor 1, xor 2, mirror of the eight
        cnt += (u32)__builtin_amdgcn_update_dpp(0, (int)cnt, 0xB1, 0xf, 0xf, true);      // quad_perm:[1,0,3,2]
        cnt += (u32)__builtin_amdgcn_update_dpp(0, (int)cnt, 0x4E, 0xf, 0xf, true);      // quad_perm:[2,3,0,1]
        cnt += (u32)__builtin_amdgcn_update_dpp(0, (int)cnt, 0x141, 0xf, 0xf, true);     // row_half_mirror
        const u32 x = cnt - 1u;
        u32 N = (u32)__builtin_amdgcn_update_dpp(0, (int)T, 0x101, 0xf, 0xf, true);      // row_shl:1 -- lane i takes lane i + 1
        N = last_lane ? TRC_PROB_ONE : N;
        const u32 V = __builtin_amdgcn_alignbit(N, T, (x << 4) & 16u);       // x even: T; odd: T.hi | N.lo << 16
        q.cc = (u32)__builtin_amdgcn_ds_bpermute((int)(((x << 1) & 0x1cu) | rowb4), (int)V);
        return x;
    };
    auto finish = [&](u32 &s, u32 &T, const Nib &q) {
        const u32 c0 = q.cc & 0xffffu;
        s = __umul24((q.cc >> 16) - c0, s >> TRC_PROB_BITS) + q.slot - c0;
        const u32 K = (u32)(__mul24((int)q.f, -32736) + (int)kbase1);
        const trc_s2 d = (trc_as_s2(K) - trc_as_s2(T)) >> (trc_s2)7;
        T = trc_as_u32(trc_as_s2(T) + d);
    };
    u32 pendH = 0;                                             // the dword of the next byte's hi table, asked for at the end of this byte
    auto get_byte = [&](bool act, u32 &sh, u32 &sl) -> u32 {    // context cx -> byte, which becomes the context
        const u32 cb = (cx << 9) + (cx << 5);                 // 32 x the context's first table (17 tables per context: 544 bytes)
        {
            const u32 noff = (cb ^ swz) + moff;
            if (act && noff != hoff) {                         // the row's hi table goes back to memory, the context's comes in
                *(u32 *)(mbase + hoff) = H;
                const u32 a = seen + 512u + ((cx >> 5) << 2), bit = 1u << (cx & 31u);
                const u32 bits = *(const lds_u32 *)(uintptr_t)a;
                *(lds_u32 *)(uintptr_t)a = bits | bit;         // (every lane of the row writes the same word)
                H = (bits & bit) ? pendH : fresh;              // (a table never written: whatever was loaded, dropped)
                hoff = noff;
            }
        }
        Nib qh, ql;
        const u32 h = find(sh, H, qh) & 15u;
        const u32 noffl = ((cb + 32u + (h << 5)) ^ swz) + moff;
        const bool needl = act && noffl != loff;
        u32 pendL = 0;
        if (needl) pendL = *(const u32 *)(mbase + noffl);
        finish(sh, H, qh);
        if (needl) {
            *(u32 *)(mbase + loff) = L;
            const u32 a = seen + cx * 2u, bit = 1u << h;
            const u32 bits = *(const lds_u16 *)(uintptr_t)a;
            *(lds_u16 *)(uintptr_t)a = (u16)(bits | bit);
            L = (bits & bit) ? pendL : fresh;
            loff = noffl;
        }
        const u32 l = find(sl, L, ql) & 15u;
        const u32 b = h << 4 | l;
        cx = act ? b : cx;
        {
            const u32 nh = (((cx << 9) + (cx << 5)) ^ swz) + moff;
            if (nh != hoff) pendH = *(const u32 *)(mbase + nh);
        }
        finish(sl, L, ql);
        return b;
    };

    struct __attribute__((packed, aligned(2))) W2 { u32 lo, hi; };
    u8 *const dst = out + (u64)(alive ? c : 0u) * chunk;
    for (u32 p0 = 0; p0 < chunk; p0 += 32u) {                  // 32 output bytes per trip: lane e keeps dword e of them
        if (!__ballot(p0 < lenc)) break;
        u32 acc = 0;
#pragma nounroll
        for (u32 d = 0; d < 8u; d++) {
            u32 w = 0;
#pragma unroll
            for (int j = 0; j < 2; j++) {                      // mndec8x2x: two bytes, then four renorms in order st0..st3
                const bool act = p0 + 4u * d + 2u * (u32)j < lenc;             // the second byte of an odd tail is the dummy
                const W2 w2 = *(const W2 *)(sbase + (soff + trc_min(rpos, lim)));   // the (up to) four words this pair's renorms take, requested now
                const u32 x0 = get_byte(act, st0, st1);
                const u32 x1 = get_byte(act, st2, st3);
                w |= (x0 | x1 << 8) << (16 * j);
                const u64 ww = ((u64)w2.hi << 32) | w2.lo;
                u32 taken = 0;                                 // bits of ww used up
                auto renorm = [&](u32 &s) {                    // (rows that are done or raw run on garbage: nothing of theirs is used)
                    const bool rn = s < TRC_ANS_LOW;
                    s = rn ? __builtin_amdgcn_perm(s, (u32)(ww >> taken), 0x05040100u) : s;      // (s << 16) | next word
                    taken += rn ? 16u : 0u;
                };
                renorm(st0); renorm(st1); renorm(st2); renorm(st3);
                rpos += act ? taken >> 3 : 0u;
            }
            acc = e == d ? w : acc;
        }
        const u32 pos = p0 + 4u * e;
        if (pos + 4u <= lenc) *(u32 *)(dst + pos) = acc;
        else if (pos < lenc)                                    // ragged end of the last chunk
            for (u32 k = 0; pos + k < lenc; k++) dst[pos + k] = (u8)(acc >> (8u * k));
    }
    trc_wave_copy_raw(__ballot(lane < O1R_CHUNKS && c0w + lane < nchunks && cl_w == len_w && len_w != 0u), gbase + ex_w, len_w,
                      out + (u64)c0w * chunk, chunk, payload);
}

// --------------------------------------------------------------- decode, FOUR or TWO lanes per chunk ---
// Round 6.  The rows form above with a CDF16 table over LANES = 4 or 2 lanes (lane e = entries 2D e .. 2D e + 2D - 1 as D = 8 / LANES packed
// dwords: a table move is one 4D-byte access per lane, 32 bytes per row as before) and 64 / LANES chunks per wave.  Two findings behind it
// (profiles/r06_notes.md): the eight-lane kernel is bound by what its SIMDs can issue (97 instructions per byte for eight chunks, three waves
// per SIMD) -- this one issues ~115 for sixteen or ~150 for thirty-two -- and a wave's chain runs a third faster with its SIMD to itself (decode
// of 4096-byte chunks: 2.2 ms alone, 2.9-3.0 sharing), so the launch picks the widest form that still puts at most one wave on a SIMD.
// The row's sum is log2(LANES) DPP steps; the pair (t[x], t[x+1]) is a window over (T[0] .. T[D-1], the next lane's T[0]) picked by
// x mod 2D in the lane x / 2D.
template <u32 LANES>
__global__ __launch_bounds__(64) void trc_o1_dec_rowsn_kernel(
    const u8 *__restrict__ payload, const u32 *__restrict__ clen, const u64 *__restrict__ goff, const u32 *__restrict__ gsum,
    u64 n, u32 chunk, u32 nchunks, u8 *__restrict__ model, u8 *__restrict__ out)
{
    typedef __attribute__((address_space(3))) u32 lds_u32;
    typedef __attribute__((address_space(3))) u16 lds_u16;
    constexpr u32 D = 8u / LANES, CHUNKS = 64u / LANES;        // dwords of a table per lane; chunks per wave
    static_assert(LANES == 2u || LANES == 4u, "two or four lanes per chunk");
    struct Tab { u32 d[D]; };
    __shared__ __attribute__((aligned(16))) u8 seen_s[CHUNKS * O1R_SEEN_BYTES];
    const u32 lane = threadIdx.x, e = lane & (LANES - 1u), row = lane / LANES;
    for (u32 i = lane; i < CHUNKS * O1R_SEEN_BYTES / 4u; i += 64u) ((u32 *)seen_s)[i] = 0u;
    __syncthreads();

    constexpr u32 WPG = 64u / CHUNKS;                          // waves per group of 64 chunks
    const u32 g0 = (blockIdx.x / WPG) * 64u, wq = blockIdx.x % WPG;
    const u32 lastlen = (u32)(n - (u64)(nchunks - 1) * chunk);
    const u32 cg = g0 + lane;                                  // the directory of the whole group: lane L reads chunk g0 + L
    const u32 lenL = cg < nchunks ? ((cg == nchunks - 1u) ? lastlen : chunk) : 0u;
    const u32 clL = cg < nchunks ? trc_min(clen[cg], lenL) : 0u;   // a directory entry above the chunk length (corrupt input) reads as raw
    const u32 exL = trc_wave_incl_scan(clL) - clL;
    const u64 gbase = trc_group_base(goff, gsum, g0 >> 6);
    // lanes 0 .. CHUNKS - 1 as "chunk wq * CHUNKS + lane of the group" (the raw copy at the end), then every lane as its row's chunk
    const u32 srcl = (wq * CHUNKS + lane) & 63u;
    const u32 cl_w = (u32)__shfl((int)clL, (int)srcl, 64), ex_w = (u32)__shfl((int)exL, (int)srcl, 64), len_w = (u32)__shfl((int)lenL, (int)srcl, 64);
    const u32 cl = (u32)__shfl((int)cl_w, (int)row, 64), ex = (u32)__shfl((int)ex_w, (int)row, 64), len = (u32)__shfl((int)len_w, (int)row, 64);
    const u32 c0w = g0 + wq * CHUNKS, c = c0w + row;
    const bool alive = c < nchunks;
    const u64 off = gbase + ex;
    const bool coded = alive && cl != len;

    u8 *const mbase = model + (u64)(c0w < nchunks ? c0w : 0u) * O1_MODEL_BYTES;
    const u32 moff = row * O1_MODEL_BYTES + e * (4u * D);      // this lane's dwords of table `id`: mbase[((32 id) ^ swz) + moff]
    const u32 swz = TRC_O1_SWIZZLE ? (c & 127u) << 5 : 0u;
    const u32 seen = trc_lds_addr(seen_s) + row * O1R_SEEN_BYTES;
    Tab fresh, kbase;
#pragma unroll
    for (u32 k = 0; k < D; k++) {
        const u32 i0 = 2u * D * e + 2u * k;
        fresh.d[k] = trc_pk(i0 << 11, (i0 + 1u) << 11);
        kbase.d[k] = trc_pk(10u * i0, 10u * i0 + 10u) + 0x7fe07fe0u;   // K of a dword none of whose entries is <= slot; every such entry takes 32736 off
    }
    const u32 rowb4 = (lane & ~(LANES - 1u)) << 2;
    const bool last_lane = e == LANES - 1u;
    auto ld_tab = [&](u32 o) -> Tab {
        Tab t;
        if constexpr (D == 2u) { const uint2 v = *(const uint2 *)(mbase + o); t.d[0] = v.x; t.d[1] = v.y; }
        else { const uint4 v = *(const uint4 *)(mbase + o); t.d[0] = v.x; t.d[1] = v.y; t.d[2] = v.z; t.d[3] = v.w; }
        return t;
    };
    auto st_tab = [&](u32 o, const Tab &t) {
        if constexpr (D == 2u) *(uint2 *)(mbase + o) = make_uint2(t.d[0], t.d[1]);
        else *(uint4 *)(mbase + o) = make_uint4(t.d[0], t.d[1], t.d[2], t.d[3]);
    };
    // the row starts out HOLDING the hi table of context 0 and the lo table (0, 0), both fresh, both marked "seen" (as above)
    Tab H = fresh, L = fresh;
    u32 hoff = (0u ^ swz) + moff, loff = (32u ^ swz) + moff;
    if (e == 0u) { *(lds_u32 *)(uintptr_t)(seen + 512u) = 1u; *(lds_u16 *)(uintptr_t)seen = (u16)1u; }
    trc_wave_lds_fence();

    u32 st0 = TRC_ANS_LOW, st1 = TRC_ANS_LOW, st2 = TRC_ANS_LOW, st3 = TRC_ANS_LOW;
    if (coded) {                                               // decoder st[i] = encoder st[3-i] (mnfill)
        st0 = trc_ld32_a2(payload + off); st1 = trc_ld32_a2(payload + off + 4u);
        st2 = trc_ld32_a2(payload + off + 8u); st3 = trc_ld32_a2(payload + off + 12u);
    }
    const u8 *const sbase = payload + gbase;                   // the words follow the four states: sbase[soff + position]
    const u32 soff = ex + 16u;
    const u32 lim = trc_sub_sat(cl, 16u);
    const u32 lenc = coded ? len : 0u;
    u32 rpos = 0, cx = 0;

    struct Nib { u32 slot, cc; Tab f; };
    auto find = [&](u32 s, const Tab &T, Nib &q) -> u32 {
        q.slot = s & (TRC_PROB_ONE - 1u);
        const u32 sp = __umul24(q.slot, 0x10001u) + 0x10001u;
        u32 cnt = 0;
#pragma unroll
        for (u32 k = 0; k < D; k++) {
            const u32 t = trc_as_u32(trc_as_s2(T.d[k]) - trc_as_s2(sp));
            q.f.d[k] = (t >> 15) & 0x10001u;                   // per half: entry <= slot
            cnt += (u32)__popc(q.f.d[k]);
        }
        cnt += (u32)__builtin_amdgcn_update_dpp(0, (int)cnt, 0xB1, 0xf, 0xf, true);      // quad_perm:[1,0,3,2]
        if constexpr (LANES >= 4u) cnt += (u32)__builtin_amdgcn_update_dpp(0, (int)cnt, 0x4E, 0xf, 0xf, true);   // quad_perm:[2,3,0,1]
        const u32 x = cnt - 1u;                                // the row's sum, in every lane
        u32 N0 = (u32)__builtin_amdgcn_update_dpp(0, (int)T.d[0], 0x101, 0xf, 0xf, true);    // row_shl:1 -- lane i takes lane i + 1
        N0 = last_lane ? TRC_PROB_ONE : N0;
        const u32 kk = (x >> 1) & (D - 1u);                    // the pair starts in dword kk of lane x / 2D
        u32 lo = T.d[0], hi = T.d[1];
#pragma unroll
        for (u32 k = 1; k < D; k++) { const bool ge = kk >= k; lo = ge ? T.d[k] : lo; hi = ge ? (k + 1u < D ? T.d[k + 1u < D ? k + 1u : 0u] : N0) : hi; }
        const u32 V = __builtin_amdgcn_alignbit(hi, lo, (x << 4) & 16u);     // x even: lo; odd: lo.hi | hi.lo << 16
        q.cc = (u32)__builtin_amdgcn_ds_bpermute((int)((((x / (2u * D)) & (LANES - 1u)) << 2) | rowb4), (int)V);
        return x;
    };
    auto finish = [&](u32 &s, Tab &T, const Nib &q) {
        const u32 c0 = q.cc & 0xffffu;
        s = __umul24((q.cc >> 16) - c0, s >> TRC_PROB_BITS) + q.slot - c0;
#pragma unroll
        for (u32 k = 0; k < D; k++) {
            const u32 K = (u32)(__mul24((int)q.f.d[k], -32736) + (int)kbase.d[k]);
            T.d[k] = trc_as_u32(trc_as_s2(T.d[k]) + ((trc_as_s2(K) - trc_as_s2(T.d[k])) >> (trc_s2)7));
        }
    };
    Tab pendH = fresh;                                         // the next byte's hi table, asked for at the end of this byte
    auto get_byte = [&](bool act, u32 &sh, u32 &sl) -> u32 {    // context cx -> byte, which becomes the context
        const u32 cb = (cx << 9) + (cx << 5);                 // 32 x the context's first table (17 tables per context: 544 bytes)
        {
            const u32 noff = (cb ^ swz) + moff;
            if (act && noff != hoff) {                         // the row's hi table goes back to memory, the context's comes in
                st_tab(hoff, H);
                const u32 a = seen + 512u + ((cx >> 5) << 2), bit = 1u << (cx & 31u);
                const u32 bits = *(const lds_u32 *)(uintptr_t)a;
                *(lds_u32 *)(uintptr_t)a = bits | bit;         // (every lane of the row writes the same word)
                H = (bits & bit) ? pendH : fresh;              // (a table never written: whatever was loaded, dropped)
                hoff = noff;
            }
        }
        Nib qh, ql;
        const u32 h = find(sh, H, qh) & 15u;
        const u32 noffl = ((cb + 32u + (h << 5)) ^ swz) + moff;
        const bool needl = act && noffl != loff;
        Tab pendL = fresh;
        if (needl) pendL = ld_tab(noffl);
        finish(sh, H, qh);
        if (needl) {
            st_tab(loff, L);
            const u32 a = seen + cx * 2u, bit = 1u << h;
            const u32 bits = *(const lds_u16 *)(uintptr_t)a;
            *(lds_u16 *)(uintptr_t)a = (u16)(bits | bit);
            L = (bits & bit) ? pendL : fresh;
            loff = noffl;
        }
        const u32 l = find(sl, L, ql) & 15u;
        const u32 b = h << 4 | l;
        cx = act ? b : cx;
        {
            const u32 nh = (((cx << 9) + (cx << 5)) ^ swz) + moff;
            if (nh != hoff) pendH = ld_tab(nh);
        }
        finish(sl, L, ql);
        return b;
    };

    struct __attribute__((packed, aligned(2))) W2 { u32 lo, hi; };
    u8 *const dst = out + (u64)(alive ? c : 0u) * chunk;
    for (u32 p0 = 0; p0 < chunk; p0 += 4u * LANES) {           // 4 LANES output bytes per trip: lane e keeps dword e of them
        if (!__ballot(p0 < lenc)) break;
        u32 acc = 0;
#pragma nounroll
        for (u32 d = 0; d < LANES; d++) {
            u32 w = 0;
#pragma unroll
            for (int j = 0; j < 2; j++) {                      // mndec8x2x: two bytes, then four renorms in order st0..st3
                const bool act = p0 + 4u * d + 2u * (u32)j < lenc;             // the second byte of an odd tail is the dummy
                const W2 w2 = *(const W2 *)(sbase + (soff + trc_min(rpos, lim)));   // the (up to) four words this pair's renorms take, requested now
                const u64 ww = ((u64)w2.hi << 32) | w2.lo;
                u32 taken = 0;                                 // bits of ww used up
                auto renorm = [&](u32 &s) {                    // (rows that are done or raw run on garbage: nothing of theirs is used)
                    const bool rn = s < TRC_ANS_LOW;
                    s = rn ? __builtin_amdgcn_perm(s, (u32)(ww >> taken), 0x05040100u) : s;      // (s << 16) | next word
                    taken += rn ? 16u : 0u;
                };
                const u32 x0 = get_byte(act, st0, st1);
                const u32 x1 = get_byte(act, st2, st3);
                w |= (x0 | x1 << 8) << (16 * j);
                renorm(st0); renorm(st1); renorm(st2); renorm(st3);   // (the first two moved up between the bytes, under the second byte's hi table load: no change, 3.19-3.20 ms either way)
                rpos += act ? taken >> 3 : 0u;
            }
            acc = e == d ? w : acc;
        }
        const u32 pos = p0 + 4u * e;
        if (pos + 4u <= lenc) *(u32 *)(dst + pos) = acc;
        else if (pos < lenc)                                    // ragged end of the last chunk
            for (u32 k = 0; pos + k < lenc; k++) dst[pos + k] = (u8)(acc >> (8u * k));
    }
    trc_wave_copy_raw(__ballot(lane < CHUNKS && c0w + lane < nchunks && cl_w == len_w && len_w != 0u), gbase + ex_w, len_w,
                      out + (u64)c0w * chunk, chunk, payload);
}

// ------------------------------------------------------------------------------------- launch ---
// returns true when the records were written to the PLANAR record space (the caller then runs the planar coding pass)
bool trc_launch_anso1_model(const uint8_t *d_in, size_t n, uint32_t chunk, const TrcWork &w, hipStream_t s)
{
    static const int chains = getenv("TRC_O1_CHAINS") ? atoi(getenv("TRC_O1_CHAINS")) : 1;   // 0: the position-order passes below
    if (chains && chunk <= 4096u) {
        TRC_LAUNCH_TIMED(trc_o1_sort_kernel, dim3(w.nchunks), dim3(64), 0, s, d_in, (u64)n, chunk, w.nchunks, w.model);
        // G chunks per workgroup: the chip's 256 CUs take one workgroup each (four waves, a SIMD each); beyond 96 a lane's share of the
        // work outlasts the longest chain a chunk can hold (4096 entries: 259 rounds) and several workgroups per CU take turns
        const u32 gq = (w.nchunks + 255u) / 256u, G = gq > O1W_GROUP_MAX ? O1W_GROUP_MAX : gq ? gq : 1u;
        const u32 lds = O1W_LDS(G) > TRC_LDS_ONE_PER_CU ? O1W_LDS(G) : TRC_LDS_ONE_PER_CU;
        TRC_RAISE_LDS_ONCE(trc_o1_walk_kernel, O1W_LDS(O1W_GROUP_MAX) > TRC_LDS_ONE_PER_CU ? O1W_LDS(O1W_GROUP_MAX) : TRC_LDS_ONE_PER_CU);
        TRC_LAUNCH_TIMED(trc_o1_walk_kernel, dim3((w.nchunks + G - 1u) / G), dim3(64 * O1W_WAVES), lds, s, w.nchunks, G, w.model);
        TRC_LAUNCH_TIMED(trc_o1_place_kernel, dim3(w.nchunks), dim3(256), 0, s, (u64)n, chunk, w.nchunks, (const u8 *)w.model, w.scratch2);
        return true;
    }
    static const int two = getenv("TRC_O1_MC") ? atoi(getenv("TRC_O1_MC")) : 1;      // 0: the one-wave pass of rounds 1-3 (interleaved records)
    // ... up to 1024 pairs: beyond that (100 MB at chunk 1024: 1526) the unit is saturated by one wave per 64 chunks already and the
    // second wave only adds its own input loads and record stores (3.30 -> 3.93 ms); chunk 4096: 5.35 -> 3.80 ms, 2048: 3.39 -> 3.19
    if (two && w.ngroups <= 1024u) {
        // ONE pair per workgroup: these waves are bound by the texture-address unit of their CU (64 scattered 16-byte accesses per
        // table move), not by their SIMD -- four pairs per workgroup put 100 MB at chunk 4096 (382 pairs) on 96 CUs and ran 30 %
        // slower than the one-wave pass; as single pairs they spread over all 256 (profiles/r04_notes.md)
        TRC_RAISE_LDS_ONCE(trc_o1_model2_kernel<1>, O1M2_LDS);
        TRC_LAUNCH_TIMED(trc_o1_model2_kernel<1>, dim3(w.ngroups), dim3(128), O1M2_LDS, s, d_in, (u64)n, chunk, w.nchunks, w.model, w.scratch2);
        return true;
    }
    TRC_LAUNCH_TIMED(trc_o1_model_kernel, dim3(w.ngroups), dim3(64), 0, s, d_in, (u64)n, chunk, w.nchunks, w.model, w.scratch2);
    return false;
}
void trc_launch_anso1_dec(const uint8_t *d_payload, const uint32_t *d_clen, size_t n, uint32_t chunk,
                          const TrcWork &w, uint8_t *d_out, hipStream_t s)
{
    static const int env_rows = getenv("TRC_O1_ROWS") ? atoi(getenv("TRC_O1_ROWS")) : 0;       // tuning aid: 64 / 16 / 8 force the form
    // eight lanes per chunk at every size (100 MB: chunk 4096 5.28 -> 3.79 ms, 2048 3.67 -> 3.71, 1024 3.83 -> 3.71: profiles/r05_notes.md);
    // the one-lane-per-chunk forms (64 / 16 / 8 chunks per wave) stay behind TRC_O1_ROWS
    const int rows = env_rows ? env_rows : TRC_O1_DEC_DEFAULT_ROWS(w.ngroups);
    if (rows == 4)                                              // four lanes per chunk
        TRC_LAUNCH_TIMED(trc_o1_dec_rowsn_kernel<4>, dim3(w.ngroups * 4u), dim3(64), 0, s, d_payload, d_clen, w.goff, w.gsum, (u64)n, chunk, w.nchunks, w.model, d_out);
    else if (rows == 2)                                         // two lanes per chunk
        TRC_LAUNCH_TIMED(trc_o1_dec_rowsn_kernel<2>, dim3(w.ngroups * 2u), dim3(64), 0, s, d_payload, d_clen, w.goff, w.gsum, (u64)n, chunk, w.nchunks, w.model, d_out);
    else if (rows == 1)                                         // eight lanes per chunk (trc_o1_dec_rows_kernel)
        TRC_LAUNCH_TIMED(trc_o1_dec_rows_kernel, dim3(w.ngroups * (64u / O1R_CHUNKS)), dim3(64), 0, s, d_payload, d_clen, w.goff, w.gsum, (u64)n, chunk, w.nchunks, w.model, d_out);
    else if (rows == 16)
        TRC_LAUNCH_TIMED(trc_o1_dec_kernel<16>, dim3(w.ngroups * 4u), dim3(64), 0, s, d_payload, d_clen, w.goff, w.gsum, (u64)n, chunk, w.nchunks, w.model, d_out);
    else if (rows == 8)
        TRC_LAUNCH_TIMED(trc_o1_dec_kernel<8>, dim3(w.ngroups * 8u), dim3(64), 0, s, d_payload, d_clen, w.goff, w.gsum, (u64)n, chunk, w.nchunks, w.model, d_out);
    else
        TRC_LAUNCH_TIMED(trc_o1_dec_kernel<64>, dim3(w.ngroups), dim3(64), 0, s, d_payload, d_clen, w.goff, w.gsum, (u64)n, chunk, w.nchunks, w.model, d_out);
}
#ifdef TRC_O1W_PROF
extern "C" __attribute__((visibility("default"))) int trc_o1w_prof_read(void *dst, size_t bytes)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(trc_o1w_prof), bytes, 0, hipMemcpyDeviceToHost);
}
#endif
