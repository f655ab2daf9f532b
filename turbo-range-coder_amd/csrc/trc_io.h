// trc_io.h -- wave-cooperative chunk I/O shared by all coder kernels (gfx950, wave64).
//
// One LANE codes one CHUNK, but in the fast (static) coders no lane talks to HBM on its own: PMC runs of the
// first version showed ~5 TA cycles per scattered 16-byte lane access and every access becoming its own
// L2 request (profiles/r01_notes.md).  Their traffic is therefore moved in 64-BYTE SEGMENTS by QUADS of lanes
// (4 x 16 B contiguous).  (The model-bound coders stream through per-lane register windows: trc_lane_io.h.)
//
//   TileIn     uniform-rate input through a 64 rows x 64 B LDS tile, loaded ahead (record stack of the adaptive rANS)
//   QuadIn/QuadOut  uniform-rate chunk bytes in/out: the same 64-byte segments, transposed inside quads of lanes in
//              registers instead of through an LDS tile
//   StreamOut  variable-rate output (coded bytes at encode): 128-B ring per lane; lanes whose ring
//              holds a full segment are ranked (ballot + mbcnt) and 16 of them are drained per
//              round, each by one quad
//   StreamIn   variable-rate input (coded bytes at decode):  128-B ring per lane, refilled the same
//              way one period ahead of use
//
// LDS rows are padded so that the per-lane accesses are bank-conflict free:
//   tiles: row stride 80 B  (20 dwords: 16 consecutive rows x 4 banks cover all 64 banks for b128)
//   rings: row stride 132 B (33 dwords: lanes at equal ring offsets hit 32 distinct banks)
#pragma once
#include "trc_dev.h"

#define TRC_SEG          64u      // bytes moved per quad
#define TRC_TILE_STRIDE  80u
#define TRC_TILE_BYTES   (64u * TRC_TILE_STRIDE)            // 5120 per wave
#define TRC_SRING        128u     // stream ring bytes per lane
#define TRC_SRING_STRIDE 132u
#define TRC_SRING_BYTES  (64u * TRC_SRING_STRIDE)           // 8448 per wave

// Byte address, inside one wave's ring array, of ring offset `off` (0..127) of lane `lane`.
//   default: lane-major rows of 132 bytes (33 dwords: conflict-free while lanes sit at equal offsets,
//     ~3-way conflicts once the data-dependent offsets drift apart).
//   TRC_RING_INTERLEAVED: dword d of every lane's ring in row d -> ((off>>2)*64 + lane)*4 + (off&3): never
//     conflicts, but costs one more VALU op per access.  Measured on MI355X (same box, interleaved runs):
//     decode 124 vs 113 us, encode 101 vs 99 us in favour of the linear form -- the ring is not where the
//     LDS conflict cycles come from (the random table reads are).
#ifdef TRC_RING_INTERLEAVED
#error "TRC_RING_INTERLEAVED is gone: StreamInT<true> is the interleaved ring (static rANS decoder); StreamOut / StreamInT<false> assume the lane-major rows"
#endif
__device__ __forceinline__ u32 trc_raddr(u32 lane, u32 off) { return lane * TRC_SRING_STRIDE + off; }

__device__ __forceinline__ u32 trc_mbcnt(u64 mask)   // number of set bits of mask below this lane
{
    return __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u));
}

// Round 6, the ARRIVAL GATE of an input that is still on its way (host-pointer encodes, trc_host.inc): the host delivers the input in
// K passes -- pass k = bytes [k * part, (k + 1) * part) of EVERY chunk, one 2-D DMA copy -- and sets *gate = k + 1 behind each pass (a
// 4-byte copy on the same stream); the kernel is launched before the first byte has arrived and a wave waits here before it asks for
// the first segment of a part.  The wave's coding time then overlaps the transfer instead of following it.  Rules that make this
// safe: chunk and part are multiples of 128 bytes, so no cache line holds bytes of two passes; a line of part k is first touched
// after the flag says it is there (the caches were invalidated at the kernel's start); the flag is read at system scope, past the
// caches; the wait is BOUNDED -- a pass that never arrives is reported to the host (word 63 of the gate area), not waited for forever.
#define TRC_GATE_SPIN_MAX (1u << 20)                            // polls ~1 us apart: about a second
#define TRC_GATE_AREA 256u                                      // the gates of a call: 64 words, 256-byte aligned; word 63 = "a wait timed out"
// returns false when the pass never arrived: the wave then stops waiting (its output is garbage) and the host, which reads word 63
// behind the kernel, fails the call loudly and stops using gates (trc_host.inc) -- no trap, no hang
__device__ __forceinline__ bool trc_gate_wait(const u32 *gate, u32 need)
{
    u32 spins = 0;
    while (__hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < need) {
        __builtin_amdgcn_s_sleep(32);
        if (++spins > TRC_GATE_SPIN_MAX) {
            u32 *err = (u32 *)(((uintptr_t)gate & ~(uintptr_t)(TRC_GATE_AREA - 1u)) + TRC_GATE_AREA - 4u);
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return false;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // (nothing of the part is in this CU's L1 either; cheap, once per part)
    return true;
}

// Geometry of the 64 chunks one wave owns.
struct WaveChunks {
    u32 c0;        // first chunk of this wave
    u32 rows;      // valid chunks in this wave (1..64; 0 => wave idle)
    u32 chunk;     // nominal chunk bytes
    u32 lastlen;   // length of chunk nchunks-1
    u32 nchunks;
    const u32 *gate = nullptr;   // arrival gate of the input (above), or null: the input is all there
    u32 gate_part = 0;           // bytes of a chunk per pass
    mutable u32 gate_seen = 0;   // passes this wave knows to have arrived
    // Round 6, the PROGRESS of an output that starts its way back before the kernel has ended (host-pointer decodes, trc_host.inc): the
    // mirror of the arrival gate.  Every wave, when it has stored bytes [k part, (k + 1) part) of its chunks -- write-through to memory
    // (QuadOut::flush under `prog`), and waited for -- adds one to prog[k]; the wave that completes the count sets prog_host[k] (page-
    // locked host memory) and the host, which polls it, starts the 2-D copy of part k of every chunk while the waves decode part k + 1.
    u32 *prog = nullptr;         // device: one counter per part, zero at the launch; or null: nobody is listening
    u32 *prog_host = nullptr;    // host (page-locked): one flag per part
    u32 prog_part = 0;           // bytes of a chunk per part
    u64 skip_rows = 0;           // rows QuadOut::flush leaves alone (chunks stored raw, copied BEFORE the loop when somebody listens to `prog`)
    __device__ __forceinline__ u32 len_of(u32 row) const { return (c0 + row == nchunks - 1) ? lastlen : chunk; }
    // behind the stores of the segment that ends at `end` of every chunk (wave-uniform)
    __device__ __forceinline__ void after_flush(u32 end) const
    {
        if (!prog) return;
        if (end % prog_part != 0u && end != chunk) return;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's write-through stores of the part have been acknowledged
        if (trc_lane() == 0u) {
            const u32 k = (end + prog_part - 1u) / prog_part - 1u, nwaves = (nchunks + 63u) / 64u;
            const u32 old = __hip_atomic_fetch_add(prog + k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == nwaves - 1u) __hip_atomic_store(prog_host + k, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    // before the loads of the segment at `segoff` of every chunk are issued (wave-uniform)
    __device__ __forceinline__ void wait_for(u32 segoff) const
    {
        if (!gate) return;
        const u32 need = segoff / gate_part + 1u;
        if (need > gate_seen) gate_seen = trc_gate_wait(gate, need) ? need : ~0u;     // (timed out: stop waiting)
    }
};

// ------------------------------------------------------------------------------------ TileIn ---
struct TileIn {
    u8 *tile;            // this wave's LDS tile
    const u8 *base;      // global address of chunk c0
    uint4 r[4];          // segment in flight
    // request the 64-byte segment at byte offset `segoff` of every chunk of the wave
    __device__ __forceinline__ void issue(const WaveChunks &w, u32 segoff)
    {
        const u32 lane = trc_lane(), part = (lane & 3u) << 4;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            u32 row = (u32)j * 16u + (lane >> 2);
            row = row < w.rows ? row : w.rows - 1;
            // a short last chunk has no bytes in its upper segments: never read past the input's pad
            const u32 so = segoff < w.len_of(row) ? segoff : 0u;
            r[j] = trc_ld16_nt(base + (size_t)row * w.chunk + so + part);     // read once: do not keep it in the caches
        }
    }
    __device__ __forceinline__ void commit()
    {
        const u32 lane = trc_lane(), part = (lane & 3u) << 4;
#pragma unroll
        for (int j = 0; j < 4; j++)
            *(uint4 *)(tile + ((u32)j * 16u + (lane >> 2)) * TRC_TILE_STRIDE + part) = r[j];
    }
    // own row, 16-byte piece k (0..3)
    __device__ __forceinline__ uint4 read(u32 k) const { return *(const uint4 *)(tile + trc_lane() * TRC_TILE_STRIDE + (k << 4)); }
};

// ----------------------------------------------------------------------------------- QuadOut ---
// Uniform-rate output without an LDS tile: every lane keeps its chunk's 64 output bytes in 4 x uint4 registers and
// the 4 lanes of a quad transpose their 4x4 blocks in registers (two DPP butterfly stages, quad_perm
// xor-1 then xor-2, ~1 VALU per byte), after which lane i of quad q holds piece i of the chunks
// 4q..4q+3 and every store instruction again writes whole 64-byte segments.  Frees 5 KiB of LDS per
// wave -- what it takes to keep 12 decoder waves per CU resident next to the 34 KiB symbol tables.
__device__ __forceinline__ u32 trc_quad_xor1(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false); }
__device__ __forceinline__ u32 trc_quad_xor2(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false); }
// 4x4 transpose of uint4 blocks inside every quad of lanes: afterwards m[j] of lane i is what m[i] of lane
// (quad base + j) held (an involution, used in both directions).
// Round 3: the DPP lane swap folded into the select (v_cndmask_b32_dpp: D = VCC ? src1 : quad_perm(src0)) -- one VALU per
// dword and stage, 32 per 64 bytes, where the C form below costs 64 (select the word to send, v_mov_dpp, two selects).
#ifndef TRC_QUAD_DPP
#define TRC_QUAD_DPP 1
#endif
// Round 5: runs of v_cndmask_b32 in the encodings that read VCC implicitly (VOP2, DPP, SDWA) are pathological on gfx950 -- from the
// third one on EACH occupies the SIMD for ~22.7 cycles whatever else is resident (scripts/probe/valu_occ.hip, profiles/r05_valu_rates.txt:
// "cnd_dpp vcc b2b" 22.7 cycles per instruction at 1..8 waves per SIMD; an s_nop or a scalar instruction between them changes
// nothing, a plain vector instruction does: "cnd_dpp,v_mov" 4.4-5.5).  Rounds 3-4 ran EIGHT of them back to back per half-stage:
// 32 x 22.7 = 727 SIMD cycles per 64 bytes and wave, ~11 % of the static decoder.  Now every v_cndmask_b32_dpp is followed by
// two instructions that do not read VCC: of a pair of rows (X, Y) that swap halves with the partner lane, the X side is
//     new X = keep ? X : dpp(Y)            one v_cndmask_b32_dpp (VCC = the lanes that keep X)
// and the Y side
//     new Y = dpp(X); new Y = keepY ? Y : new Y    v_mov_b32_dpp + v_cndmask_b32_e64 on an SGPR-pair mask (VOP3: no such penalty)
// 48 vector instructions of the 4-cycle class per 64 bytes instead of 32 of the 22.7-cycle kind.  One block = one pair of rows
// (4 columns); a DPP source must have been written at least two instructions earlier and the compiler cannot see DPP inside an
// asm block: every block starts with s_mov vcc + s_nop.
#define TRC_QT_PAIR(PERM, KEEPX, KEEPY, A0, A1, A2, A3, B0, B1, B2, B3, X0, X1, X2, X3, Y0, Y1, Y2, Y3)                             \
    asm volatile("s_mov_b64 vcc, %[kx]\n\ts_nop 0\n\t"                                                                              \
        "v_cndmask_b32_dpp %[a0], %[y0], %[x0], vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"                                        \
        "v_mov_b32_dpp %[b0], %[x0] " PERM " row_mask:0xf bank_mask:0xf\n\t"                                                        \
        "v_cndmask_b32_e64 %[b0], %[b0], %[y0], %[ky]\n\t"                                                                          \
        "v_cndmask_b32_dpp %[a1], %[y1], %[x1], vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"                                        \
        "v_mov_b32_dpp %[b1], %[x1] " PERM " row_mask:0xf bank_mask:0xf\n\t"                                                        \
        "v_cndmask_b32_e64 %[b1], %[b1], %[y1], %[ky]\n\t"                                                                          \
        "v_cndmask_b32_dpp %[a2], %[y2], %[x2], vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"                                        \
        "v_mov_b32_dpp %[b2], %[x2] " PERM " row_mask:0xf bank_mask:0xf\n\t"                                                        \
        "v_cndmask_b32_e64 %[b2], %[b2], %[y2], %[ky]\n\t"                                                                          \
        "v_cndmask_b32_dpp %[a3], %[y3], %[x3], vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"                                        \
        "v_mov_b32_dpp %[b3], %[x3] " PERM " row_mask:0xf bank_mask:0xf\n\t"                                                        \
        "v_cndmask_b32_e64 %[b3], %[b3], %[y3], %[ky]"                                                                              \
        : [a0] "=&v"(A0), [a1] "=&v"(A1), [a2] "=&v"(A2), [a3] "=&v"(A3), [b0] "=&v"(B0), [b1] "=&v"(B1), [b2] "=&v"(B2), [b3] "=&v"(B3) \
        : [kx] "s"(KEEPX), [ky] "s"(KEEPY), [x0] "v"(X0), [x1] "v"(X1), [x2] "v"(X2), [x3] "v"(X3),                                 \
          [y0] "v"(Y0), [y1] "v"(Y1), [y2] "v"(Y2), [y3] "v"(Y3) : "vcc")
__device__ __forceinline__ void trc_quad_transpose_dpp(u32 (&m)[4][4])
{
    const u64 even = 0x5555555555555555ull, odd = 0xAAAAAAAAAAAAAAAAull, lo2 = 0x3333333333333333ull, hi2 = 0xCCCCCCCCCCCCCCCCull;
    u32 a[4][4];
    // stage 1, lane ^ 1 on rows (k, k + 1), k = 0, 2: even lanes keep m[k] and take the partner's m[k] as m[k+1]; odd lanes keep
    // m[k+1] and take the partner's m[k+1] as m[k]
    TRC_QT_PAIR("quad_perm:[1,0,3,2]", even, odd, a[0][0], a[0][1], a[0][2], a[0][3], a[1][0], a[1][1], a[1][2], a[1][3],
                m[0][0], m[0][1], m[0][2], m[0][3], m[1][0], m[1][1], m[1][2], m[1][3]);
    TRC_QT_PAIR("quad_perm:[1,0,3,2]", even, odd, a[2][0], a[2][1], a[2][2], a[2][3], a[3][0], a[3][1], a[3][2], a[3][3],
                m[2][0], m[2][1], m[2][2], m[2][3], m[3][0], m[3][1], m[3][2], m[3][3]);
    // stage 2, lane ^ 2 on rows (k, k + 2), k = 0, 1
    TRC_QT_PAIR("quad_perm:[2,3,0,1]", lo2, hi2, m[0][0], m[0][1], m[0][2], m[0][3], m[2][0], m[2][1], m[2][2], m[2][3],
                a[0][0], a[0][1], a[0][2], a[0][3], a[2][0], a[2][1], a[2][2], a[2][3]);
    TRC_QT_PAIR("quad_perm:[2,3,0,1]", lo2, hi2, m[1][0], m[1][1], m[1][2], m[1][3], m[3][0], m[3][1], m[3][2], m[3][3],
                a[1][0], a[1][1], a[1][2], a[1][3], a[3][0], a[3][1], a[3][2], a[3][3]);
}
__device__ __forceinline__ void trc_quad_transpose_c(u32 (&m)[4][4]);
__device__ __forceinline__ void trc_quad_transpose(u32 (&m)[4][4])
{
#if TRC_QUAD_DPP
    trc_quad_transpose_dpp(m);
#else
    trc_quad_transpose_c(m);
#endif
}
__device__ __forceinline__ void trc_quad_transpose_c(u32 (&m)[4][4])
{
    const u32 lane = trc_lane();
    const bool b0 = lane & 1u, b1 = lane & 2u;
#pragma unroll
    for (int c = 0; c < 4; c++) {
#pragma unroll
        for (int k = 0; k < 4; k += 2) {                       // swap bit 0 of (lane, piece) with lane^1
            const u32 r = trc_quad_xor1(b0 ? m[k][c] : m[k + 1][c]);
            const u32 n0 = b0 ? r : m[k][c], n1 = b0 ? m[k + 1][c] : r;
            m[k][c] = n0; m[k + 1][c] = n1;
        }
#pragma unroll
        for (int k = 0; k < 2; k++) {                          // swap bit 1 with lane^2
            const u32 r = trc_quad_xor2(b1 ? m[k][c] : m[k + 2][c]);
            const u32 n0 = b1 ? r : m[k][c], n1 = b1 ? m[k + 2][c] : r;
            m[k][c] = n0; m[k + 2][c] = n1;
        }
    }
}

// QuadIn: TileIn without the LDS tile (same quad-coalesced 64-byte loads, transposed in registers).
struct QuadIn {
    const u8 *base;      // global address of chunk c0
    uint4 r[4];          // segment in flight: r[j] = piece (lane&3) of the chunk of lane (lane&~3)+j
    uint4 r2[4];         // second segment in flight (paired mode: the other 64-byte half of the same 128-byte line)
    uint4 p[4];          // current segment: p[k] = piece k of this lane's chunk
    // pair = true: lanes 2i and 2i + 1 both take chunk i of the wave (one lane per stream of a two-stream coder): the quad's
    // four rows are two chunks, each requested twice (the second request hits the first one's line)
    __device__ __forceinline__ void issue(const WaveChunks &w, u32 segoff, bool pair = false)
    {
        const u32 lane = trc_lane(), part = (lane & 3u) << 4;
        w.wait_for(segoff);                                    // (host-pointer encodes: the part this segment belongs to has arrived)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            u32 row = (lane & ~3u) + (u32)j;
            if (pair) row >>= 1;
            row = row < w.rows ? row : w.rows - 1;
            const u32 so = segoff < w.len_of(row) ? segoff : 0u;     // never read past the input's pad
            r[j] = trc_ld16_nt(base + (size_t)row * w.chunk + so + part);     // read once: do not keep it in the caches
        }
    }
    __device__ __forceinline__ void commit()
    {
        u32 m[4][4];
#pragma unroll
        for (int j = 0; j < 4; j++) { m[j][0] = r[j].x; m[j][1] = r[j].y; m[j][2] = r[j].z; m[j][3] = r[j].w; }
        trc_quad_transpose(m);
#pragma unroll
        for (int k = 0; k < 4; k++) p[k] = make_uint4(m[k][0], m[k][1], m[k][2], m[k][3]);
    }
    __device__ __forceinline__ uint4 read(u32 k) const { return p[k]; }           // k is a literal after unrolling

    // Paired mode, for kernels that walk their chunks DOWNWARD: the two 64-byte halves of a 128-byte line are requested in
    // the same breath (odd segment -> r, the even one below it -> r2) instead of one segment-time apart.  With the
    // nontemporal hint the first half's line was gone from the L2 by the time the second half was asked for: round 1's PMC
    // showed 167 MB read for a 100 MB input.  Protocol (s = segment index, wave-uniform):
    //   start:   issue_slot<(S-1)&1>(S-1), and issue_slot<0>(S-2) too if S-1 is odd
    //   every s: commit_slot<s&1>(); if s is even and s >= 1: issue_slot<1>(s-1) and, if s >= 2, issue_slot<0>(s-2); code segment s
    template <int ODD>                                         // ODD = s & 1 (a literal: the register sets must not be indexed at run time)
    __device__ __forceinline__ void issue_slot(const WaveChunks &w, u32 s)
    {
        const u32 lane = trc_lane(), part = (lane & 3u) << 4, segoff = s * TRC_SEG;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            u32 row = (lane & ~3u) + (u32)j;
            row = row < w.rows ? row : w.rows - 1;
            const u32 so = segoff < w.len_of(row) ? segoff : 0u;     // never read past the input's pad
            const uint4 v = trc_ld16_nt(base + (size_t)row * w.chunk + so + part);
            if (ODD) r[j] = v; else r2[j] = v;
        }
    }
    template <int ODD>
    __device__ __forceinline__ void commit_slot()
    {
        u32 m[4][4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint4 v = ODD ? r[j] : r2[j];
            m[j][0] = v.x; m[j][1] = v.y; m[j][2] = v.z; m[j][3] = v.w;
        }
        trc_quad_transpose(m);
#pragma unroll
        for (int k = 0; k < 4; k++) p[k] = make_uint4(m[k][0], m[k][1], m[k][2], m[k][3]);
    }
    // The same for kernels that walk their chunks UPWARD (round 5: the model passes of the adaptive coders read 1.8 x their input
    // from the fabric -- each nontemporal 64-byte request pulled its 128-byte line, and the line's other half was asked for one
    // segment-time later, when it was gone).  Segments s (even) and s + 1 are requested together, one segment ahead of their use:
    //   start_fwd(w, S);   every s: take_fwd(w, s, S); code segment s (read(k))
    __device__ __forceinline__ void start_fwd(const WaveChunks &w, u32 S)
    {
        issue_slot<0>(w, 0);
        if (S > 1u) issue_slot<1>(w, 1);
    }
    __device__ __forceinline__ void take_fwd(const WaveChunks &w, u32 s, u32 S)
    {
        if (s & 1u) {
            commit_slot<1>();
            if (s + 1u < S) { issue_slot<0>(w, s + 1u); if (s + 2u < S) issue_slot<1>(w, s + 2u); }     // both register sets are free now
        } else commit_slot<0>();
    }
};

struct QuadOut {
    u8 *base;            // global address of chunk c0 in the output
    uint4 p[4];
    __device__ __forceinline__ void put(u32 k, uint4 v) { p[k] = v; }
    // pair = true: lanes 2i and 2i + 1 hold the SAME 64 bytes of chunk i (one lane per stream, bytes merged across the pair):
    // after the transpose the quad's rows 0, 2 are its two chunks, rows 1, 3 their duplicates and are not stored
    __device__ __forceinline__ void flush(const WaveChunks &w, u32 segoff, bool pair = false)
    {
        const u32 lane = trc_lane();
        u32 m[4][4];                                           // m[k][c]: dword c of piece k
#pragma unroll
        for (int k = 0; k < 4; k++) { m[k][0] = p[k].x; m[k][1] = p[k].y; m[k][2] = p[k].z; m[k][3] = p[k].w; }
        trc_quad_transpose(m);
        const u32 part = (lane & 3u) << 4;
#pragma unroll
        for (int j = 0; j < 4; j++) {                          // m[j] = piece (lane&3) of the chunk of lane (lane&~3)+j
            if (pair && (j & 1)) continue;
            const u32 vrow = (lane & ~3u) + (u32)j, row = pair ? vrow >> 1 : vrow;
            if (row < w.rows && segoff + part + 16u <= w.len_of(row) && !((w.skip_rows >> row) & 1ull)) {
                u8 *const a = base + (size_t)row * w.chunk + segoff + part;
                const uint4 v = make_uint4(m[j][0], m[j][1], m[j][2], m[j][3]);
                if (w.prog) trc_st16_sys(a, v); else trc_st16_nt(a, v);       // (somebody copies the part away before the kernel ends: past the caches)
            }
        }
        w.after_flush(segoff + TRC_SEG);
    }
};

// --------------------------------------------------------------------------------- StreamOut ---
// DOWN = true : units are appended downward from the END of the chunk's scratch region (rANS)
// DOWN = false: upward from the START of the region (range coders)
// PAIR = true (round 3, the two-stream range coder with one LANE PER STREAM): lanes 2i and 2i + 1 work on chunk c0 + i;
// the even lane's stream goes to the chunk's region in `scratch`, the odd lane's to its region in `scratch_b`.
// QUAD = true (round 4, the four-lanes-per-chunk rANS coding pass): lanes 4i .. 4i + 3 work on chunk c0 + i and share ONE ring
// (row i) and one stream; wpos / nfl are kept equal across the quad by the caller, every lane writes its own units at
// positions it computes itself (put16_at), the quad's first lane is the one that drains.
// WT (round 5): the drain's 64-byte stores go out write-through at agent scope (`global_store_dwordx4 ... sc1`) instead of staying
// dirty in the L2: the kernel's end no longer writes back up to 32 MB of dirty lines at once (static rANS encoder 55.0 -> 53.2 us,
// rocprofv3; the gather behind it unchanged at 29.8 -- unlike the nontemporal hint, which took the staged payload out of the L2 and
// cost the gather 9 us).  INVARIANT of every WT = true instantiation (the static rANS encoder, `rcs` with one stream, the code-quad
// passes of `anscdf` / `ansb`): nothing in the kernel loads from, or stores again into, a segment that has been drained -- the store is
// inline asm, outside the compiler's vmcnt accounting, so neither a later access nor a release at the kernel's end would wait for it
// on the compiler's say-so.  A coder that patches a header behind a drain (the two-stream range coders) must keep WT = false; code that
// READS drained bytes in the same kernel (the fused gather, trc_gather.h) puts its own `s_waitcnt vmcnt(0)` + barrier in between.
template <bool DOWN, bool PAIR = false, bool QUAD = false, bool WT = false>
struct StreamOut {
    u8 *rings;           // this wave's ring array (LDS)
    u8 *scratch;         // global scratch, region of chunk c is [c*stride, (c+1)*stride)
    u32 stride;
    u32 c0;
    u8 *scratch_b;       // PAIR only: the second stream's regions
    u32 stride_b;
    __device__ __forceinline__ u32 my_row() const { return PAIR ? trc_lane() >> 1 : QUAD ? trc_lane() >> 2 : trc_lane(); }
    __device__ __forceinline__ u32 ring_row() const { return QUAD ? trc_lane() >> 2 : trc_lane(); }
    // (QUAD) a 16-bit unit at stream position p of the shared stream, where `take`; nothing moves
    __device__ __forceinline__ void put16_at(bool take, u32 p, u32 v)
    {
        if (take) *(u16 *)(rings + trc_raddr(ring_row(), roff16(p))) = (u16)v;
    }
    __device__ __forceinline__ u32 my_stride() const { return (PAIR && (trc_lane() & 1u)) ? stride_b : stride; }
    __device__ __forceinline__ u8 *my_region() const
    {
        return (PAIR && (trc_lane() & 1u)) ? scratch_b + (size_t)(c0 + my_row()) * stride_b : scratch + (size_t)(c0 + my_row()) * stride;
    }
    u32 wpos;            // bytes appended by this lane so far
    u32 nfl;             // 64-byte segments already moved to the region

    // ring offset of the unit that starts at stream position p
    __device__ __forceinline__ u32 roff16(u32 p) const { return DOWN ? ((0u - (p + 2u)) & (TRC_SRING - 1)) : (p & (TRC_SRING - 1)); }
    __device__ __forceinline__ u32 roff32(u32 p) const { return DOWN ? ((0u - (p + 4u)) & (TRC_SRING - 1)) : (p & (TRC_SRING - 1)); }
    // speculative append: the slot of the next unit is always free, so write unconditionally and
    // only advance when `take` (no exec-mask branch in the symbol loop)
    __device__ __forceinline__ void put16_if(bool take, u32 v)
    {
        *(u16 *)(rings + trc_raddr(trc_lane(), roff16(wpos))) = (u16)v;
        wpos += take ? 2u : 0u;
    }
    __device__ __forceinline__ void put32_if(bool take, u32 v)          // same, 32-bit unit (a full ring is drained before 4 bytes are missing)
    {
        *(u32 *)(rings + trc_raddr(trc_lane(), roff32(wpos))) = v;
        wpos += take ? 4u : 0u;
    }
    __device__ __forceinline__ void put16(u32 v) { *(u16 *)(rings + trc_raddr(trc_lane(), roff16(wpos))) = (u16)v; wpos += 2; }
    __device__ __forceinline__ void put32(u32 v) { *(u32 *)(rings + trc_raddr(trc_lane(), roff32(wpos))) = v; wpos += 4; }

    __device__ __forceinline__ u32 pending() const { return wpos - TRC_SEG * nfl; }
    // one lane moving its own oldest segment to the region (bursts only: runs of 0xFFFFFFFF words
    // released by a range-coder carry can exceed what a period may append)
    __device__ __forceinline__ void self_drain()
    {
        const u32 ro = DOWN ? ((0u - TRC_SEG * (nfl + 1u)) & (TRC_SRING - 1)) : ((TRC_SEG * nfl) & (TRC_SRING - 1));
        u8 *reg = my_region();
        u8 *d = DOWN ? reg + my_stride() - (size_t)TRC_SEG * (nfl + 1u) : reg + (size_t)TRC_SEG * nfl;
        for (u32 i = 0; i < 16; i++) ((u32 *)d)[i] = *(const u32 *)(rings + trc_raddr(trc_lane(), ro + 4u * i));
        nfl++;
    }
    __device__ __forceinline__ void put32_slow(u32 v)
    {
        if (pending() + 4u > TRC_SRING - 4u && (size_t)TRC_SEG * (nfl + 2u) <= my_stride()) self_drain();
        put32(v);
    }
    __device__ __forceinline__ void put16_slow(u32 v)
    {
        if (pending() + 2u > TRC_SRING - 4u && (size_t)TRC_SEG * (nfl + 2u) <= my_stride()) self_drain();
        put16(v);
    }

    // Move finished segments to HBM.  Called by the whole wave at uniform points.
    // final = false: lanes holding >= 64 pending bytes;  final = true: every lane with pending bytes
    // (the partial segment is written as a whole 64 B; the surplus lands in the region's slack).
    // Round 3: the up to 16 lanes drained in a round PUSH the LDS address of their oldest segment and its place in the scratch
    // array (relative to the wave's first region) to the leader of helper quad `rank` with two ds_permute_b32 (lanes not
    // picked push to lane 1, which leads no quad), and the quad takes them from its leader by DPP -- one cross-lane round
    // trip per round where round 2 made two (rank -> lane table written to and read back from LDS, then a shuffle).
    __device__ __forceinline__ void drain(bool final, bool alive)
    {
        const u32 lane = trc_lane();
        const u32 rw = trc_lds_addr(rings);
        bool ready = alive && (final ? pending() > 0 : pending() >= TRC_SEG) && (!QUAD || (lane & 3u) == 0u);
        u64 mask = __ballot(ready);
        while (mask) {
            const u32 rank = trc_mbcnt(mask);
            const bool pick = ready && rank < 16u;
            const u32 cnt = (u32)__popcll(mask);
            const u32 ro = DOWN ? ((0u - TRC_SEG * (nfl + 1u)) & (TRC_SRING - 1)) : ((TRC_SEG * nfl) & (TRC_SRING - 1));
            const u32 from = rw + trc_raddr(ring_row(), ro);
            // place in the scratch array relative to the wave's first region (63 regions of at most 64 KiB + slack: 32 bits; offsets
            // are multiples of 64, so bit 0 can name the array: PAIR, odd lane = second stream)
            const u32 to = (my_row() * my_stride() + (DOWN ? my_stride() - TRC_SEG * (nfl + 1u) : TRC_SEG * nfl)) | (PAIR ? lane & 1u : 0u);
            const int dst = (int)((pick ? rank << 2 : 1u) << 2);
            const u32 from_l = (u32)__builtin_amdgcn_ds_permute(dst, (int)from), to_l = (u32)__builtin_amdgcn_ds_permute(dst, (int)to);
            const u32 from_q = (u32)__builtin_amdgcn_update_dpp(0, (int)from_l, 0x00, 0xf, 0xf, false);   // quad_perm [0,0,0,0]
            const u32 to_q = (u32)__builtin_amdgcn_update_dpp(0, (int)to_l, 0x00, 0xf, 0xf, false);
            const u32 q = lane >> 2, part = (lane & 3u) << 4;
            if (q < cnt && q < 16u) {
                typedef __attribute__((address_space(3))) u32 lds_u32;
                const u32 a = from_q + part;
                const u32 s0 = *(const lds_u32 *)(uintptr_t)a, s1 = *(const lds_u32 *)(uintptr_t)(a + 4u);
                const u32 s2 = *(const lds_u32 *)(uintptr_t)(a + 8u), s3 = *(const lds_u32 *)(uintptr_t)(a + 12u);
                u8 *base = (PAIR && (to_q & 1u)) ? scratch_b + (size_t)c0 * stride_b : scratch + (size_t)c0 * stride;
if constexpr (WT) {                                // (an asm store is not in the compiler's vmcnt books: nothing in these kernels reads it back)
                    const trc_v4u vv = { s0, s1, s2, s3 };
                    u8 *pp = base + (to_q & ~1u) + part;
                    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(pp), "v"(vv) : "memory");
                } else
                    *(uint4 *)(base + (to_q & ~1u) + part) = make_uint4(s0, s1, s2, s3);
            }
            if (pick) { nfl++; ready = final ? (wpos > TRC_SEG * nfl) : pending() >= TRC_SEG; }
            mask = __ballot(ready);
        }
        if (QUAD) nfl = (u32)__builtin_amdgcn_update_dpp(0, (int)nfl, 0x00, 0xf, 0xf, false);       // the quad follows its first lane
    }
};

// ---------------------------------------------------------------------------------- StreamIn ---
// Look-ahead is TWO periods: the round requested in period k lands in period k+2 (two register sets
// A/B alternate by period parity; `par` is a literal after unrolling), which covers HBM/MALL latency
// with ~32 symbols of work.  A lane asks for its next segment as soon as its ring has room
// (avail + in-flight <= 64), i.e. long before it runs dry; if a lane nevertheless gets low
// (adversarial data: every symbol renormalising) the caller falls back to sync_refill().
// IL = false: lane-major rows of 132 bytes (trc_raddr).  IL = true (round 3, static rANS decoder): dword d of every lane's ring
// in row d (row = 64 lanes x 4 B = 256 B): a wave's accesses to its rings then never conflict, whatever the lanes' cursors
// are (bank = lane), against ~3.4 cycles per 32-lane group for the drifted cursors of the lane-major form.  Row 32 (lane-major:
// the row's 4 pad bytes) mirrors the ring's first dword, so a reader may take the dword behind its cursor and the next one
// from one address without wrapping.  Both forms take 8448 bytes per wave.
template <bool IL>
struct StreamInT {
    static __device__ __forceinline__ u32 ra(u32 lane, u32 off)
    {
        return IL ? ((off >> 2) << 8) + (lane << 2) + (off & 3u) : trc_raddr(lane, off);
    }
    u8 *rings;           // this wave's ring array (LDS)
    const u8 *gbase;     // payload base (kernel argument: keeps the loads in the global address space)
    u64 soff;            // this lane's stream start, bytes from gbase (2-byte aligned)
    u64 wbase;           // (set by prime) soff of the wave's first lane: wave-uniform
    u32 srel;            // (set by prime) soff - wbase: the lanes' streams follow one another, all within 64 chunks
    u32 lim;             // no segment is fetched from beyond this stream offset (a valid stream never consumes from
                         // there; a corrupt one re-reads its last segment instead of running off the payload buffer)
    u32 rpos;            // bytes consumed
    u32 lbytes;          // bytes committed to the ring
    u32 infl;            // segments requested for this lane and not yet committed (0..2)
    u32 rw;              // (set by prime) LDS byte address of this wave's ring array
    bool mineA, mineB;   // ... and in which register set they travel
    // helper side: this lane moves one 16-byte piece of some lane's segment, per register set
    uint4 hvA, hvB; u32 hdA, hdB; bool hokA, hokB;

    // Round 5: segments aligned to 64 bytes of the PAYLOAD, not of the stream.  Call after gbase / soff / lim are set and before
    // prime: the ring's "stream" then starts at the 64-byte boundary below the first byte; set rpos to the returned r0 after prime
    // (r0 bytes of the first segment are somebody else's).  Every 64-byte refill request is then one 64-byte sector of one line;
    // stream-relative segments at 2-byte alignment straddled two (the static rANS decoder fetched its payload 2.7 times,
    // profiles/r04_pmc_traffic.txt).
    __device__ __forceinline__ u32 align_start(bool on)
    {
#ifdef TRC_DEC_NOALIGN
        return 0u;
#else
        const u32 r0 = on ? trc_min((u32)(((uintptr_t)gbase + soff) & 63u), soff < 62u ? (u32)soff : 62u) & ~1u : 0u;
        soff -= r0; lim += r0;
        return r0;
#endif
    }
    __device__ __forceinline__ u32 peek16() const { return *(const u16 *)(rings + ra(trc_lane(), rpos & (TRC_SRING - 1))); }
    __device__ __forceinline__ u32 peek32() const { return *(const u32 *)(rings + ra(trc_lane(), rpos & (TRC_SRING - 1))); }
    __device__ __forceinline__ u32 avail() const { return lbytes - rpos; }
    __device__ __forceinline__ void skip_if(bool take) { rpos += take ? 4u : 0u; }

    // first fill: 128 bytes per live lane, fetched by QUADS (round 3): helper quad q takes lanes q, 16 + q, 32 + q, 48 + q in
    // turn, its four lanes 16 bytes each of a 64-byte segment -- the texture-address unit sees one 64-byte request where the
    // per-lane form (every lane its own eight 16-byte loads at 2-byte alignment) gave it four or more: 6.3 -> ~2 us at the
    // start of every decoder wave (ablation in profiles/r03_notes.md).  Eight loads in flight, one round trip.
    __device__ __forceinline__ void prime(bool alive)
    {
        rpos = 0; lbytes = 0; infl = 0; mineA = mineB = false; hokA = hokB = false;
        rw = trc_lds_addr(rings);
        {
            const u32 blo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)soff), bhi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(soff >> 32));
            wbase = ((u64)bhi << 32) | blo;
            srel = (u32)(soff - wbase);
        }
        hvA = hvB = make_uint4(0, 0, 0, 0); hdA = hdB = 0;
        const u64 amask = __ballot(alive);
        if (!amask) return;
        const u32 lane = trc_lane(), q = lane >> 2, part = lane & 3u;
        uint4 v[8];
        u32 at[4];
#pragma unroll
        for (u32 g = 0; g < 4; g++) {
            const u32 j = g * 16u + q;                          // the lane this quad serves in turn g
            const u32 sr = (u32)__shfl((int)srel, (int)j, 64);
            at[g] = ((amask >> j) & 1u) ? rw + ra(j, 0) + part * PIECE_STEP : ~0u;
            const u8 *src = gbase + wbase + sr + (part << 4);
            v[2 * g] = make_uint4(0, 0, 0, 0); v[2 * g + 1] = v[2 * g];
            if (at[g] != ~0u) { v[2 * g] = trc_ld16_a2(src); v[2 * g + 1] = trc_ld16_a2(src + TRC_SEG); }
        }
#pragma unroll
        for (u32 g = 0; g < 4; g++)
            if (at[g] != ~0u) {
                put_piece(at[g] | (part == 0u ? 1u : 0u), v[2 * g]);               // ring offset 0: its first dword is mirrored behind the ring
                put_piece(at[g] + (ra(0, TRC_SEG) - ra(0, 0)), v[2 * g + 1]);      // ring offset 64
            }
        if (alive) lbytes = TRC_SRING;
    }
    // Round 5: the first fill in two steps.  At the start of a decoder all waves of the launch ask for the first 128 bytes of all
    // their streams at once -- 100 MB at chunk 512: 25 MB in one burst -- and, in-kernel clocks say (profiles/r05_notes.md), wait
    // ~8 us for it.  prime_issue() requests the FIRST halves of all rings, then the second halves (loads return in order), and
    // lands nothing; prime_land(P, 0) needs only the first four to be back, the wave decodes its first period on 64 bytes per ring
    // (<= 32 are consumed), prime_land(P, 1) lands the rest.  Between the two the second half counts as a segment in flight
    // (infl = 1), so the refill protocol neither asks for it again nor takes its ring slot.
    struct Prime { uint4 v[8]; u32 at[4]; };
    __device__ __forceinline__ void prime_issue(bool alive, Prime &P)
    {
        rpos = 0; lbytes = 0; infl = 0; mineA = mineB = false; hokA = hokB = false;
        rw = trc_lds_addr(rings);
        {
            const u32 blo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)soff), bhi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(soff >> 32));
            wbase = ((u64)bhi << 32) | blo;
            srel = (u32)(soff - wbase);
        }
        hvA = hvB = make_uint4(0, 0, 0, 0); hdA = hdB = 0;
        const u64 amask = __ballot(alive);
        const u32 lane = trc_lane(), q = lane >> 2, part = lane & 3u;
        const u8 *src[4];
#pragma unroll
        for (u32 g = 0; g < 4; g++) {
            const u32 j = g * 16u + q;                          // the lane this quad serves in turn g
            const u32 sr = (u32)__shfl((int)srel, (int)j, 64);
            const bool on = (amask >> j) & 1u;
            P.at[g] = on ? rw + ra(j, 0) + part * PIECE_STEP : ~0u;
            src[g] = gbase + wbase + (on ? sr : 0u) + (part << 4);      // (a turn with nobody to serve re-reads the wave's first stream: every load is issued, the compiler counts them)
        }
#pragma unroll
        for (u32 g = 0; g < 4; g++) P.v[2 * g] = trc_ld16_a2(src[g]);
#pragma unroll
        for (u32 g = 0; g < 4; g++) P.v[2 * g + 1] = trc_ld16_a2(src[g] + TRC_SEG);
    }
    __device__ __forceinline__ void prime_land(const Prime &P, int half, bool alive)
    {
        const u32 part = trc_lane() & 3u;
#pragma unroll
        for (u32 g = 0; g < 4; g++)
            if (P.at[g] != ~0u) {
                if (half == 0) put_piece(P.at[g] | (part == 0u ? 1u : 0u), P.v[2 * g]);      // ring offset 0: its first dword is mirrored behind the ring
                else put_piece(P.at[g] + (ra(0, TRC_SEG) - ra(0, 0)), P.v[2 * g + 1]);       // ring offset 64
            }
        if (alive) { if (half == 0) { lbytes = TRC_SEG; infl = 1; } else { lbytes = TRC_SRING; infl -= 1; } }
    }
    // Round 3: a refill round costs ONE cross-lane round trip and the landing of a piece one address operation.
    //   * the up to 16 lanes picked in a round PUSH what their helpers need -- the source offset of the segment and the LDS
    //     address of its ring slot -- to the leader of helper quad `rank` with two ds_permute_b32 (lanes not picked push to
    //     lane 1, which leads no quad); the leader's values reach its quad by DPP.  Round 2 wrote the picked lanes' numbers to
    //     an LDS table, read it back and fetched the two values with two shuffles: two dependent LDS round trips per period;
    //   * the helper keeps the piece's final LDS address (bit 0: the piece opens the ring, its first dword is mirrored behind
    //     the ring); landing it is four stores at constant offsets -- rows of the interleaved ring are 256 bytes apart -- where
    //     round 2 rebuilt four addresses from (lane, offset) with some twenty VALU instructions.
    static constexpr u32 PIECE_STEP = IL ? 1024u : 16u;        // LDS distance of consecutive 16-byte pieces of a segment
    static constexpr u32 GUARD_OFF = IL ? 8192u : TRC_SRING;   // from a ring's first dword to its mirror
    typedef __attribute__((address_space(3))) u32 lds_u32;
    __device__ __forceinline__ void put_piece(u32 hd, uint4 v)
    {
        const u32 a = hd & ~3u;
        if (IL) {
            *(lds_u32 *)(uintptr_t)a = v.x;          *(lds_u32 *)(uintptr_t)(a + 256u) = v.y;
            *(lds_u32 *)(uintptr_t)(a + 512u) = v.z; *(lds_u32 *)(uintptr_t)(a + 768u) = v.w;
        } else {
            *(lds_u32 *)(uintptr_t)a = v.x;        *(lds_u32 *)(uintptr_t)(a + 4u) = v.y;
            *(lds_u32 *)(uintptr_t)(a + 8u) = v.z; *(lds_u32 *)(uintptr_t)(a + 12u) = v.w;
        }
        if (hd & 1u) *(lds_u32 *)(uintptr_t)(a + GUARD_OFF) = v.x;   // guard: a reader may take what lies at ring offset p and just behind it from one address
    }
    // land the round that travels in set `par` (requested two periods ago)
    __device__ __forceinline__ void commit(int par)
    {
        if (par == 0) {
            if (hokA) { put_piece(hdA, hvA); hokA = false; }
            if (mineA) { lbytes += TRC_SEG; infl--; mineA = false; }
        } else {
            if (hokB) { put_piece(hdB, hvB); hokB = false; }
            if (mineB) { lbytes += TRC_SEG; infl--; mineB = false; }
        }
    }
    // request one more segment for up to 16 lanes whose ring has room, into register set `par`
    __device__ __forceinline__ void refill(bool alive, int par)
    {
        const u32 lane = trc_lane();
        const bool needy = alive && avail() + TRC_SEG * infl <= TRC_SEG;
        const u64 mask = __ballot(needy);
        if (!mask) return;
        const u32 rank = trc_mbcnt(mask);
        const bool pick = needy && rank < 16u;
        const u32 cnt = (u32)__popcll(mask);
        const u32 nx = lbytes + TRC_SEG * infl;                 // stream offset of this lane's next segment
        const u32 ro = nx & (TRC_SRING - 1);                    // its ring offset: 0 or 64
        const u32 src = srel + trc_min(nx, lim);                // where the bytes come from, relative to the wave's first stream
        const u32 slot = (rw + ra(lane, ro)) | (ro == 0u ? 1u : 0u);
        const int dst = (int)((pick ? rank << 2 : 1u) << 2);    // byte index of the receiving lane
        const u32 src_l = (u32)__builtin_amdgcn_ds_permute(dst, (int)src), slot_l = (u32)__builtin_amdgcn_ds_permute(dst, (int)slot);
        const u32 src_q = (u32)__builtin_amdgcn_update_dpp(0, (int)src_l, 0x00, 0xf, 0xf, false);     // quad_perm [0,0,0,0]: the leader's value
        const u32 slot_q = (u32)__builtin_amdgcn_update_dpp(0, (int)slot_l, 0x00, 0xf, 0xf, false);
        const u32 q = lane >> 2, part = lane & 3u;
        if (q < cnt && q < 16u) {
            const uint4 v = trc_ld16_a2(gbase + wbase + src_q + (part << 4));
            const u32 dd = ((slot_q & ~1u) + part * PIECE_STEP) | (part == 0u ? slot_q & 1u : 0u);
            if (par == 0) { hvA = v; hdA = dd; hokA = true; } else { hvB = v; hdB = dd; hokB = true; }
        }
        if (pick) { infl++; if (par == 0) mineA = true; else mineB = true; }
    }
    // emergency: some lane is about to run dry.  Land everything in flight (older set first: at this
    // point only set !par can still be in flight), then fill every ring synchronously.
    __device__ __forceinline__ void sync_refill(bool alive, int par)
    {
        commit(par ^ 1);
        for (;;) {
            const bool needy = alive && avail() <= TRC_SEG;
            if (!__ballot(needy)) break;
            refill(alive, 0);
            commit(0);
        }
    }
    // the per-period protocol every decoder runs at a uniform point (<= 32 bytes consumed per period)
    __device__ __forceinline__ void period(bool alive, int par)
    {
        commit(par);
        if (__ballot(alive && avail() < 36u)) sync_refill(alive, par);
        refill(alive, par);
    }
};
typedef StreamInT<false> StreamIn;
