// trc_io.h -- wave-cooperative chunk I/O shared by all coder kernels (gfx950, wave64).
//
// One LANE codes one CHUNK, but no lane ever talks to HBM on its own: PMC runs of the first
// version showed ~5 TA cycles per scattered 16-byte lane access and every access becoming its own
// L2 request (profiles/r01_notes.md).  All global traffic is therefore moved in 64-BYTE SEGMENTS by
// QUADS of lanes (4 x 16 B contiguous), through LDS:
//
//   TileIn     uniform-rate input  (chunk bytes at encode):  64 rows x 64 B tile, loaded ahead
//   TileOut    uniform-rate output (chunk bytes at decode):  64 rows x 64 B tile
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
#define TRC_SEL_BYTES    64u      // per-wave scratch for the rank -> lane table

__device__ __forceinline__ u32 trc_mbcnt(u64 mask)   // number of set bits of mask below this lane
{
    return __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u));
}

// Geometry of the 64 chunks one wave owns.
struct WaveChunks {
    u32 c0;        // first chunk of this wave
    u32 rows;      // valid chunks in this wave (1..64; 0 => wave idle)
    u32 chunk;     // nominal chunk bytes
    u32 lastlen;   // length of chunk nchunks-1
    u32 nchunks;
    __device__ __forceinline__ u32 len_of(u32 row) const { return (c0 + row == nchunks - 1) ? lastlen : chunk; }
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
            r[j] = *(const uint4 *)(base + (size_t)row * w.chunk + segoff + part);
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

// ----------------------------------------------------------------------------------- TileOut ---
struct TileOut {
    u8 *tile;
    u8 *base;            // global address of chunk c0 in the output
    __device__ __forceinline__ void put(u32 k, uint4 v) { *(uint4 *)(tile + trc_lane() * TRC_TILE_STRIDE + (k << 4)) = v; }
    // store the tile as the 64-byte segment at offset segoff of every chunk (whole 16-B pieces only)
    __device__ __forceinline__ void flush(const WaveChunks &w, u32 segoff)
    {
        const u32 lane = trc_lane(), part = (lane & 3u) << 4;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const u32 row = (u32)j * 16u + (lane >> 2);
            if (row < w.rows && segoff + part + 16u <= w.len_of(row))
                *(uint4 *)(base + (size_t)row * w.chunk + segoff + part) = *(const uint4 *)(tile + row * TRC_TILE_STRIDE + part);
        }
    }
};

// --------------------------------------------------------------------------------- StreamOut ---
// DOWN = true : units are appended downward from the END of the chunk's scratch region (rANS)
// DOWN = false: upward from the START of the region (range coders)
template <bool DOWN>
struct StreamOut {
    u8 *rings;           // this wave's ring array (LDS)
    u8 *sel;             // this wave's rank->lane table (LDS)
    u8 *scratch;         // global scratch, region of chunk c is [c*stride, (c+1)*stride)
    u32 stride;
    u32 c0;
    u32 wpos;            // bytes appended by this lane so far
    u32 nfl;             // 64-byte segments already moved to the region

    __device__ __forceinline__ u8 *myring() const { return rings + trc_lane() * TRC_SRING_STRIDE; }
    // ring address of the unit that starts at stream position p
    __device__ __forceinline__ u32 roff16(u32 p) const { return DOWN ? ((0u - (p + 2u)) & (TRC_SRING - 1)) : (p & (TRC_SRING - 1)); }
    __device__ __forceinline__ u32 roff32(u32 p) const { return DOWN ? ((0u - (p + 4u)) & (TRC_SRING - 1)) : (p & (TRC_SRING - 1)); }
    // speculative append: the slot of the next unit is always free, so write unconditionally and
    // only advance when `take` (no exec-mask branch in the symbol loop)
    __device__ __forceinline__ void put16_if(bool take, u32 v)
    {
#ifndef TRC_ABL_NOWRITE
        *(u16 *)(myring() + roff16(wpos)) = (u16)v;
#endif
        wpos += take ? 2u : 0u;
    }
    __device__ __forceinline__ void put16(u32 v) { *(u16 *)(myring() + roff16(wpos)) = (u16)v; wpos += 2; }
    __device__ __forceinline__ void put32(u32 v) { *(u32 *)(myring() + roff32(wpos)) = v; wpos += 4; }

    __device__ __forceinline__ u32 pending() const { return wpos - TRC_SEG * nfl; }
    // one lane moving its own oldest segment to the region (bursts only: runs of 0xFFFFFFFF words
    // released by a range-coder carry can exceed what a period may append)
    __device__ __forceinline__ void self_drain()
    {
        const u32 ro = DOWN ? ((0u - TRC_SEG * (nfl + 1u)) & (TRC_SRING - 1)) : ((TRC_SEG * nfl) & (TRC_SRING - 1));
        const u32 *s = (const u32 *)(myring() + ro);
        u8 *reg = scratch + (size_t)(c0 + trc_lane()) * stride;
        u8 *d = DOWN ? reg + stride - (size_t)TRC_SEG * (nfl + 1u) : reg + (size_t)TRC_SEG * nfl;
        for (int i = 0; i < 4; i++) ((uint4 *)d)[i] = make_uint4(s[4 * i], s[4 * i + 1], s[4 * i + 2], s[4 * i + 3]);
        nfl++;
    }
    __device__ __forceinline__ void put32_slow(u32 v)
    {
        if (pending() + 4u > TRC_SRING - 4u && (size_t)TRC_SEG * (nfl + 2u) <= stride) self_drain();
        put32(v);
    }

    // Move finished segments to HBM.  Called by the whole wave at uniform points.
    // final = false: lanes holding >= 64 pending bytes;  final = true: every lane with pending bytes
    // (the partial segment is written as a whole 64 B; the surplus lands in the region's slack).
    __device__ __forceinline__ void drain(bool final, bool alive)
    {
        const u32 lane = trc_lane();
        bool ready = alive && (final ? pending() > 0 : pending() >= TRC_SEG);
        u64 mask = __ballot(ready);
        while (mask) {
            const u32 rank = trc_mbcnt(mask);
            const bool pick = ready && rank < 16u;
            if (pick) sel[rank] = (u8)lane;
            const u32 cnt = (u32)__popcll(mask);
            const u32 q = lane >> 2, part = (lane & 3u) << 4;
            const u32 j = sel[q];
            const u32 nfl_j = (u32)__shfl((int)nfl, (int)j, 64);
            if (q < cnt && q < 16u) {
                const u32 ro = DOWN ? ((0u - TRC_SEG * (nfl_j + 1u)) & (TRC_SRING - 1)) : ((TRC_SEG * nfl_j) & (TRC_SRING - 1));
                const u32 *s = (const u32 *)(rings + j * TRC_SRING_STRIDE + ro + part);
                u8 *reg = scratch + (size_t)(c0 + j) * stride;
                u8 *d = DOWN ? reg + stride - (size_t)TRC_SEG * (nfl_j + 1u) + part : reg + (size_t)TRC_SEG * nfl_j + part;
                *(uint4 *)d = make_uint4(s[0], s[1], s[2], s[3]);
            }
            if (pick) { nfl++; ready = final ? (wpos > TRC_SEG * nfl) : pending() >= TRC_SEG; }
            mask = __ballot(ready);
        }
    }
};

// ---------------------------------------------------------------------------------- StreamIn ---
struct StreamIn {
    u8 *rings;           // this wave's ring array (LDS)
    u8 *sel;
    const u8 *gbase;     // payload base (kernel argument: keeps the loads in the global address space)
    u64 soff;            // this lane's stream start, bytes from gbase (2-byte aligned)
    u32 rpos;            // bytes consumed
    u32 lbytes;          // bytes committed to the ring
    bool pend;           // this lane has a segment in flight
    // helper-side state of the round in flight (this lane moves one 16-B piece for lane j)
    uint4 hv; u32 hdst; bool hvalid;

    __device__ __forceinline__ const u8 *myring() const { return rings + trc_lane() * TRC_SRING_STRIDE; }
    __device__ __forceinline__ u32 peek16() const { return *(const u16 *)(myring() + (rpos & (TRC_SRING - 1))); }
    __device__ __forceinline__ u32 peek32() const { return *(const u32 *)(myring() + (rpos & (TRC_SRING - 1))); }
    __device__ __forceinline__ u32 avail() const { return lbytes - rpos; }

    // first fill: every live lane's first 128 bytes, 16 lanes per round, synchronous
    __device__ __forceinline__ void prime(bool alive)
    {
        rpos = 0; lbytes = 0; pend = false; hvalid = false;
        for (int rep = 0; rep < 2; rep++) {
            refill(alive, 1u << 30, true);
            commit();
        }
    }
    // land the round in flight (uniform point)
    __device__ __forceinline__ void commit()
    {
        if (hvalid) {
            u32 *d = (u32 *)(rings + hdst);
            d[0] = hv.x; d[1] = hv.y; d[2] = hv.z; d[3] = hv.w;
            hvalid = false;
        }
        if (pend) { lbytes += TRC_SEG; pend = false; }
    }
    // request one more segment for lanes whose ring has room (avail <= 64): up to 16 lanes per
    // round.  all = true drains every needy lane (synchronous rounds: used for priming/emergencies).
    __device__ __forceinline__ void refill(bool alive, u32 thresh, bool all)
    {
        const u32 lane = trc_lane();
        bool needy = alive && !pend && avail() <= TRC_SEG && avail() <= thresh;
        u64 mask = __ballot(needy);
        while (mask) {
            const u32 rank = trc_mbcnt(mask);
            const bool pick = needy && rank < 16u;
            if (pick) sel[rank] = (u8)lane;
            const u32 cnt = (u32)__popcll(mask);
            const u32 q = lane >> 2, part = (lane & 3u) << 4;
            const u32 j = sel[q];
            const u32 lb_j = (u32)__shfl((int)lbytes, (int)j, 64);
            const u32 lo = (u32)__shfl((int)(u32)soff, (int)j, 64);
            const u32 hi = (u32)__shfl((int)(u32)(soff >> 32), (int)j, 64);
            if (q < cnt && q < 16u) {
                const u8 *s = gbase + ((((u64)hi) << 32) | lo) + lb_j + part;
                hv = trc_ld16_a2(s);
                hdst = j * TRC_SRING_STRIDE + (lb_j & (TRC_SRING - 1)) + part;
                hvalid = true;
            }
            if (pick) { pend = true; needy = false; }
            if (!all) break;
            commit();
            needy = alive && avail() <= TRC_SEG;
            mask = __ballot(needy);
        }
    }
};
