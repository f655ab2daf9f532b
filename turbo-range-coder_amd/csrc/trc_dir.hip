// trc_dir.hip -- everything around the coders: CDF-derived tables, the chunk directory
// (per-group sums -> exclusive scan), the payload gather, and cdfini (histogram -> CDF) on device.
#include <stdlib.h>
#include "trc_dev.h"
#include "trc_gather.h"
#include "trc_launch.h"

// ---------------------------------------------------------------------------------------------
// Static-coder tables.  grid = 128 x 256 threads = one thread per 15-bit slot.
//   enc[x] = { m, (32768-f) | (l-1)<<24, f<<16, c0 }   with  q = umulhi(st, m) >> (l-1) == st / f  for st < 2^31
//            (m = floor(2^(31+l)/f) + 1, l = ceil(log2 f): exact round-up reciprocal; f = 1 special-cased)
//   dec[x] = f<<16 | c0
//   lut[s] = largest x with cdf[x] <= s            (replaces the reference's per-symbol CDF search)
__global__ __launch_bounds__(256) void trc_static_prep_kernel(const u16 *__restrict__ cdf, u32 cdfnum,
                                                              u8 *__restrict__ tables)
{
    __shared__ u32 c[258];
    const u32 tid = threadIdx.x;
    for (u32 i = tid; i < 258; i += 256) c[i] = (i <= cdfnum) ? (u32)cdf[i] : TRC_PROB_ONE;
    __syncthreads();
    const u32 slot = blockIdx.x * 256 + tid;
    u32 x = 0, hi = 256;
    while (x + 1 < hi) { u32 mid = (x + hi) >> 1; if (c[mid] > slot) hi = mid; else x = mid; }
    tables[TRC_TAB_LUT + slot] = (u8)x;
    if (blockIdx.x == 0) {
        const u32 c0 = c[tid], f = c[tid + 1] - c0;
        uint4 e; u32 d;
        if (f == 0) {                       // symbol outside the alphabet: never emits, state untouched
            e = make_uint4(0u, 0u, 0xffffffffu, 0u); d = 1u << 16;
        } else if (f == 1) {                // umulhi(st, 2^32-1) = st-1:  st + c0 + (2^15-1) + (st-1)(2^15-1) = st*2^15 + c0
            e = make_uint4(0xffffffffu, TRC_PROB_ONE - 1u, 1u << 16, c0 + TRC_PROB_ONE - 1u); d = (1u << 16) | c0;
        } else {
            const u32 l = 32u - (u32)__clz((int)(f - 1));            // ceil(log2 f) >= 1
            const u32 m = (u32)((((u64)1) << (31 + l)) / f) + 1u;     // q = (st*m) >> (31+l), exact for st < 2^31
            e = make_uint4(m, (TRC_PROB_ONE - f) | ((l - 1u) << 24), f << 16, c0);
            d = (f << 16) | c0;
        }
        ((uint4 *)(tables + TRC_TAB_ENC))[tid] = e;
        ((u32 *)(tables + TRC_TAB_DEC))[tid] = d;
        for (u32 i = tid; i < 260; i += 256) ((u16 *)(tables + TRC_TAB_CDF))[i] = (u16)((i <= cdfnum) ? cdf[i] : TRC_PROB_ONE);
        // the sync area of the encoders that gather their own payload (trc_gather.h) starts out zero; each such launch leaves it zero
        for (u32 i = tid; i < TRC_SYNC_BYTES / 4u; i += 256) ((u32 *)(tables + TRC_TAB_SYNC))[i] = 0u;
    }
}

void trc_launch_static_prep(const uint16_t *d_cdf, unsigned cdfnum, uint8_t *tables, hipStream_t s)
{
    hipLaunchKernelGGL(trc_static_prep_kernel, dim3(128), dim3(256), 0, s, d_cdf, (u32)cdfnum, tables);
}

// ---------------------------------------------------------------------------------------------
// Directory: gsum[g] = sum of clen over the 64 chunks of group g
__global__ __launch_bounds__(256) void trc_group_sums_kernel(const u32 *__restrict__ clen, u32 nchunks, u64 n, u32 chunk,
                                                             u32 *__restrict__ gsum)
{
    const u32 c = blockIdx.x * 256 + threadIdx.x;
    const u64 cstart = (u64)c * chunk;
    const u32 len = c < nchunks ? (u32)((n - cstart) < chunk ? (n - cstart) : chunk) : 0u;
    const u32 v = trc_wave_sum(c < nchunks ? trc_min(clen[c], len) : 0u);      // same clamp as the decoders
    if (trc_lane() == 0 && (c >> 6) < ((nchunks + 63) >> 6)) gsum[c >> 6] = v;
}
void trc_launch_group_sums(const uint32_t *d_clen, uint32_t nchunks, size_t n, uint32_t chunk, uint32_t *gsum, hipStream_t s)
{
    hipLaunchKernelGGL(trc_group_sums_kernel, dim3((nchunks + 255) / 256), dim3(256), 0, s, d_clen, nchunks, (u64)n, chunk, gsum);
}

// exclusive scan of gsum -> goff (u64), single workgroup (ngroups = nchunks/64 is small: 382 for
// 100 MB / 4 KiB, 32 K for 8 GB)
__global__ __launch_bounds__(1024) void trc_scan_groups_kernel(const u32 *__restrict__ gsum, u32 ngroups,
                                                               u64 *__restrict__ goff, u64 *__restrict__ total)
{
    __shared__ u32 wsum[16];
    __shared__ u64 base_s;
    const u32 tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (u32 t0 = 0; t0 < ngroups; t0 += 1024) {
        const u32 i = t0 + tid;
        const u32 v = i < ngroups ? gsum[i] : 0u;
        const u32 inc = trc_wave_incl_scan(v);
        if (lane == 63) wsum[wid] = inc;
        __syncthreads();
        u32 wbase = 0;
        for (u32 k = 0; k < wid; k++) wbase += wsum[k];
        const u64 base = base_s;
        if (i < ngroups) goff[i] = base + wbase + inc - v;
        __syncthreads();
        if (tid == 1023) base_s = base + wbase + inc;
        __syncthreads();
    }
    if (tid == 0) { goff[ngroups] = base_s; if (total) *total = base_s; }
}
void trc_launch_scan_groups(const uint32_t *gsum, uint32_t ngroups, uint64_t *goff, uint64_t *d_total, hipStream_t s)
{
    TRC_LAUNCH_TIMED(trc_scan_groups_kernel, dim3(1), dim3(1024), 0, s, gsum, ngroups, goff, d_total);
}

// Payload gather: one workgroup per group of 64 chunks moves the group's bytes to payload + base as dst-aligned
// 16-byte vectors.  A chunk contributes one PIECE (modes 0/1: its scratch region, start- or end-aligned; or the input
// chunk itself when stored raw) or two (mode 2, the two-stream coders: [4 + len0 bytes at the start of region A]
// [rest at the start of region B]).  A vector belongs to the piece that holds its FIRST byte and a fixed set of
// threads walks the vectors of one piece, so nothing is searched: every load address follows from the prefix table
// in LDS and a thread's loads are independent.  A vector that runs past its piece takes the rest from the next piece
// (second load, merged by byte mask); if even the next piece ends inside it (tiny pieces) it is assembled byte by
// byte.  (History, 100 MB / chunk 512: chunk-after-chunk copy 55 us, per-vector binary search 54 us, this walk 49 us;
// the two-part layout went from a per-chunk wave copy, 152 us, to the same walk.)
#ifndef TRC_GATHER_VPT
#define TRC_GATHER_VPT 6
#endif
#ifndef TRC_GATHER_ATTR
#define TRC_GATHER_ATTR
#endif
template <int PARTS>
__global__ __launch_bounds__(256) TRC_GATHER_ATTR void trc_gather_kernel(const u8 *__restrict__ in, u64 n, u32 chunk, u32 nchunks,
                                                         const u8 *__restrict__ scratch, u32 stride, int from_end,
                                                         const u8 *__restrict__ scratch2, u32 stride2, const u32 *__restrict__ aux,
                                                         const u32 *__restrict__ clen, const u64 *__restrict__ goff,
                                                         const u32 *__restrict__ gsum, u32 ngroups,
                                                         u8 *__restrict__ payload, u64 *__restrict__ total, u64 *__restrict__ goff_out)
{
    constexpr u32 NP = 64u * PARTS;                            // pieces per group
    constexpr u32 TPP = 256u / NP;                             // threads per piece (4 or 2)
    constexpr u32 VPT = TRC_GATHER_VPT;                        // vectors per thread per trip: 4 x 6 x 16 B covers a 384-byte piece in one trip
    __shared__ u32 ex_s[NP + 1];                               // exclusive prefix of the piece lengths
    __shared__ u64 src_s[NP];                                  // where piece p's bytes are
    __shared__ u64 base_s;
    const u32 g = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    if (wid == 1 || blockDim.x == 64) {                        // one wave sums, the next builds the table
        const u64 b = trc_group_base(goff, gsum, g);
        if (lane == 0) { base_s = b; if (goff_out) goff_out[g] = b; }      // (kept for a decode of this directory: TrcWork::goff_area)
    }
    if (wid == 0) {
        const u32 c = g * 64 + lane;
        const u32 l = c < nchunks ? clen[c] : 0u;
        const u64 cstart = (u64)c * chunk;
        const u32 len = c < nchunks ? (u32)((n - cstart) < chunk ? (n - cstart) : chunk) : 0u;
        const bool raw = l == len;
        const u32 inc = trc_wave_incl_scan(l);
        if (PARTS == 1) {
            ex_s[lane] = inc - l;
            src_s[lane] = (u64)(uintptr_t)(raw ? in + cstart : from_end ? scratch + (u64)(c + 1) * stride - l : scratch + (u64)c * stride);
        } else {
            // from_end == 0: [4 + len0 bytes at the start of region A][rest at the start of region B], len0 = u32 at region A;
            // from_end == 3: [la bytes at the start of the region][rest at its END], la = aux[2c]
            // from_end == 4: [la bytes at the END of region A][rest at the END of region B]
            const u8 *a = scratch + (u64)c * stride;
            const u32 la = (raw || c >= nchunks) ? l : from_end >= 3 ? aux[2 * c] : 4u + *(const u32 *)a;   // raw: the input chunk is piece 0, piece 1 is empty
            ex_s[2 * lane] = inc - l; ex_s[2 * lane + 1] = inc - l + la;
            src_s[2 * lane] = (u64)(uintptr_t)(raw ? in + cstart : from_end == 4 ? scratch + (u64)(c + 1) * stride - la : a);
            src_s[2 * lane + 1] = (u64)(uintptr_t)(from_end == 3 ? scratch + (u64)(c + 1) * stride - (l - la) :
                                                   from_end == 4 ? scratch2 + (u64)(c + 1) * stride2 - (l - la) : scratch2 + (u64)c * stride2);
        }
        if (lane == 63) ex_s[NP] = inc;
    }
    __syncthreads();
    const u64 base = base_s;
    const u32 tot = ex_s[NP];
    if (total && g == ngroups - 1 && tid == 0) *total = base + tot;
    u8 *dst0 = payload + base;
    auto src_of = [&](u32 p) -> const u8 * { return (const u8 *)(uintptr_t)src_s[p]; };

    u32 head = (u32)((16u - ((uintptr_t)dst0 & 15u)) & 15u);
    if (head > tot) head = tot;
    const u32 nvec = (tot - head) >> 4;
    const u32 tail0 = head + (nvec << 4);
    for (u32 b = tid; b < head + (tot - tail0); b += 256) {     // bytes before the first / after the last aligned vector
        const u32 d = b < head ? b : tail0 + (b - head);
        u32 lo = 0, hi = NP;
        while (lo + 1 < hi) { const u32 mid = (lo + hi) >> 1; if (ex_s[mid] <= d) lo = mid; else hi = mid; }
        dst0[d] = src_of(lo)[d - ex_s[lo]];
    }
    {
        const u32 k = tid / TPP, sub = tid % TPP;
        const u32 e0 = ex_s[k], e1 = ex_s[k + 1];
        // vectors whose first byte lies in [e0, e1): first index = ceil((e0 - head)/16) (0 if e0 <= head), end likewise from e1
        const u32 v_lo = e0 <= head ? 0u : (e0 - head + 15u) >> 4;
        u32 v_hi = e1 <= head ? 0u : (e1 - head + 15u) >> 4;
        if (v_hi > nvec) v_hi = nvec;
        const u8 *sa = src_of(k);
        const u8 *sb = k + 1 < NP ? src_of(k + 1) : sa;
        const u32 e2 = ex_s[k + 2 > NP ? NP : k + 2];
        const u32 vl = v_hi - 1u;                               // only the piece's last vector can straddle its end
        for (u32 v0 = v_lo + sub; v0 < v_hi; v0 += VPT * TPP) {  // VPT vectors per thread per trip
            uint4 a[VPT], b[VPT];
            u32 d[VPT];
            bool ok[VPT];
#pragma unroll
            for (int j = 0; j < (int)VPT; j++) {
                const u32 v = v0 + TPP * (u32)j;
                ok[j] = v < v_hi;
                d[j] = head + ((ok[j] ? v : v0) << 4);
                a[j] = trc_ld16_a2(sa + (d[j] - e0));
                b[j] = a[j];
            }
#pragma unroll
            for (int j = 0; j < (int)VPT; j++) {
                const u32 v = v0 + TPP * (u32)j;
                if (ok[j] && v == vl && d[j] + 16u > e1 && d[j] + 16u <= e2) b[j] = trc_ld16_a2(sb - (e1 - d[j]));
            }
#pragma unroll
            for (int j = 0; j < (int)VPT; j++) {
                if (!ok[j]) continue;
                const u32 sp = e1 - d[j];                       // bytes of this vector inside piece k (>= 16: all)
                // (write-through stores here -- what took 2 us off the encoders' tails, StreamOut<.., WT> -- cost the gather 11 us:
                // 29.9 -> 41.0; the nontemporal hint 29.8 -> 39.4 and the decoder behind it 60.4 -> 69.5, profiles/r05_notes.md; its
                // stores stay plain)
                if (sp >= 16u) { *(uint4 *)(dst0 + d[j]) = a[j]; continue; }
                if (d[j] + 16u <= e2) {
                    const u32 aw[4] = { a[j].x, a[j].y, a[j].z, a[j].w }, bw[4] = { b[j].x, b[j].y, b[j].z, b[j].w };
                    u32 r[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int nb = (int)sp - 4 * q;
                        const u32 m = nb >= 4 ? 0xffffffffu : nb <= 0 ? 0u : ((1u << (8 * nb)) - 1u);
                        r[q] = (aw[q] & m) | (bw[q] & ~m);
                    }
                    *(uint4 *)(dst0 + d[j]) = make_uint4(r[0], r[1], r[2], r[3]);
                } else {                                       // three or more pieces inside 16 bytes: byte by byte
                    u32 kk = k;
                    const u8 *p = sa;
                    for (u32 q = 0; q < 16; q++) {
                        while (d[j] + q >= ex_s[kk + 1]) { kk++; p = src_of(kk); }
                        dst0[d[j] + q] = p[d[j] + q - ex_s[kk]];
                    }
                }
            }
        }
    }
}
void trc_launch_gather(const uint8_t *d_in, size_t n, uint32_t chunk, const TrcWork &w, int from_end,
                       const uint32_t *d_clen, uint8_t *d_payload, uint64_t *d_total, hipStream_t s)
{
    if (from_end >= 2)                                         // two pieces per chunk: mode 2 (two regions) or 3 (both ends of one region)
        TRC_LAUNCH_TIMED((trc_gather_kernel<2>), dim3(w.ngroups), dim3(256), 0, s, d_in, (u64)n, chunk, w.nchunks,
                           w.scratch, w.stride, from_end >= 3 ? from_end : 0, w.scratch2, w.stride2, w.aux, d_clen, w.goff, w.gsum, w.ngroups,
                           d_payload, w.goff ? (u64 *)nullptr : d_total, w.goff ? (u64 *)nullptr : w.goff_area);
    else
        TRC_LAUNCH_TIMED((trc_gather_kernel<1>), dim3(w.ngroups), dim3(256), 0, s, d_in, (u64)n, chunk, w.nchunks,
                           w.scratch, w.stride, from_end, w.scratch2, w.stride2, w.aux, d_clen, w.goff, w.gsum, w.ngroups,
                           d_payload, w.goff ? (u64 *)nullptr : d_total, w.goff ? (u64 *)nullptr : w.goff_area);
}

// ---------------------------------------------------------------------------------------------
// cdfini on device (reference: rccdf.c:50-68).  Histogram with per-wave LDS privatisation, then
// one wave builds the CDF with the reference's normalisation rule.
// Histogram (round 3): every LANE counts into its own column -- 128 rows (bins 2r | 2r+1 packed as two 16-bit halves of a
// dword) x 64 lanes per wave, 32 KiB of LDS per wave, a workgroup of four waves per CU.  The update is one ds_add_u32 whose 64
// lanes touch 64 different dwords in two conflict-free groups (bank = lane mod 32): no two lanes ever meet on a counter, whatever
// the data (rounds 1-2 used LDS atomics on 1 / 16 shared copies per workgroup: text puts 17 % of all bytes on one symbol and
// the lanes of an instruction serialised on it -- 66 / 56 us for 100 MB).  A lane counts at most 65 520 bytes between two
// reductions, so the halves cannot overflow.
#define TRC_HIST_WAVE_LDS (128u * 64u * 4u)
#define TRC_HIST_ROUND_VECS 4095u                                 // uint4 per lane per round: 65 520 bytes < 2^16
__global__ __launch_bounds__(256) void trc_hist_kernel(const u8 *__restrict__ in, u64 n, u64 *__restrict__ hist)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    typedef __attribute__((address_space(3))) u32 lds_u32;
    const u32 tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    u32 *mine = (u32 *)(smem + wv * TRC_HIST_WAVE_LDS);            // [128 rows][64 lanes]
    u64 *tot = (u64 *)(smem + 4u * TRC_HIST_WAVE_LDS);             // [256] per workgroup
    const u32 col = trc_lds_addr(mine) + lane * 4u;
    tot[tid] = 0;
    const u64 nvec = n >> 4, stride = (u64)gridDim.x * 256;
    const uint4 *v = (const uint4 *)in;
    // four bytes -> four ds_add_u32 on this lane's column: per byte v_bfe (row) + v_lshl_add (address), v_bfe (parity) + v_mad_u32_u24
    // (increment 1 or 0x10000) -- with one wave per SIMD the kernel's time is its instruction count plus whatever memory latency is
    // left exposed
    auto count = [&](u32 w) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u32 a = (__builtin_amdgcn_ubfe(w, 8 * k + 1, 7) << 8) + col;
            const u32 inc = __umul24(__builtin_amdgcn_ubfe(w, 8 * k, 1), 0xffffu) + 1u;
            __hip_atomic_fetch_add((lds_u32 *)(uintptr_t)a, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
    };
    auto count4 = [&](const uint4 q) { count(q.x); count(q.y); count(q.z); count(q.w); };
    for (u64 r0 = 0; r0 == 0 || r0 < nvec; r0 += stride * TRC_HIST_ROUND_VECS) {      // (r0 is uniform: every wave runs every round)
        for (u32 r = lane; r < 128u * 64u; r += 64u) mine[r] = 0;  // own wave's counters (row-major: lanes write consecutive dwords)
        u64 i = r0 + (u64)blockIdx.x * 256 + tid;
        u32 left = TRC_HIST_ROUND_VECS;
        // Round 4: the NEXT four vectors are requested before the current four are counted (round 3 requested four and counted them
        // at once: a memory round trip in front of every 64 bytes, ~24 of them per lane at 100 MB).
        if (left >= 4u && i + 3 * stride < nvec) {
            uint4 q0 = v[i], q1 = v[i + stride], q2 = v[i + 2 * stride], q3 = v[i + 3 * stride];
            left -= 4u; i += 4 * stride;
            for (; left >= 4u && i + 3 * stride < nvec; left -= 4u, i += 4 * stride) {
                const uint4 n0 = v[i], n1 = v[i + stride], n2 = v[i + 2 * stride], n3 = v[i + 3 * stride];
                count4(q0); count4(q1); count4(q2); count4(q3);
                q0 = n0; q1 = n1; q2 = n2; q3 = n3;
            }
            count4(q0); count4(q1); count4(q2); count4(q3);
        }
        for (; left && i < nvec; left--, i += stride) count4(v[i]);
        if (r0 == 0 && blockIdx.x == 0 && wv == 0)                  // the input's last n % 16 bytes, once
            for (u64 t = (nvec << 4) + lane; t < n; t += 64) { const u32 b = in[t]; mine[(b >> 1) * 64u + lane] += (b & 1u) ? 0x10000u : 1u; }
        // reduce this wave's columns: lane l sums rows l and l + 64, walking the columns rotated by l (bank = column mod 32)
        u32 lo0 = 0, hi0 = 0, lo1 = 0, hi1 = 0;
        for (u32 c = 0; c < 64u; c++) {
            const u32 cc = (c + lane) & 63u;
            const u32 a = mine[lane * 64u + cc], b = mine[(lane + 64u) * 64u + cc];
            lo0 += a & 0xffffu; hi0 += a >> 16; lo1 += b & 0xffffu; hi1 += b >> 16;
        }
        __syncthreads();                                            // (first round: orders the zeroing of tot[])
        atomicAdd((unsigned long long *)&tot[2 * lane], (unsigned long long)lo0);
        atomicAdd((unsigned long long *)&tot[2 * lane + 1], (unsigned long long)hi0);
        atomicAdd((unsigned long long *)&tot[2 * (lane + 64)], (unsigned long long)lo1);
        atomicAdd((unsigned long long *)&tot[2 * (lane + 64) + 1], (unsigned long long)hi1);
    }
    __syncthreads();
    // 256 workgroups finish together and each adds up to 256 bins: rotated by the workgroup, so that at any moment they are at
    // different bins (all starting at bin 0 serialises them on one L2 atomic after the other)
    const u32 bin = (tid + 37u * blockIdx.x) & 255u;
    if (tot[bin]) atomicAdd((unsigned long long *)&hist[bin], (unsigned long long)tot[bin]);
}
// Round 4 form: SIXTEEN waves per CU instead of four.  The per-lane columns above are conflict-free but cost 32 KiB per wave: one
// wave per SIMD, whose ~8 500 instructions and every exposed load are the kernel's time (49 us per 100 MB).  Here a wave keeps
// 16 copies of the packed histogram (128 rows x 16 dwords = 8 KiB; lanes l, l + 16, l + 32, l + 48 share copy l & 15): a
// ds_add_u32 of random bytes meets ~2.5-way bank conflicts and the occasional same-counter pair, but four waves per SIMD hide
// both that and the loads.  A copy counts at most 4 x 1023 x 16 = 65 472 bytes between two reductions (no 16-bit overflow).
#define TRC_HIST2_WAVE_LDS (128u * 16u * 4u)
#define TRC_HIST2_WAVES 16u
#define TRC_HIST2_ROUND_VECS 1023u
__global__ __launch_bounds__(1024) void trc_hist2_kernel(const u8 *__restrict__ in, u64 n, u64 *__restrict__ hist, u32 round_vecs)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    typedef __attribute__((address_space(3))) u32 lds_u32;
    const u32 tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    u32 *mine = (u32 *)(smem + wv * TRC_HIST2_WAVE_LDS);           // [128 rows][16 copies]
    u64 *tot = (u64 *)(smem + TRC_HIST2_WAVES * TRC_HIST2_WAVE_LDS);   // [256] per workgroup
    const u32 col = trc_lds_addr(mine) + (lane & 15u) * 4u;
    if (tid < 256u) tot[tid] = 0;
    const u64 nvec = n >> 4, stride = (u64)gridDim.x * 1024;
    const uint4 *v = (const uint4 *)in;
    auto count = [&](u32 w) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u32 a = (__builtin_amdgcn_ubfe(w, 8 * k + 1, 7) << 6) + col;
            const u32 inc = __umul24(__builtin_amdgcn_ubfe(w, 8 * k, 1), 0xffffu) + 1u;
            __hip_atomic_fetch_add((lds_u32 *)(uintptr_t)a, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
    };
    auto count4 = [&](const uint4 q) __attribute__((always_inline)) { count(q.x); count(q.y); count(q.z); count(q.w); };
    for (u64 r0 = 0; r0 == 0 || r0 < nvec; r0 += stride * round_vecs) {
        for (u32 r = lane; r < 128u * 16u; r += 64u) mine[r] = 0;
        u64 i = r0 + (u64)blockIdx.x * 1024 + tid;
        u32 left = round_vecs;
        if (left >= 2u && i + stride < nvec) {                     // two vectors in flight per lane, the next two requested before these are counted
            uint4 q0 = v[i], q1 = v[i + stride];
            left -= 2u; i += 2 * stride;
            for (; left >= 2u && i + stride < nvec; left -= 2u, i += 2 * stride) {
                const uint4 n0 = v[i], n1 = v[i + stride];
                count4(q0); count4(q1);
                q0 = n0; q1 = n1;
            }
            count4(q0); count4(q1);
        }
        for (; left && i < nvec; left--, i += stride) count4(v[i]);
        if (r0 == 0 && blockIdx.x == 0 && wv == 0)                  // the input's last n % 16 bytes, once
            for (u64 t = (nvec << 4) + lane; t < n; t += 64) {
                const u32 b = in[t];
                __hip_atomic_fetch_add((lds_u32 *)(uintptr_t)(col + (b >> 1) * 64u), (b & 1u) ? 0x10000u : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        // reduce this wave's copies: lane l sums rows l and l + 64 over the 16 copies
        u32 lo0 = 0, hi0 = 0, lo1 = 0, hi1 = 0;
#pragma unroll
        for (u32 c4 = 0; c4 < 4u; c4++) {
            const uint4 a = *(const uint4 *)(mine + lane * 16u + c4 * 4u), b = *(const uint4 *)(mine + (lane + 64u) * 16u + c4 * 4u);
            lo0 += (a.x & 0xffffu) + (a.y & 0xffffu) + (a.z & 0xffffu) + (a.w & 0xffffu); hi0 += (a.x >> 16) + (a.y >> 16) + (a.z >> 16) + (a.w >> 16);
            lo1 += (b.x & 0xffffu) + (b.y & 0xffffu) + (b.z & 0xffffu) + (b.w & 0xffffu); hi1 += (b.x >> 16) + (b.y >> 16) + (b.z >> 16) + (b.w >> 16);
        }
        __syncthreads();                                            // (first round: orders the zeroing of tot[])
        atomicAdd((unsigned long long *)&tot[2 * lane], (unsigned long long)lo0);
        atomicAdd((unsigned long long *)&tot[2 * lane + 1], (unsigned long long)hi0);
        atomicAdd((unsigned long long *)&tot[2 * (lane + 64)], (unsigned long long)lo1);
        atomicAdd((unsigned long long *)&tot[2 * (lane + 64) + 1], (unsigned long long)hi1);
    }
    __syncthreads();
    if (tid < 256u) {
        const u32 bin = (tid + 37u * blockIdx.x) & 255u;
        if (tot[bin]) atomicAdd((unsigned long long *)&hist[bin], (unsigned long long)tot[bin]);
    }
}
__global__ __launch_bounds__(256) void trc_cdf_build_kernel(const u64 *__restrict__ hist, u64 n, u32 cdfnum,
                                                            u16 *__restrict__ cdf, int *__restrict__ status)
{
    __shared__ u64 f[256];
    __shared__ u32 pre[257];
    __shared__ int bad;
    const u32 tid = threadIdx.x;
    if (tid == 0) bad = 0;
    u64 v = 0;
    if (tid < cdfnum) { v = (hist[tid] << TRC_PROB_BITS) / n; if (!v) v = 1; }
    f[tid] = v;
    __syncthreads();
    if (tid == 0) {                                    // 256-entry serial pass: max (strict >, lowest index), sum, fix-up
        u64 best = 0, sum = 0; u32 bi = 0;
        for (u32 i = 0; i < cdfnum; i++) { sum += f[i]; if (f[i] > best) { best = f[i]; bi = i; } }
        f[bi] -= sum - TRC_PROB_ONE;
        u32 acc = 0; pre[0] = 0;
        for (u32 i = 0; i < cdfnum; i++) { acc = (acc + (u32)f[i]) & 0xffffu; pre[i + 1] = acc; }   // u16 wrap like cdf_t
    }
    __syncthreads();
    if (tid < cdfnum && pre[tid] >= pre[tid + 1]) atomicOr(&bad, 1);
    if (tid == 0 && pre[cdfnum] != (TRC_PROB_ONE & 0xffffu)) atomicOr(&bad, 1);
    __syncthreads();
    for (u32 i = tid; i <= cdfnum; i += 256) cdf[i] = (u16)pre[i];
    if (tid == 0) *status = bad ? -1 : (int)(n > 0x7fffffffull ? 0x7fffffffull : n);   // (int)inlen like the reference, saturated
}
void trc_launch_hist(const uint8_t *d_in, size_t n, uint64_t *d_hist, hipStream_t s)
{
    (void)hipMemsetAsync(d_hist, 0, 256 * sizeof(uint64_t), s);
    u64 blocks = ((n >> 4) + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 256) blocks = 256;                             // one workgroup (4 x 32 KiB of counters) per CU
    static const int form = getenv("TRC_HIST_FORM") ? atoi(getenv("TRC_HIST_FORM")) : 2;        // 1: the per-lane columns of round 3
    if (form == 2) {
        const size_t sm2 = TRC_HIST2_WAVES * TRC_HIST2_WAVE_LDS + 256u * sizeof(u64);
        u64 b2 = ((n >> 4) + 1023) / 1024;
        b2 = b2 < 1 ? 1 : b2 > 256 ? 256 : b2;                  // one workgroup of 16 waves per CU
        TRC_RAISE_LDS_ONCE(trc_hist2_kernel, sm2);
        // TRC_HIST_ROUND_VECS (test aid): vectors per lane between two reductions, so that a few MB exercise the many-round path
        // that real inputs take only beyond 4.29 GB
        static const u32 rv = getenv("TRC_HIST_ROUND_VECS") ? (u32)atoi(getenv("TRC_HIST_ROUND_VECS")) : TRC_HIST2_ROUND_VECS;
        hipLaunchKernelGGL(trc_hist2_kernel, dim3((u32)b2), dim3(1024), sm2, s, d_in, (u64)n, d_hist,
                           rv >= 1u && rv <= TRC_HIST2_ROUND_VECS ? rv : TRC_HIST2_ROUND_VECS);
        return;
    }
    const size_t sm = 4u * TRC_HIST_WAVE_LDS + 256u * sizeof(u64);
    TRC_RAISE_LDS_ONCE(trc_hist_kernel, sm);                    // per DEVICE (a process-wide flag left a second GPU at the 64 KiB default)
    hipLaunchKernelGGL(trc_hist_kernel, dim3((u32)blocks), dim3(256), sm, s, d_in, (u64)n, d_hist);
}
void trc_launch_cdf_build(const uint64_t *d_hist, size_t n_total, uint16_t *d_cdf, unsigned cdfnum, int32_t *d_status, hipStream_t s)
{
    hipLaunchKernelGGL(trc_cdf_build_kernel, dim3(1), dim3(256), 0, s, d_hist, (u64)n_total, (u32)cdfnum, d_cdf, d_status);
}
void trc_launch_cdfini(const uint8_t *d_in, size_t n, uint16_t *d_cdf, unsigned cdfnum,
                       int32_t *d_status, uint64_t *d_hist, hipStream_t s)
{
    trc_launch_hist(d_in, n, d_hist, s);
    trc_launch_cdf_build(d_hist, n, d_cdf, cdfnum, d_status, s);
}
