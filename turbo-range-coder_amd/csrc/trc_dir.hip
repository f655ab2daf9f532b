// trc_dir.hip -- everything around the coders: CDF-derived tables, the chunk directory
// (per-group sums -> exclusive scan), the payload gather, and cdfini (histogram -> CDF) on device.
#include "trc_dev.h"
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
    }
}

void trc_launch_static_prep(const uint16_t *d_cdf, unsigned cdfnum, uint8_t *tables, hipStream_t s)
{
    hipLaunchKernelGGL(trc_static_prep_kernel, dim3(128), dim3(256), 0, s, d_cdf, (u32)cdfnum, tables);
}

// ---------------------------------------------------------------------------------------------
// Directory: gsum[g] = sum of clen over the 64 chunks of group g
__global__ __launch_bounds__(256) void trc_group_sums_kernel(const u32 *__restrict__ clen, u32 nchunks,
                                                             u32 *__restrict__ gsum)
{
    const u32 c = blockIdx.x * 256 + threadIdx.x;
    const u32 v = trc_wave_sum(c < nchunks ? clen[c] : 0u);
    if (trc_lane() == 0 && (c >> 6) < ((nchunks + 63) >> 6)) gsum[c >> 6] = v;
}
void trc_launch_group_sums(const uint32_t *d_clen, uint32_t nchunks, uint32_t *gsum, hipStream_t s)
{
    hipLaunchKernelGGL(trc_group_sums_kernel, dim3((nchunks + 255) / 256), dim3(256), 0, s, d_clen, nchunks, gsum);
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
    hipLaunchKernelGGL(trc_scan_groups_kernel, dim3(1), dim3(1024), 0, s, gsum, ngroups, goff, d_total);
}

// ---------------------------------------------------------------------------------------------
// Gather: one workgroup (4 waves) per group of 64 chunks; wave w moves chunks w, w+4, ...
__global__ __launch_bounds__(256) void trc_gather_kernel(const u8 *__restrict__ in, u64 n, u32 chunk, u32 nchunks,
                                                         const u8 *__restrict__ scratch, u32 stride, int from_end,
                                                         const u8 *__restrict__ scratch2, u32 stride2,
                                                         const u32 *__restrict__ clen, const u64 *__restrict__ goff,
                                                         u8 *__restrict__ payload)
{
    const u32 g = blockIdx.x, lane = trc_lane(), wid = threadIdx.x >> 6;
    const u32 c_l = g * 64 + lane;
    const u32 l_l = c_l < nchunks ? clen[c_l] : 0u;
    const u32 ex_l = trc_wave_incl_scan(l_l) - l_l;
    const u64 base = goff[g];
    for (u32 k = wid; k < 64; k += 4) {
        const u32 c = g * 64 + k;
        if (c >= nchunks) break;
        const u32 l = __shfl(l_l, k, 64), ex = __shfl(ex_l, k, 64);
        const u64 cstart = (u64)c * chunk;
        const u32 len = (u32)((n - cstart) < chunk ? (n - cstart) : chunk);
        if (l == len) trc_wave_copy(payload + base + ex, in + cstart, l);
        else if (from_end == 2) {
            const u8 *a = scratch + (u64)c * stride;
            const u32 la = 4u + *(const u32 *)a;
            trc_wave_copy(payload + base + ex, a, la);
            trc_wave_copy(payload + base + ex + la, scratch2 + (u64)c * stride2, l - la);
        } else
            trc_wave_copy(payload + base + ex, from_end ? scratch + (u64)(c + 1) * stride - l : scratch + (u64)c * stride, l);
    }
}
void trc_launch_gather(const uint8_t *d_in, size_t n, uint32_t chunk, const TrcWork &w, int from_end,
                       const uint32_t *d_clen, uint8_t *d_payload, hipStream_t s)
{
    hipLaunchKernelGGL(trc_gather_kernel, dim3(w.ngroups), dim3(256), 0, s, d_in, (u64)n, chunk, w.nchunks,
                       w.scratch, w.stride, from_end, w.scratch2, w.stride2, d_clen, w.goff, d_payload);
}

// ---------------------------------------------------------------------------------------------
// cdfini on device (reference: rccdf.c:50-68).  Histogram with per-wave LDS privatisation, then
// one wave builds the CDF with the reference's normalisation rule.
__global__ __launch_bounds__(256) void trc_hist_kernel(const u8 *__restrict__ in, u64 n, u64 *__restrict__ hist)
{
    __shared__ u32 h[4][256];
    const u32 tid = threadIdx.x, wid = tid >> 6;
    for (u32 i = tid; i < 1024; i += 256) (&h[0][0])[i] = 0;
    __syncthreads();
    const u64 nvec = n >> 4;
    const uint4 *v = (const uint4 *)in;
    for (u64 i = (u64)blockIdx.x * 256 + tid; i < nvec; i += (u64)gridDim.x * 256) {
        const uint4 q = v[i];
        const u32 w[4] = { q.x, q.y, q.z, q.w };
#pragma unroll
        for (int k = 0; k < 4; k++) {
            atomicAdd(&h[wid][w[k] & 255], 1u);         atomicAdd(&h[wid][(w[k] >> 8) & 255], 1u);
            atomicAdd(&h[wid][(w[k] >> 16) & 255], 1u); atomicAdd(&h[wid][w[k] >> 24], 1u);
        }
    }
    if (blockIdx.x == 0) for (u64 i = (nvec << 4) + tid; i < n; i += 256) atomicAdd(&h[wid][in[i]], 1u);
    __syncthreads();
    const u32 t = h[0][tid] + h[1][tid] + h[2][tid] + h[3][tid];
    if (t) atomicAdd((unsigned long long *)&hist[tid], (unsigned long long)t);
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
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(trc_hist_kernel, dim3((u32)blocks), dim3(256), 0, s, d_in, (u64)n, d_hist);
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
