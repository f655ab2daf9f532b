// trc_dev.h -- device-side building blocks shared by the gfx950 kernels (wave64 only).
//
// Layout conventions (DESIGN.md "Data layout in HBM"):
//   * one LANE codes one CHUNK (64 chunks per wave); the coder state lives in VGPRs;
//   * symbol/probability tables live in LDS, staged once per workgroup from a tiny global table
//     that a prep kernel derived from the caller's CDF;
//   * chunk bytes and coded streams move as described in trc_io.h (static coders) and trc_lane_io.h
//     (model-bound coders).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint8_t  u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

#define TRC_WAVE        64
#define TRC_PROB_BITS   15u
#define TRC_PROB_ONE    (1u << TRC_PROB_BITS)
#define TRC_ANS_LOW     (1u << 15)   // rANS state lower bound (anscdf_.h:41)

// ---- unaligned global access (payloads are only 2-byte aligned inside the container) -------------
typedef u32 u32_a1 __attribute__((aligned(1)));
typedef u32 u32_a2 __attribute__((aligned(2)));
struct __attribute__((packed, aligned(2))) trc_u128_a2 { u32 x, y, z, w; };

__device__ __forceinline__ uint4 trc_ld16_a2(const u8 *p)
{
    trc_u128_a2 v = *(const trc_u128_a2 *)p;
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ u32 trc_ld32_a2(const u8 *p) { return *(const u32_a2 *)p; }

// streaming 16-byte accesses (chunk bytes: read once by an encoder, written once by a decoder): the nontemporal hint
// keeps them from displacing the lines that are revisited (staged segments, payload lines).  Headline step 227 ->
// 217 us.  NOT for the gather's reads of the staged payloads: with the hint the step is back to 228 us.
typedef u32 trc_v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 trc_ld16_nt(const u8 *p)
{
    const trc_v4u v = __builtin_nontemporal_load((const trc_v4u *)p);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void trc_st16_nt(u8 *p, uint4 q)
{
    const trc_v4u v = { q.x, q.y, q.z, q.w };
    __builtin_nontemporal_store(v, (trc_v4u *)p);
}

// write-through to memory at system scope: for bytes another agent (a copy engine) reads before the kernel has ended.  Inline asm:
// outside the compiler's vmcnt accounting -- the only reader is outside the kernel, and WaveChunks::after_flush waits before it signals.
__device__ __forceinline__ void trc_st16_sys(u8 *p, uint4 q)
{
    const trc_v4u v = { q.x, q.y, q.z, q.w };
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}

// ---- LDS accesses written out by hand (the symbol loops of the static coders) ----------------------------------------
// Not in the compiler's s_waitcnt bookkeeping: a read's destination is valid only after a counted s_waitcnt statement
// that names it "+v" (cdna_hip_programming.md 5.7 form ii); `addr` is an LDS byte address.
__device__ __forceinline__ trc_v4u trc_lds_read128(u32 addr)
{
    trc_v4u v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ void trc_lds_write16(u32 addr, u32 v)
{
    asm volatile("ds_write_b16 %0, %1" :: "v"(addr), "v"(v) : "memory");
}

// LDS accesses by integer byte address (address space 3): the compiler then folds constant parts into the instruction's
// offset field; through a generic pointer into `extern __shared__` memory it adds the (zero) segment base with a VALU op
// per access (v_add_u32 v, 0, v in the round-1 listings of the model-bound coders).
typedef __attribute__((address_space(3))) u8 trc_lds_u8;
typedef __attribute__((address_space(3))) u16 trc_lds_u16;
__device__ __forceinline__ u32 trc_lds_addr(const void *p) { return (u32)(uintptr_t)(const trc_lds_u8 *)p; }   // p must point into LDS
__device__ __forceinline__ u32 trc_ldsr16(u32 a) { return *(const trc_lds_u16 *)(uintptr_t)a; }
__device__ __forceinline__ void trc_ldsw16(u32 a, u32 v) { *(trc_lds_u16 *)(uintptr_t)a = (u16)v; }

typedef __attribute__((address_space(3))) trc_v4u trc_lds_v4u;
__device__ __forceinline__ uint4 trc_ldsr128(u32 a) { const trc_v4u v = *(const trc_lds_v4u *)(uintptr_t)a; return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void trc_ldsw128(u32 a, uint4 q) { const trc_v4u v = { q.x, q.y, q.z, q.w }; *(trc_lds_v4u *)(uintptr_t)a = v; }

// a 16-byte store through an address the compiler cannot trace back to a kernel argument (pointers carried in arrays across loop
// rounds): said to be global here, or it becomes flat_store, which counts on lgkmcnt too and makes every LDS wait a memory wait
typedef __attribute__((address_space(1))) trc_v4u trc_glb_v4u;
__device__ __forceinline__ void trc_gst128(void *p, uint4 q) { const trc_v4u v = { q.x, q.y, q.z, q.w }; *(trc_glb_v4u *)(uintptr_t)p = v; }
typedef __attribute__((address_space(1))) u32 trc_glb_u32;
__device__ __forceinline__ u32 trc_gld32(const void *p) { return *(const trc_glb_u32 *)(uintptr_t)p; }

// LDS-only workgroup barrier (the two-wave encoders: a model wave feeding a coder wave through an LDS queue).
// __syncthreads() also waits for vmcnt(0), i.e. for the coder wave's word stores and the model wave's input loads in
// flight -- a memory round trip per period that nothing there needs.
__device__ __forceinline__ void trc_lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---- pace-keeping among the waves of a SIMD (round 5) -------------------------------------------------------------------
// The SIMD's arbiter serves its OLDEST wave first.  In a launch that is one residency round (100 MB at the bench chunk: every wave
// starts within 0.6 us of the first) the three waves of a SIMD therefore do not advance together: wall clocks per wave of the
// static rANS decoder (profiles/r05_notes.md) -- the first ends at 48 us, the second at 54, the third at 63; for the last 9 us of
// the launch every SIMD runs ONE wave, at a lone wave's issue rate (6-7 cycles per instruction against ~4 per SIMD with three).
// Here every wave publishes a progress counter in LDS ([SIMD][age], 64 bytes per workgroup) at a wave-uniform point of its main
// loop and reads the counters of its SIMD: behind the furthest -> s_setprio 3, in front -> s_setprio 0.  They end together:
// decoder 67 -> 62-63 us.  A workgroup's own waves k, k + 4, k + 8 ... share SIMD k (scripts/probe/residency.hip), so this needs
// workgroups of 8 and more waves; with four or fewer it does nothing.
typedef __attribute__((address_space(3))) u32 trc_lds_u32_t;
struct TrcPace {
    u32 mine, simd;
    // `prog` = LDS byte address of 64 bytes nobody else uses; call before a workgroup barrier
    __device__ __forceinline__ void init(u32 prog, u32 tid, u32 wv)
    {
        if (tid < 16u) *(trc_lds_u32_t *)(uintptr_t)(prog + tid * 4u) = 0u;
        mine = prog + ((wv & 3u) * 4u + ((wv >> 2) & 3u)) * 4u; simd = prog + (wv & 3u) * 16u;
    }
    // `progress` counts up, wave-uniform
    __device__ __forceinline__ void step(u32 progress)
    {
        *(trc_lds_u32_t *)(uintptr_t)mine = progress;
        const uint4 pr = trc_ldsr128(simd);
        const u32 a = pr.x > pr.y ? pr.x : pr.y, b = pr.z > pr.w ? pr.z : pr.w;
        const u32 lead = (u32)__builtin_amdgcn_readfirstlane((int)(a > b ? a : b));
        if (lead > progress) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
    }
};

// (a & m) | (b & ~m) as the one instruction it is.  From the C form the compiler builds and / and-or pairs, compares and selects
// or -- for a group of selects on one condition -- a divergent if / else; in the one-wave-per-SIMD kernels every one of those is slower.
__device__ __forceinline__ u32 trc_bfi(u32 m, u32 a, u32 b)
{
    u32 r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ u32 trc_min(u32 a, u32 b) { return a < b ? a : b; }
__device__ __forceinline__ u32 trc_sub_sat(u32 a, u32 b) { return a > b ? a - b : 0u; }

// ---- wave64 helpers -----------------------------------------------------------------------------
__device__ __forceinline__ u32 trc_lane() { return threadIdx.x & 63u; }

// ---- workgroup shape of the one-wave-per-64-chunks kernels (round 4) -------------------------------------------------
// These kernels hold ONE wave per SIMD (the model fills the LDS) and a launch of ~1000 waves is one residency round, so a
// launch lasts as long as its SLOWEST wave -- and a wave that shares its SIMD with another runs ~1.4x longer.  Which SIMD a
// wave lands on is the dispatcher's business: with one-wave workgroups it deals them round-robin from wherever the launch
// before left off, and scripts/probe/residency.hip found a fresh launch of 1018 such workgroups with 105 SIMDs holding two
// waves and 111 holding none (all 1018 on their own SIMD only right behind a launch of the same shape).  A workgroup's
// OWN waves, however, are dealt over consecutive SIMDs: a workgroup of FOUR waves puts one on every SIMD of its CU, one of
// eight puts two, whatever came before (measured: 1012 of 1024 SIMDs at exactly two waves, 12 at one).  So these kernels
// run as workgroups of TRC_WPG = 4 independent waves (no barrier between them; each has its own slice of the dynamic LDS),
// one workgroup per CU; the two-wave encoders as workgroups of 8 (waves 0-3 model, 4-7 coder: SIMD k gets pair k).
#define TRC_WPG 4u
#define TRC_QUAD_GRID(ngroups) dim3(((ngroups) + TRC_WPG - 1u) / TRC_WPG)
// kernel prologue: this wave's 64-chunk group `grp_`, its LDS slice `smem`, `lane`; waves without a group leave at once
// Round 5: the workgroup may also be LARGER than TRC_WPG waves (the nibble coders in a one-round launch: twelve waves, so that the
// waves of a SIMD sit in one workgroup and keep each other's pace: TrcPace below; the launch then asks for 64 bytes behind the
// waves' slices) -- the prologue takes the shape from blockDim; TRC_PACE_STEP(p) at a wave-uniform point of the main loop.
#define TRC_QUAD_PROLOGUE(WAVE_LDS_BYTES)                                                               \
    extern __shared__ __attribute__((aligned(16))) u8 smem_wg_[];                                      \
    const u32 wv_ = (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));                      \
    const u32 wpg_ = blockDim.x >> 6;                                                                   \
    TrcPace pace_;                                                                                      \
    if (wpg_ > TRC_WPG) { pace_.init(trc_lds_addr(smem_wg_) + wpg_ * (u32)(WAVE_LDS_BYTES), threadIdx.x, wv_); __syncthreads(); } \
    const u32 grp_ = blockIdx.x * wpg_ + wv_;                                                           \
    if (grp_ >= (nchunks + 63u) / 64u) return;                                                          \
    u8 *const smem = smem_wg_ + wv_ * (u32)(WAVE_LDS_BYTES);                                            \
    const u32 lane = trc_lane()
#define TRC_PACE_STEP(p) do { if (wpg_ > TRC_WPG) pace_.step(p); } while (0)
#define TRC_NIB_WPG 12u                                        // the nibble coders' large workgroup
// orders a wave's own LDS stores before its later loads of what OTHER lanes stored (the hardware executes a wave's LDS
// instructions in order; this keeps the compiler from moving them)
__device__ __forceinline__ void trc_wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// inclusive prefix sum over the 64 lanes of a wave (all lanes must call)
__device__ __forceinline__ u32 trc_wave_incl_scan(u32 v)
{
    const u32 lane = trc_lane();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        u32 t = __shfl_up(v, d, 64);
        if (lane >= (u32)d) v += t;
    }
    return v;
}
__device__ __forceinline__ u32 trc_wave_sum(u32 v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// ---- cooperative byte copy by one wave (dst/src arbitrary alignment) ------------------------------
// dst-aligned 16-byte stores in the middle, byte granularity at both ends.
__device__ __forceinline__ void trc_wave_copy(u8 *dst, const u8 *src, u32 len)
{
    const u32 lane = trc_lane();
    if (((uintptr_t)src | (uintptr_t)dst) & 1u) {              // never on the product path (offsets are even)
        for (u32 i = lane; i < len; i += 64) dst[i] = src[i];
        return;
    }
    u32 head = (u32)((16u - ((uintptr_t)dst & 15u)) & 15u);
    if (head > len) head = len;
    if (lane < head) dst[lane] = src[lane];
    const u32 nvec = (len - head) >> 4;
    u8 *d = dst + head; const u8 *s = src + head;
    for (u32 i = lane; i < nvec; i += 64)
        *(uint4 *)(d + (size_t)i * 16) = trc_ld16_a2(s + (size_t)i * 16);
    const u32 done = head + (nvec << 4);
    if (lane < len - done) dst[done + lane] = src[done + lane];
}

// The chunks a wave's lanes hold RAW (rawmask: stored length == chunk length), copied to their places in `out`: four at a time,
// one per 16-lane group (the copy of one 512-byte chunk is a memory round trip with half a wave's lanes busy; a wave of mixed
// data has a dozen of them, and one after the other they were a quarter of the decoder's time on `mix100m`).
// off / len: this lane's payload offset and chunk length; out_wave = out + first chunk of the wave * chunk.
__device__ __forceinline__ void trc_wave_copy_raw(u64 rawmask, u64 off, u32 len, u8 *out_wave, u32 chunk, const u8 *payload)
{
    const u32 lane = trc_lane(), grp = lane >> 4, gl = lane & 15u;
    if (((uintptr_t)out_wave | chunk) & 15u) {                  // (unaligned output: the general copy, one chunk at a time)
        while (rawmask) {
            const int k = __ffsll((long long)rawmask) - 1;
            rawmask &= rawmask - 1;
            const u32 olo = (u32)__shfl((int)(u32)off, k, 64), ohi = (u32)__shfl((int)(u32)(off >> 32), k, 64);
            const u32 l = (u32)__shfl((int)len, k, 64);
            trc_wave_copy(out_wave + (u64)(u32)k * chunk, payload + (((u64)ohi << 32) | olo), l);
        }
        return;
    }
    while (rawmask) {
        int mine = -1;                                          // the chunk (lane index) this 16-lane group copies in this trip
#pragma unroll
        for (u32 g = 0; g < 4; g++) {
            const int k = rawmask ? __ffsll((long long)rawmask) - 1 : -1;
            if (rawmask) rawmask &= rawmask - 1;
            if (g == grp) mine = k;
        }
        const int src_lane = mine < 0 ? 0 : mine;
        const u32 olo = (u32)__shfl((int)(u32)off, src_lane, 64), ohi = (u32)__shfl((int)(u32)(off >> 32), src_lane, 64);
        const u32 l = (u32)__shfl((int)len, src_lane, 64);
        if (mine >= 0) {
            u8 *d = out_wave + (u64)(u32)mine * chunk;
            const u8 *sp = payload + (((u64)ohi << 32) | olo);
            const u32 nvec = l >> 4;
            for (u32 i = gl; i < nvec; i += 16) *(uint4 *)(d + (size_t)i * 16) = trc_ld16_a2(sp + (size_t)i * 16);
            const u32 done = nvec << 4;
            if (gl < l - done) d[done + gl] = sp[done + gl];
        }
    }
}

// Payload offset of group g (64 chunks): goff[g] when the caller ran the scan kernel, otherwise the sum
// of the per-group byte counts below g, computed by the calling wave (all 64 lanes must call).
// (A variant with atomically accumulated per-64-group super-sums was measured and dropped: the atomics
// cost more than these loads -- group_sums went from 4 to 18 us.)
__device__ __forceinline__ u64 trc_group_base(const u64 *goff, const u32 *gsum, u32 g)
{
    if (goff) return goff[g];
    u64 acc = 0;
    u32 i = trc_lane();
    for (; i + 448u < g; i += 512u) {                          // 8 independent loads in flight per trip
        const u32 a0 = gsum[i], a1 = gsum[i + 64], a2 = gsum[i + 128], a3 = gsum[i + 192];
        const u32 a4 = gsum[i + 256], a5 = gsum[i + 320], a6 = gsum[i + 384], a7 = gsum[i + 448];
        acc += (u64)a0 + a1 + a2 + a3 + ((u64)a4 + a5 + a6 + a7);
    }
    for (; i < g; i += 64) acc += gsum[i];
    u32 lo = (u32)acc, hi = (u32)(acc >> 32);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const u32 l2 = (u32)__shfl_xor((int)lo, d, 64), h2 = (u32)__shfl_xor((int)hi, d, 64);
        const u64 t = ((((u64)hi) << 32) | lo) + ((((u64)h2) << 32) | l2);
        lo = (u32)t; hi = (u32)(t >> 32);
    }
    return (((u64)hi) << 32) | lo;
}

// u32 -> f32 as the one instruction it is (from `(float)(u32)(x >> 32)` of a 64-bit x the compiler builds a 64-bit conversion)
__device__ __forceinline__ float trc_u2f(u32 v)
{
    float r;
    asm("v_cvt_f32_u32_e32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}

// One rANS step with a run-time divisor (ece, anscdf_.h:90-94): st -> (st / f) << 15 + st % f + c0, computed as
// st + (st / f) * g + c0 with g = 2^15 - f -- only the QUOTIENT has to be exact, no remainder fix-up.  st < 2^31 and st / f <= 2^16,
// so the f32 estimate is within +-1 of it; the sign of st - q * f and the comparison with f say which way (7 operations less
// than correcting quotient and remainder in turn: these passes are bound by instruction issue).
__device__ __forceinline__ u32 trc_rans_step(u32 st, u32 f, u32 g, u32 c0)
{
    const u32 q = (u32)((float)st * __builtin_amdgcn_rcpf((float)f));
    const int rm = (int)(st - __umul24(q, f));
    const u32 qe = q + (u32)(rm >> 31) + (rm >= (int)f ? 1u : 0u);
    return st + __umul24(qe, g) + c0;
}

