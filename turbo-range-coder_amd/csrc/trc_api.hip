// trc_api.hip -- the C-ABI of libturborc_hip.so (include/trc_hip.h, include/anscdf.h, include/turborc.h).
//
// Layer 2 (*_dev): enqueue-only, caller-owned device buffers and stream.
// Layer 1 (reference prototypes): host pointers; stage through a process-wide device context,
// run layer 2, wrap the result in the TRC1 container.  There is NO CPU coding path in this
// library: if HIP is unavailable every call fails loudly (stderr + trc_last_error, return 0).
#include <hip/hip_runtime.h>
#include <mutex>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/trc_hip.h"
#include "trc_launch.h"

typedef unsigned short cdf_t;
static const unsigned TRC_PROB_ONE_HOST = 32768u;

// ------------------------------------------------------------------------------------ errors ---
static thread_local char g_err[512];
static int fail(int code, const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    fprintf(stderr, "turborc_hip: ERROR: %s\n", g_err);
    return code;
}
// the same for the other translation units of the library (trc_rccl.hip)
int trc_fail(int code, const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    fprintf(stderr, "turborc_hip: ERROR: %s\n", g_err);
    return code;
}
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(TRC_E_HIP, "%s -> %s", #x, hipGetErrorString(e_)); } while (0)

extern "C" const char *trc_last_error(void) { return g_err; }

extern "C" int trc_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ------------------------------------------------------------------------------------ config ---
// The chunk is the parallel unit AND the price of the format: every chunk costs 8 bytes of coder state and 4 bytes of
// directory, and an adaptive model learns its statistics anew in every chunk (20 MB of `drift` through rccdfenc: 38.7 % stored
// at chunk 512, 30.6 % at 1536, 28.0 % at 4096, 26.9 % at 16 384; one whole-buffer call of the reference 26.7 %).
// Rounds 3-5 picked the chunk of a host-pointer call for the KERNELS (512 below 1 GB: the chip full of short waves) -- on a path
// whose pace is set by PCIe, not by the kernels (the policy optimised the one quantity the caller cannot see and
// gave away the one it can).  Round 6: unless the caller fixes the size (trc_set_chunk, TRC_CHUNK), a host-pointer call takes the
// LARGEST chunk whose one-wave time still hides behind the call's transfer time:
//     wave time(chunk) = trc_wave_ns(codec) x chunk        (a launch lasts one wave's time however few chunks it has, and the
//                                                           slices of a call are coded concurrently: trc_host.inc)
//     budget           = max(1.7 ms, 0.35 x n / 50 GB/s)    (a third of what the link needs for the call: the last slice to arrive
//                                                           still has a wave's time to go, so a call lasts link time + wave time;
//                                                           the floor is what 100 MB are given: rccdf at 4096, 28.0 % stored)
// from the ladder 512 .. 16 384; static coders stop at 4096 (text100m: 63.50 % of payload against 63.35 % whole-buffer -- nothing
// left to gain), the bitwise rANS at one reference block (8192), the order-1 coder never goes below 4096.
// 100 MB: rccdf / rccdfi / anscdf 4096, rcs 2048, static 4096; 1 GB: rccdf / anscdf 16 384, rcs 8192.
#define TRC_MODEL_ROUND_CHUNKS 65536u                         // 256 CUs x 4 waves x 64 lanes
static uint32_t g_chunk = 0;                                  // 0: not read yet;  ~0u: automatic
static bool chunk_ok(uint32_t c) { return c >= TRC_CHUNK_MIN && c <= TRC_CHUNK_MAX && (c % 64u) == 0; }
static inline bool is_static(int codec);
// one wave's time per byte of its chunk, ns, the slower of encode and decode (profiles/r05_all_codecs.txt: kernel time at chunk
// 4096 with the chip a third full / 4096)
static double trc_wave_ns(int codec)
{
    switch (codec) {
    case TRC_ANS4S: return 67;  case TRC_RCS1: return 184; case TRC_RCS2: return 112; case TRC_RCSM: return 230;
    case TRC_RCB: return 592;   case TRC_ANSB: return 635; case TRC_RCA: return 406;  case TRC_RCAI: return 397;
    case TRC_ANSA: return 336;  case TRC_ANSO1: return 884;
    case TRC_RCA4: return 245;  case TRC_RCAI4: return 260; case TRC_ANSA4: return 255;
    case TRC_RCV8: return 746;  case TRC_RCVI8: return 797;
    case TRC_VLCU16: case TRC_VLCV16: case TRC_VLCVZ16: return 391;
    case TRC_VLCU32: case TRC_VLCV32: case TRC_VLCVZ32: return 214;
    case TRC_VLAU16: case TRC_VLAV16: case TRC_VLAVZ16: return 370;
    case TRC_VLAUZ16: return 260;
    case TRC_VLAV32: case TRC_VLAVZ32: return 201;
    }
    return 400;
}
#define TRC_AUTO_CHUNK_MAX 16384u
extern "C" uint32_t trc_auto_chunk_codec(int codec, size_t n)
{
    static const uint32_t ladder[] = { 16384u, 12288u, 8192u, 6144u, 4096u, 3072u, 2048u, 1536u, 1024u, 768u, 512u };
    const double link_ns = 0.35 * (double)n / 50.0;               // 50 GB/s = 50 bytes per ns
    const double budget_ns = link_ns > 1.7e6 ? link_ns : 1.7e6;
    const uint32_t cap = is_static(codec) ? 4096u : codec == TRC_ANSB ? TRC_ANSB_CHUNK_MAX : TRC_AUTO_CHUNK_MAX;
    const uint32_t lo = codec == TRC_ANSO1 ? 4096u : TRC_CHUNK_AUTO_MIN;
    for (uint32_t c : ladder)
        if (c <= cap && (c <= lo || trc_wave_ns(codec) * c <= budget_ns)) return c;
    return lo;
}
extern "C" uint32_t trc_auto_chunk(size_t n) { return trc_auto_chunk_codec(TRC_ANS4S, n); }      // the static coders' rule
// The chunk for a DEVICE-RESIDENT call of n bytes (one launch over the whole input: trc_encode_dev, bench.py).  A launch lasts
// (residency rounds) x (one wave's time, proportional to its chunk), so the input should be a whole number of rounds of the
// coder's resident lanes, barely: the LARGEST chunk (multiple of 64) with ceil(n / chunk) <= k rounds for the smallest k that
// allows it.  Lanes per round: static coders 12 waves per CU (196 608 chunks; the two-lanes-per-chunk `-e45` coder 98 304), a
// model per lane in LDS 4 waves per CU (65 536), the small-model coders (nibble, vnibble, Turbo-VLC) ~20 (327 680).  100 MB:
// 512 / 1024 / 1536 / 512 as measured best in round 3 (profiles/r03_notes.md section 7: 1280 instead of 1536 is a factor 1.9);
// 70 / 120 / 150 / 333 MB: tests/test_gpu_chunk_policy.py.
// Round 5: the cap is TRC_ROUND_CHUNK_MAX = 16 384 bytes, not 4096.  Rounds pack (profiles/r04_notes.md 1): k rounds of chunk c
// cost what ONE round of chunk k c costs -- and the larger chunk pays the model's learning phase and the coder's flush bytes k
// times less often.  1 GB of `rccdf` took four rounds of chunk 3840 (28.0 % stored; one whole-buffer call of the reference:
// 26.7 %); it now takes one round of chunk 15 296.  The kernels take chunks up to 65 536; above 16 KiB the
// ratio has nothing left to gain.  (The bitwise rANS stays within one reference block, the order-1 coder at 4096.)
#define TRC_ROUND_CHUNK_MAX 16384u
static size_t round_chunks(int codec)
{
    if (codec == TRC_RCS2) return 98304u;
    if (is_static(codec)) return 196608u;
    if (codec == TRC_RCA || codec == TRC_RCAI || codec == TRC_ANSA || codec == TRC_RCB || codec == TRC_ANSB) return TRC_MODEL_ROUND_CHUNKS;
    return 327680u;
}
extern "C" uint32_t trc_round_chunk(int codec, size_t n)
{
    if (codec == TRC_ANSO1) return 4096u;                     // (see trc_auto_chunk_codec)
    const size_t rc = round_chunks(codec);
    const size_t cap = codec == TRC_ANSB ? TRC_ANSB_CHUNK_MAX : TRC_ROUND_CHUNK_MAX;
    for (size_t k = 1;; k++) {
        size_t c = (n + rc * k - 1) / (rc * k);               // ceil(n / c) <= rc * k
        c = (c + 63u) & ~(size_t)63u;
        if (c <= cap) return c < TRC_CHUNK_AUTO_MIN ? TRC_CHUNK_AUTO_MIN : (uint32_t)c;
    }
}
extern "C" uint32_t trc_get_chunk(void)
{
    if (!g_chunk) {
        const char *e = getenv("TRC_CHUNK");
        uint32_t c = e ? (uint32_t)strtoul(e, 0, 10) : 0;
        g_chunk = chunk_ok(c) ? c : ~0u;
    }
    return g_chunk == ~0u ? 0u : g_chunk;
}
extern "C" int trc_set_chunk(uint32_t chunk)
{
    if (chunk == 0) { g_chunk = ~0u; return TRC_OK; }         // back to automatic
    if (!chunk_ok(chunk)) return fail(TRC_E_ARG, "chunk %u: must be 0 (automatic) or a multiple of 64 in [%u,%u]", chunk, TRC_CHUNK_MIN, TRC_CHUNK_MAX);
    g_chunk = chunk;
    return TRC_OK;
}

// ------------------------------------------------------------------------------ workspace map ---
// up to this many groups every consumer wave sums the per-group byte counts itself (one scan kernel less
// per call); beyond it a single-workgroup scan kernel runs
#define TRC_INKERNEL_SCAN_MAX 8192u
static inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }
static inline bool is_static(int codec) { return codec == TRC_ANS4S || codec == TRC_RCS1 || codec == TRC_RCS2 || codec == TRC_RCSM; }
static inline bool codec_ok(int codec) { return codec >= TRC_ANS4S && codec <= TRC_RCVI8; }
static inline bool is_vlc(int codec) { return codec >= TRC_VLCU16 && codec <= TRC_VLCVZ32; }
static inline int vlc_variant(int codec) { return (codec - TRC_VLCU16) >> 1; }      // 0 u, 1 v, 2 vz
static inline int vlc_elem(int codec) { return ((codec - TRC_VLCU16) & 1) ? 4 : 2; }
static inline bool is_vla(int codec) { return codec >= TRC_VLAU16 && codec <= TRC_VLAVZ32; }            // ... over rANS
static inline int vla_variant(int codec) { return codec <= TRC_VLAUZ16 ? 0 : 1; }                          // 0 u, 1 v
static inline int vla_zz(int codec) { return (codec - TRC_VLAU16) & 1; }
static inline int vla_elem(int codec) { return codec >= TRC_VLAV32 ? 4 : 2; }
static inline bool two_streams(int codec) { return codec == TRC_RCS2 || codec == TRC_RCAI || codec == TRC_RCAI4 || codec == TRC_RCVI8; }
// second scratch array: RCS2 stream 1 (same stride) or ANSA's record stack (8 B per input byte + one segment)
static inline size_t scratch2_stride(int codec, uint32_t chunk)
{
    if (codec >= TRC_VLAU16 && codec <= TRC_VLAVZ32)            // record stack (8 B per element) + room for the mantissa bytes at the slot's end
        return 8 * (size_t)(chunk / (codec >= TRC_VLAV32 ? 4 : 2)) + chunk + 64;
    if (codec == TRC_RCB) return 256;                            // the bitwise encoder's deepest tree level (128 nodes x u16 per chunk, trc_rc_bit.hip)
    return two_streams(codec) ? chunk + 128 : (codec == TRC_ANSA || codec == TRC_ANSO1) ? 8 * (size_t)chunk : codec == TRC_ANSB ? 16 * (size_t)chunk : codec == TRC_ANSA4 ? 4 * (size_t)chunk : 0;
}

// bytes of the second scratch array (the bitwise coder's rows are indexed by LANE, dead lanes of the last wave included)
static inline size_t scratch2_bytes(int codec, size_t nchunks, uint32_t chunk)
{
    const size_t rows = codec == TRC_RCB ? ((nchunks + 63) & ~(size_t)63) : nchunks;
    return up256(rows * scratch2_stride(codec, chunk) + 256);
}

static uint32_t scratch_stride(int codec, uint32_t chunk)
{
    (void)codec;
    return chunk + 128;                // payload + one period of look-ahead, moved in whole 64-B segments (trc_io.h)
}

extern "C" size_t trc_work_bytes(int codec, size_t n, uint32_t chunk)
{
    if (codec == 0) return 4096;       // cdfini: histogram bins
    if (!chunk_ok(chunk)) return 0;
    const size_t nchunks = (n + chunk - 1) / chunk, ngroups = (nchunks + 63) / 64;
    return up256(TRC_TAB_BYTES) + up256(4 * ngroups) + up256(8 * (ngroups + 1)) +
           up256(nchunks * (size_t)scratch_stride(codec, chunk)) + scratch2_bytes(codec, nchunks, chunk) +
           (codec == TRC_ANSO1 ? up256(nchunks * (size_t)TRC_O1_MODEL_BYTES) : 0) +
           ((codec >= TRC_VLCU16 && codec <= TRC_VLAVZ32) ? up256(nchunks * 8) : 0) + 4096;
}

static int carve(int codec, size_t n, uint32_t chunk, void *d_work, size_t work_bytes, TrcWork &w)
{
    const size_t need = trc_work_bytes(codec, n, chunk);
    if (!need || work_bytes < need) return fail(TRC_E_WORK, "workspace %zu B < required %zu B", work_bytes, need);
    if (((uintptr_t)d_work) & 255) return fail(TRC_E_ARG, "workspace must be 256-byte aligned");
    const size_t nchunks = (n + chunk - 1) / chunk, ngroups = (nchunks + 63) / 64;
    uint8_t *p = (uint8_t *)d_work;
    w.tables = p;               p += up256(TRC_TAB_BYTES);
    w.gsum = (uint32_t *)p;     p += up256(4 * ngroups);
    static const uint32_t scan_max = getenv("TRC_SCAN_MAX") ? (uint32_t)atoi(getenv("TRC_SCAN_MAX")) : TRC_INKERNEL_SCAN_MAX;   // tuning aid
    w.goff_area = (uint64_t *)p;
    w.goff = ngroups > scan_max ? (uint64_t *)p : nullptr;     p += up256(8 * (ngroups + 1));
    w.scratch = p;
    w.stride = scratch_stride(codec, chunk);
    w.stride2 = (uint32_t)scratch2_stride(codec, chunk);
    w.scratch2 = p + up256(nchunks * (size_t)w.stride);
    w.model = w.scratch2 + scratch2_bytes(codec, nchunks, chunk);
    w.aux = (uint32_t *)(w.model + (codec == TRC_ANSO1 ? up256(nchunks * (size_t)TRC_O1_MODEL_BYTES) : 0));
    w.nchunks = (uint32_t)nchunks; w.ngroups = (uint32_t)ngroups;
    return TRC_OK;
}

static int check_common(int codec, size_t n, uint32_t chunk, const uint16_t *d_cdf, unsigned cdfnum)
{
    if (!codec_ok(codec)) return fail(TRC_E_ARG, "codec %d not available", codec);
    if (!chunk_ok(chunk)) return fail(TRC_E_ARG, "chunk %u: must be a multiple of 64 in [%u,%u]", chunk, TRC_CHUNK_MIN, TRC_CHUNK_MAX);
    if ((n + chunk - 1) / chunk > 0x7fffffffu) return fail(TRC_E_ARG, "too many chunks");
    if (codec == TRC_ANSB && chunk > TRC_ANSB_CHUNK_MAX) return fail(TRC_E_ARG, "bitwise rANS: chunk %u exceeds one reference block (%u)", chunk, TRC_ANSB_CHUNK_MAX);
    if (is_static(codec) && (!d_cdf || cdfnum < 1 || cdfnum > 256)) return fail(TRC_E_CDF, "static coder needs a CDF with 1..256 symbols");
    return TRC_OK;
}

// ------------------------------------------------------------------ coder-kernel timing (opt-in) ---
// Event pairs ride on every coder launch of a call made while timing is on (TRC_LAUNCH_TIMED, trc_launch.h): two-pass
// encoders contribute both passes.  The pool is process-wide and guarded by a mutex; which direction a launch belongs
// to is a per-thread flag set around the launches of one trc_encode_dev / trc_decode_dev call.
#define TRC_TM_MAX 4096
static struct TmState {
    std::mutex mu;
    bool on = false;
    int calls[3] = { 0, 0, 0 }, pairs[3] = { 0, 0, 0 };       // 0 encode coder kernels, 1 decode coder kernels, 2 the encode path's scan + gather
    hipEvent_t ev[3][TRC_TM_MAX][2];
    bool made[3][TRC_TM_MAX] = {};
    bool paused = false;
} g_tm;
static thread_local int tm_dir = -1;                            // direction of the call in progress on this thread, -1 = not timing
extern "C" int trc_timing_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_tm.mu);
    g_tm.on = on != 0; g_tm.paused = false;
    for (int d = 0; d < 3; d++) g_tm.calls[d] = g_tm.pairs[d] = 0;
    return TRC_OK;
}
// suspend / resume without touching what has been collected (a caller that times a sample of its calls)
extern "C" int trc_timing_pause(int paused)
{
    std::lock_guard<std::mutex> lk(g_tm.mu);
    g_tm.paused = paused != 0;
    return TRC_OK;
}
bool trc_tm_next(hipEvent_t *start, hipEvent_t *stop)
{
    if (tm_dir < 0) return false;
    std::lock_guard<std::mutex> lk(g_tm.mu);
    const int d = tm_dir, i = g_tm.pairs[d];
    if (!g_tm.on || i >= TRC_TM_MAX) return false;
    if (!g_tm.made[d][i]) {
        if (hipEventCreate(&g_tm.ev[d][i][0]) != hipSuccess || hipEventCreate(&g_tm.ev[d][i][1]) != hipSuccess) return false;
        g_tm.made[d][i] = true;
    }
    *start = g_tm.ev[d][i][0]; *stop = g_tm.ev[d][i][1];
    g_tm.pairs[d] = i + 1;
    return true;
}
static inline void tm_begin(int dec)
{
    std::lock_guard<std::mutex> lk(g_tm.mu);
    tm_dir = (g_tm.on && !g_tm.paused) ? dec : -1;
}
static inline void tm_end(int dec)
{
    if (tm_dir < 0) return;
    tm_dir = -1;
    std::lock_guard<std::mutex> lk(g_tm.mu);
    g_tm.calls[dec]++;
}
// decode = 2: the encode path's own directory work, the group scan (large inputs) and the payload gather.
// total_ms = summed duration of every coder kernel launched by the calls of that direction since enable(1);
// launches = number of CALLS (so total_ms / launches is the coder-kernel time of one encode or decode, all passes)
extern "C" int trc_timing_read(int decode, double *total_ms, int *launches)
{
    const int d = decode == 2 ? 2 : decode ? 1 : 0;
    int np, nc;
    { std::lock_guard<std::mutex> lk(g_tm.mu); np = g_tm.pairs[d]; nc = g_tm.calls[d]; }
    double sum = 0;
    for (int i = 0; i < np; i++) {
        float ms = 0;
        HIPCHK(hipEventSynchronize(g_tm.ev[d][i][1]));
        HIPCHK(hipEventElapsedTime(&ms, g_tm.ev[d][i][0], g_tm.ev[d][i][1]));
        sum += ms;
    }
    if (total_ms) *total_ms = sum;
    if (launches) *launches = nc;
    return TRC_OK;
}
thread_local TrcGate trc_gate_tls = { nullptr, 0 };
bool trc_gate_ok(int codec)
{
    static const bool off = getenv("TRC_HOST_NO_GATE") != nullptr;       // tuning aid / tests: the slice pipeline for every coder
    if (off) return false;
    switch (codec) {
    case TRC_RCA: case TRC_RCAI: return trc_rca_enc_gate_ok();
    case TRC_ANSA: return trc_ansa_enc_gate_ok();
    case TRC_RCB: return trc_rcb_enc_gate_ok();
    }
    return false;
}
thread_local TrcProg trc_prog_tls = { nullptr, nullptr, 0 };
bool trc_prog_ok(int codec)
{
    static const bool off = getenv("TRC_HOST_NO_GATE") != nullptr;
    if (off) return false;
    switch (codec) {
    case TRC_RCA: case TRC_RCAI: return trc_rca_dec_prog_ok();
    case TRC_ANSA: return trc_ansa_dec_prog_ok();
    case TRC_RCB: return trc_rcb_dec_prog_ok();
    }
    return false;
}
bool trc_first_use_on_device(unsigned long long *mask)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return true;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    const unsigned long long bit = 1ull << dev;
    if (*mask & bit) return false;
    *mask |= bit;
    return true;
}

// ------------------------------------------------------------------------------ layer 2: *_dev ---
extern "C" int trc_cdfini_dev(const void *d_in, size_t n, uint16_t *d_cdf, unsigned cdfnum,
                              int32_t *d_status, void *d_work, void *stream)
{
    if (!n || cdfnum < 1 || cdfnum > 256) return fail(TRC_E_ARG, "cdfini: n=%zu cdfnum=%u", n, cdfnum);
    trc_launch_cdfini((const uint8_t *)d_in, n, d_cdf, cdfnum, d_status, (uint64_t *)d_work, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return TRC_OK;
}

extern "C" int trc_hist_dev(const void *d_in, size_t n, uint64_t *d_hist, void *stream)
{
    if (!d_hist) return fail(TRC_E_ARG, "hist: null histogram");
    if (n) trc_launch_hist((const uint8_t *)d_in, n, d_hist, (hipStream_t)stream);
    else HIPCHK(hipMemsetAsync(d_hist, 0, 256 * sizeof(uint64_t), (hipStream_t)stream));
    HIPCHK(hipGetLastError());
    return TRC_OK;
}
extern "C" int trc_cdf_from_hist_dev(const uint64_t *d_hist, size_t n_total, uint16_t *d_cdf, unsigned cdfnum,
                                     int32_t *d_status, void *stream)
{
    if (!n_total || cdfnum < 1 || cdfnum > 256) return fail(TRC_E_ARG, "cdf_from_hist: n=%zu cdfnum=%u", n_total, cdfnum);
    trc_launch_cdf_build(d_hist, n_total, d_cdf, cdfnum, d_status, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return TRC_OK;
}

extern "C" int trc_tables_dev(const uint16_t *d_cdf, unsigned cdfnum, void *d_work, size_t work_bytes, void *stream)
{
    if (!d_cdf || cdfnum < 1 || cdfnum > 256) return fail(TRC_E_CDF, "tables: need a CDF with 1..256 symbols");
    if (!d_work || work_bytes < up256(TRC_TAB_BYTES) || (((uintptr_t)d_work) & 255)) return fail(TRC_E_WORK, "tables: workspace too small or misaligned");
    trc_launch_static_prep(d_cdf, cdfnum, (uint8_t *)d_work, (hipStream_t)stream);   // the table area opens the workspace (carve)
    HIPCHK(hipGetLastError());
    return TRC_OK;
}

extern "C" int trc_encode_dev(int codec, const void *d_in, size_t n, uint32_t chunk,
                              const uint16_t *d_cdf, unsigned cdfnum,
                              uint32_t *d_clen, void *d_payload, uint64_t *d_total,
                              void *d_work, size_t work_bytes, void *stream)
{
    const bool tables_ready = codec & TRC_TABLES_READY;
    codec &= ~TRC_TABLES_READY;
    int rc = check_common(codec, n, chunk, d_cdf, cdfnum);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (((uintptr_t)d_in & 15) || ((uintptr_t)d_clen & 3) || ((uintptr_t)d_payload & 1) || ((uintptr_t)d_total & 7))
        return fail(TRC_E_ARG, "encode: d_in must be 16-byte, d_clen 4-byte, d_payload 2-byte, d_total 8-byte aligned");
    if (n == 0) { HIPCHK(hipMemsetAsync(d_total, 0, 8, s)); return TRC_OK; }
    TrcWork w;
    if ((rc = carve(codec, n, chunk, d_work, work_bytes, w))) return rc;
    if (is_static(codec) && !tables_ready) trc_launch_static_prep(d_cdf, cdfnum, w.tables, s);
    int from_end = 0;
    bool gathered = false;                                       // the coder's own waves have put the payload in place (trc_gather.h)
    tm_begin(0);
    switch (codec) {
    case TRC_ANS4S: gathered = trc_launch_ans4s_enc((const uint8_t *)d_in, n, chunk, w, d_clen, (uint8_t *)d_payload, d_total, s); from_end = 1; break;
    case TRC_RCS1:  trc_launch_rcs_enc(1, (const uint8_t *)d_in, n, chunk, w, d_clen, s); from_end = 0; break;
    case TRC_RCS2:  trc_launch_rcs_enc(2, (const uint8_t *)d_in, n, chunk, w, d_clen, s); from_end = 2; break;
    case TRC_RCSM:  trc_launch_rcs_enc(-1, (const uint8_t *)d_in, n, chunk, w, d_clen, s); from_end = 0; break;
    case TRC_RCB:   trc_launch_rcb_enc((const uint8_t *)d_in, n, chunk, w, d_clen, s); from_end = 0; break;
    case TRC_RCA:   trc_launch_rca_enc(1, 0, (const uint8_t *)d_in, n, chunk, w, d_clen, s); from_end = 0; break;
    case TRC_RCAI:  trc_launch_rca_enc(2, 0, (const uint8_t *)d_in, n, chunk, w, d_clen, s); from_end = 2; break;
    case TRC_RCA4:  trc_launch_rca_enc(1, 1, (const uint8_t *)d_in, n, chunk, w, d_clen, s); from_end = 0; break;
    case TRC_RCAI4: trc_launch_rca_enc(2, 1, (const uint8_t *)d_in, n, chunk, w, d_clen, s); from_end = 2; break;
    case TRC_ANSA:  trc_launch_ansa_enc(0, (const uint8_t *)d_in, n, chunk, w, d_clen, s); from_end = 1; break;
    case TRC_ANSA4: trc_launch_ansa_enc(1, (const uint8_t *)d_in, n, chunk, w, d_clen, s); from_end = 1; break;
    case TRC_ANSB:  trc_launch_ansb_enc((const uint8_t *)d_in, n, chunk, w, d_clen, s); from_end = 1; break;
    case TRC_RCV8:  trc_launch_rcv_enc(1, (const uint8_t *)d_in, n, chunk, w, d_clen, s); from_end = 0; break;
    case TRC_RCVI8: trc_launch_rcv_enc(2, (const uint8_t *)d_in, n, chunk, w, d_clen, s); from_end = 2; break;
    default:        if (is_vlc(codec)) { trc_launch_vlc_enc(vlc_variant(codec), vlc_elem(codec), (const uint8_t *)d_in, n, chunk, w, d_clen, s); from_end = 3; }
                    else if (is_vla(codec)) { trc_launch_vla_enc(vla_variant(codec), vla_zz(codec), vla_elem(codec), (const uint8_t *)d_in, n, chunk, w, d_clen, s); from_end = 4; }
                    break;
    case TRC_ANSO1: if (trc_launch_anso1_model((const uint8_t *)d_in, n, chunk, w, s)) trc_launch_ansa_code_planar(n, chunk, w, d_clen, s);
                    else trc_launch_ansa_code(0, n, chunk, w, d_clen, s);
                    from_end = 1; break;
    }
    tm_end(0);
    if (!gathered) {
        tm_begin(2);
        if (w.goff) trc_launch_scan_groups(w.gsum, w.ngroups, w.goff, d_total, s);
        trc_launch_gather((const uint8_t *)d_in, n, chunk, w, from_end, d_clen, (uint8_t *)d_payload, d_total, s);
        tm_end(2);
    }
    HIPCHK(hipGetLastError());
    return TRC_OK;
}

extern "C" int trc_decode_dev(int codec, const uint32_t *d_clen, const void *d_payload, size_t n, uint32_t chunk,
                              const uint16_t *d_cdf, unsigned cdfnum,
                              void *d_out, void *d_work, size_t work_bytes, void *stream)
{
    const bool tables_ready = codec & TRC_TABLES_READY, dir_ready = codec & TRC_DIR_READY;
    codec &= ~(TRC_TABLES_READY | TRC_DIR_READY);
    int rc = check_common(codec, n, chunk, d_cdf, cdfnum);
    if (rc) return rc;
    if (n == 0) return TRC_OK;
    if (((uintptr_t)d_out & 15) || ((uintptr_t)d_clen & 3) || ((uintptr_t)d_payload & 1))
        return fail(TRC_E_ARG, "decode: d_out must be 16-byte, d_clen 4-byte, d_payload 2-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    TrcWork w;
    if ((rc = carve(codec, n, chunk, d_work, work_bytes, w))) return rc;
    if (is_static(codec) && !tables_ready) trc_launch_static_prep(d_cdf, cdfnum, w.tables, s);
    // TRC_DIR_READY promises that the last encode OR DECODE on this workspace was of this directory (include/trc_hip.h), so both must
    // leave every group's base in goff_area: an encode's gather does (trc_launch.h); a decode derives the sums from clen[] and scans
    // them into the same place.  (Round 4 read goff_area after a decode that had never written it -- a workspace that only ever
    // decoded fed its second decode whatever the memory held: ADVICE r4.)
    if (!dir_ready) {
        trc_launch_group_sums(d_clen, w.nchunks, n, chunk, w.gsum, s);
        trc_launch_scan_groups(w.gsum, w.ngroups, w.goff_area, nullptr, s);
    }
    w.goff = w.goff_area;
    tm_begin(1);
    switch (codec) {
    case TRC_ANS4S: trc_launch_ans4s_dec((const uint8_t *)d_payload, d_clen, n, chunk, w, (uint8_t *)d_out, s); break;
    case TRC_RCS1:  trc_launch_rcs_dec(1, (const uint8_t *)d_payload, d_clen, n, chunk, w, (uint8_t *)d_out, s); break;
    case TRC_RCS2:  trc_launch_rcs_dec(2, (const uint8_t *)d_payload, d_clen, n, chunk, w, (uint8_t *)d_out, s); break;
    case TRC_RCSM:  trc_launch_rcs_dec(-1, (const uint8_t *)d_payload, d_clen, n, chunk, w, (uint8_t *)d_out, s); break;
    case TRC_RCB:   trc_launch_rcb_dec((const uint8_t *)d_payload, d_clen, n, chunk, w, (uint8_t *)d_out, s); break;
    case TRC_RCA:   trc_launch_rca_dec(1, 0, (const uint8_t *)d_payload, d_clen, n, chunk, w, (uint8_t *)d_out, s); break;
    case TRC_RCAI:  trc_launch_rca_dec(2, 0, (const uint8_t *)d_payload, d_clen, n, chunk, w, (uint8_t *)d_out, s); break;
    case TRC_RCA4:  trc_launch_rca_dec(1, 1, (const uint8_t *)d_payload, d_clen, n, chunk, w, (uint8_t *)d_out, s); break;
    case TRC_RCAI4: trc_launch_rca_dec(2, 1, (const uint8_t *)d_payload, d_clen, n, chunk, w, (uint8_t *)d_out, s); break;
    case TRC_ANSA:  trc_launch_ansa_dec(0, (const uint8_t *)d_payload, d_clen, n, chunk, w, (uint8_t *)d_out, s); break;
    case TRC_ANSA4: trc_launch_ansa_dec(1, (const uint8_t *)d_payload, d_clen, n, chunk, w, (uint8_t *)d_out, s); break;
    case TRC_ANSO1: trc_launch_anso1_dec((const uint8_t *)d_payload, d_clen, n, chunk, w, (uint8_t *)d_out, s); break;
    case TRC_ANSB:  trc_launch_ansb_dec((const uint8_t *)d_payload, d_clen, n, chunk, w, (uint8_t *)d_out, s); break;
    case TRC_RCV8:  trc_launch_rcv_dec(1, (const uint8_t *)d_payload, d_clen, n, chunk, w, (uint8_t *)d_out, s); break;
    case TRC_RCVI8: trc_launch_rcv_dec(2, (const uint8_t *)d_payload, d_clen, n, chunk, w, (uint8_t *)d_out, s); break;
    default:        if (is_vlc(codec)) trc_launch_vlc_dec(vlc_variant(codec), vlc_elem(codec), (const uint8_t *)d_payload, d_clen, n, chunk, w, (uint8_t *)d_out, s);
                    else if (is_vla(codec)) trc_launch_vla_dec(vla_variant(codec), vla_zz(codec), vla_elem(codec), (const uint8_t *)d_payload, d_clen, n, chunk, w, (uint8_t *)d_out, s);
                    break;
    }
    tm_end(1);
    HIPCHK(hipGetLastError());
    return TRC_OK;
}

// The kernel that takes the longest in the DEFAULT dispatch of the coder at the bench configurations (100 MB, the library's chunk):
// what a rocprofv3 --kernel-trace of bench.py lists first for that direction.  Two-pass encoders launch more than one kernel (the
// timing pairs sum them); forms behind tuning variables or other sizes (one-lane order-1 decoder, one-wave model passes) have
// other names.  Descriptive: nothing is dispatched by this string.
extern "C" const char *trc_kernel_name(int codec, int decode)
{
    switch (codec) {
    case TRC_ANS4S: return decode ? "trc_ans4s_dec_kernel" : "trc_ans4s_enc_kernel";
    case TRC_RCS1: case TRC_RCSM: return decode ? "trc_rcs_dec_kernel" : "trc_rcs_enc_kernel";
    case TRC_RCS2: return decode ? "trc_rcs2p_dec_kernel" : "trc_rcs2p_enc_kernel";
    case TRC_RCB: return decode ? "trc_rcb_dec_kernel" : "trc_rcb_enc_mc_kernel";
    case TRC_RCA: case TRC_RCAI: return decode ? "trc_rca_dec_kernel" : "trc_rca_enc_mc_kernel";
    case TRC_RCA4: case TRC_RCAI4: return decode ? "trc_rca_dec_kernel" : "trc_rca_enc_kernel";
    case TRC_ANSA: return decode ? "trc_ansa_dec_kernel" : "trc_ansa_model2_kernel";
    case TRC_ANSA4: return decode ? "trc_ansa_dec_kernel" : "trc_ansa_model_kernel";
    case TRC_ANSO1: return decode ? "trc_o1_dec_rowsn_kernel" : "trc_o1_sort_kernel";
    case TRC_ANSB: return decode ? "trc_ansb_dec_kernel" : "trc_ansb_model_kernel";
    case TRC_RCV8: case TRC_RCVI8: return decode ? "trc_rcv_dec_kernel" : "trc_rcv_enc_kernel";
    default: if (is_vlc(codec)) return decode ? "trc_vlc_dec_kernel" : "trc_vlc_enc_kernel";
             if (is_vla(codec)) return decode ? "trc_vla_dec_kernel" : "trc_vla_model_kernel";
    }
    return "";
}

// --------------------------------------------------- layer 1: reference prototypes (host pointers) ---
#include "trc_host.inc"

// Page-lock a caller's buffer for the host-pointer calls (hipHostRegister / hipHostUnregister behind plain C: a harness needs
// no HIP header).  Registration costs ~55 us per MB: once per buffer, not per call.
extern "C" int trc_host_pin(void *p, size_t len)
{
    if (!p || !len) return fail(TRC_E_ARG, "host_pin: bad arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(TRC_E_NODEV, "no HIP device");
    HIPCHK(hipHostRegister(p, len, hipHostRegisterDefault));
    return TRC_OK;
}
extern "C" int trc_host_unpin(void *p)
{
    if (!p) return fail(TRC_E_ARG, "host_unpin: bad arguments");
    HIPCHK(hipHostUnregister(p));
    return TRC_OK;
}

extern "C" size_t trc_container_bound(size_t n, uint32_t chunk)
{
    if (!chunk_ok(chunk)) return 0;
    return sizeof(trc_container_hdr) + 4 * ((n + chunk - 1) / chunk) + n;
}
extern "C" size_t trc_encode_host(int codec, const void *in, size_t n, uint32_t chunk, void *out, size_t outcap,
                                  const uint16_t *cdf, unsigned cdfnum)
{
    if (!codec_ok(codec)) { fail(TRC_E_ARG, "codec %d not available", codec); return 0; }
    if (!outcap) { fail(TRC_E_ARG, "encode_host: outcap must be given"); return 0; }
    return host_encode(codec, (const unsigned char *)in, n, (unsigned char *)out, (const cdf_t *)cdf, (int)cdfnum, chunk, outcap);
}

static int container_verdict(const void *buf, size_t buflen, int codec, size_t outlen, char *why, size_t whysz);
// The bounded decoder: like the reference-named decoder of `codec`, but the caller states how many bytes `in` really holds and the
// container is validated against THAT before anything is read (the reference prototypes carry no input length, so through them a
// forged header can make a decoder read past a short buffer).  Raw streams (inlen == outlen) are copied.  Returns outlen, 0 on error.
extern "C" size_t trc_decode_host(int codec, const void *in, size_t inlen, void *out, size_t outlen,
                                  const uint16_t *cdf, unsigned cdfnum)
{
    if (!codec_ok(codec)) { fail(TRC_E_ARG, "codec %d not available", codec); return 0; }
    if (!in || !out) { fail(TRC_E_ARG, "decode_host: bad arguments"); return 0; }
    if (outlen == 0) return 0;
    // inlen == outlen is the reference's "stored raw" convention -- but trc_encode_host with a stated capacity returns the
    // container whatever its size, so a container of exactly outlen bytes is possible (nearly incompressible input): what
    // validates as a container of this coder and this length is decoded, anything else of that size is the raw copy.
    char why[200];
    if (inlen == outlen && container_verdict(in, inlen, codec, outlen, why, sizeof why)) { memcpy(out, in, outlen); return outlen; }   // raw: not an error, nothing reported
    if (inlen != outlen && trc_container_check(in, inlen, codec, outlen)) return 0;
    return host_decode(codec, (const unsigned char *)in, outlen, (unsigned char *)out, (const cdf_t *)cdf, (int)cdfnum);
}

// ---- container validation for untrusted input (the reference prototypes carry no input length) ------
// Everything a decoder will read from buf is checked against buflen: header fields, the directory, and that the
// directory's (clamped) lengths add up to exactly the stated payload, which must end inside the buffer.
// quiet form: the verdict and, on failure, its reason in `why` -- nothing printed, trc_last_error() untouched (the raw-or-container
// probe of trc_decode_host is not an error)
static int container_verdict(const void *buf, size_t buflen, int codec, size_t outlen, char *why, size_t whysz)
{
#define BAD(...) do { snprintf(why, whysz, __VA_ARGS__); return TRC_E_ARG; } while (0)
    trc_container_hdr h;
    if (!buf || buflen < sizeof h) BAD("container: %zu bytes is shorter than the header", buflen);
    memcpy(&h, buf, sizeof h);
    if (h.magic != TRC_MAGIC || h.version != 1) BAD("container: bad magic/version");
    if (!codec_ok(h.codec) || (codec && h.codec != codec)) BAD("container: codec %u (expected %d)", h.codec, codec);
    if (!chunk_ok(h.chunk)) BAD("container: chunk %u", h.chunk);
    if (outlen != (size_t)-1 && h.n != outlen) BAD("container: holds %llu bytes, caller expects %zu", (unsigned long long)h.n, outlen);
    if (h.n == 0 || (h.n + h.chunk - 1) / h.chunk != h.nchunks) BAD("container: nchunks %u does not match n/chunk", h.nchunks);
    const size_t dir = 4 * (size_t)h.nchunks;
    if (dir > buflen - sizeof h) BAD("container: directory (%zu B) runs past the buffer", dir);
    if (h.payload > h.n || h.payload > buflen - sizeof h - dir) BAD("container: payload (%llu B) runs past the buffer", (unsigned long long)h.payload);
    const uint8_t *d = (const uint8_t *)buf + sizeof h;
    uint64_t sum = 0;
    for (uint32_t c = 0; c < h.nchunks; c++) {
        uint32_t l; memcpy(&l, d + 4 * (size_t)c, 4);
        const uint64_t len = (c + 1 == h.nchunks) ? h.n - (uint64_t)c * h.chunk : h.chunk;
        sum += l < len ? l : len;                               // the decoders read an entry above the chunk length as "raw"
    }
    if (sum != h.payload) BAD("container: directory sums to %llu, header says %llu", (unsigned long long)sum, (unsigned long long)h.payload);
    return TRC_OK;
#undef BAD
}
extern "C" int trc_container_check(const void *buf, size_t buflen, int codec, size_t outlen)
{
    char why[200];
    return container_verdict(buf, buflen, codec, outlen, why, sizeof why) ? fail(TRC_E_ARG, "%s", why) : TRC_OK;
}

// ---- exports with the reference's names (include/turborc.h:500, include/anscdf.h:40-96) ----------
extern "C" {

// cdfini: reference rccdf.c:50-68.  Returns (int)inlen; -1 where the reference would die().
int cdfini(unsigned char *in, size_t inlen, cdf_t *cdf, unsigned cdfnum)
{
    if (!inlen || cdfnum < 1 || cdfnum > 256) { fail(TRC_E_ARG, "cdfini: inlen=%zu cdfnum=%u", inlen, cdfnum); return -1; }
#define ICHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fail(TRC_E_HIP, "%s -> %s", #x, hipGetErrorString(e_)); return -1; } } while (0)
    const std::vector<int> devs = devs_get();
    if (devs.size() > 1) {
        // over the device list: every pipeline counts the bytes of its share (trc_hist_dev), the counts are summed on the host -- exact, so the
        // CDF is the one-device CDF bit for bit -- and the first pipeline turns the sum into the CDF (trc_cdf_from_hist_dev)
        const int nd = (int)devs.size();
        std::vector<uint64_t> hist((size_t)nd * 256, 0);
        std::vector<std::string> errs((size_t)nd);
        int prev = 0;
        (void)hipGetDevice(&prev);
        auto one = [&](int s) {
            HostCtx &c = g_mctxs[s];
            std::lock_guard<std::mutex> lk(c.mu);
            const size_t o = inlen / nd * s, l = s + 1 == nd ? inlen - o : inlen / nd;
            if (!l) return;
            uint64_t *d_hist = nullptr;
            if (hipSetDevice(devs[s]) != hipSuccess || ctx_init(c, devs[s]) || grow(&c.d_in, &c.cap_in, l)) { errs[s] = "device setup failed"; return; }
            d_hist = (uint64_t *)(c.d_small + 40960);
            if (hipMemcpyAsync(c.d_in, in + o, l, hipMemcpyHostToDevice, c.s_k[0]) != hipSuccess || trc_hist_dev(c.d_in, l, d_hist, c.s_k[0]) ||
                hipMemcpyAsync(&hist[(size_t)s * 256], d_hist, 2048, hipMemcpyDeviceToHost, c.s_k[0]) != hipSuccess ||
                hipStreamSynchronize(c.s_k[0]) != hipSuccess) errs[s] = "histogram failed";
        };
        std::vector<std::thread> th;
        for (int s = 1; s < nd; s++) th.emplace_back(one, s);
        one(0);
        for (auto &x : th) x.join();
        for (int s = 0; s < nd; s++) if (!errs[s].empty()) { (void)hipSetDevice(prev); fail(TRC_E_HIP, "cdfini over the device list: %s", errs[s].c_str()); return -1; }
        for (int s = 1; s < nd; s++) for (int k = 0; k < 256; k++) hist[k] += hist[(size_t)s * 256 + k];
        HostCtx &c = g_mctxs[0];
        std::lock_guard<std::mutex> lk(c.mu);
        int32_t st = -1;
        uint64_t *d_hist = (uint64_t *)(c.d_small + 40960);
        uint16_t *d_cdf = (uint16_t *)c.d_small;
        int32_t *d_status = (int32_t *)(c.d_small + 32768);
        const bool ok = hipSetDevice(devs[0]) == hipSuccess &&
                        hipMemcpyAsync(d_hist, hist.data(), 2048, hipMemcpyHostToDevice, c.s_k[0]) == hipSuccess &&
                        trc_cdf_from_hist_dev(d_hist, inlen, d_cdf, cdfnum, d_status, c.s_k[0]) == TRC_OK &&
                        hipMemcpyAsync(&st, d_status, 4, hipMemcpyDeviceToHost, c.s_k[0]) == hipSuccess &&
                        hipMemcpyAsync(cdf, d_cdf, (cdfnum + 1) * sizeof(cdf_t), hipMemcpyDeviceToHost, c.s_k[0]) == hipSuccess &&
                        hipStreamSynchronize(c.s_k[0]) == hipSuccess;
        (void)hipSetDevice(prev);
        if (!ok) { fail(TRC_E_HIP, "cdfini over the device list: building the CDF failed"); return -1; }
        if (st < 0) { fail(TRC_E_CDF, "cdfini: distribution cannot be normalised to a strictly increasing 15-bit CDF"); return -1; }
        return (int)inlen;
    }
    HostCtx *cp = nullptr;
    int dev = 0;
    if (devs.size() == 1) { cp = &g_mctxs[0]; dev = devs[0]; }
    else { if (ctx_get(cp)) return -1; if (hipGetDevice(&dev) != hipSuccess) return -1; }
    HostCtx &c = *cp;
    std::lock_guard<std::mutex> lk(c.mu);
    int prev = dev;
    (void)hipGetDevice(&prev);
    if (prev != dev) ICHK(hipSetDevice(dev));
    struct Back { int p, d; ~Back() { if (p != d) (void)hipSetDevice(p); } } back = { prev, dev };
    if (ctx_init(c, dev)) return -1;
    if (grow(&c.d_in, &c.cap_in, inlen) || grow(&c.d_work[0], &c.cap_work[0], 4096)) return -1;
    uint16_t *d_cdf = (uint16_t *)c.d_small;
    int32_t *d_status = (int32_t *)(c.d_small + 32768);
    ICHK(hipMemcpyAsync(c.d_in, in, inlen, hipMemcpyHostToDevice, c.s_k[0]));
    if (trc_cdfini_dev(c.d_in, inlen, d_cdf, cdfnum, d_status, c.d_work[0], c.s_k[0])) return -1;
    int32_t st = -1;
    ICHK(hipMemcpyAsync(&st, d_status, 4, hipMemcpyDeviceToHost, c.s_k[0]));
    ICHK(hipMemcpyAsync(cdf, d_cdf, (cdfnum + 1) * sizeof(cdf_t), hipMemcpyDeviceToHost, c.s_k[0]));
    ICHK(hipStreamSynchronize(c.s_k[0]));
    if (st < 0) { fail(TRC_E_CDF, "cdfini: distribution cannot be normalised to a strictly increasing 15-bit CDF"); return -1; }
    return (int)inlen;
}

void anscdfini(unsigned id) { (void)id; }   // reference: CPU ISA dispatch (anscdf.c:759-808); nothing to select here

#define TRC_EXPORT_ANS4S(sfx) \
    size_t anscdf4senc##sfx(unsigned char *in, size_t inlen, unsigned char *out, cdf_t *cdf) { return host_encode(TRC_ANS4S, in, inlen, out, cdf, 0); } \
    size_t anscdf4sdec##sfx(unsigned char *in, size_t outlen, unsigned char *out, cdf_t *cdf) { return host_decode(TRC_ANS4S, in, outlen, out, cdf, 0); }
TRC_EXPORT_ANS4S()
TRC_EXPORT_ANS4S(0)
TRC_EXPORT_ANS4S(s)
TRC_EXPORT_ANS4S(x)

// static-CDF range coder: rccdfsenc + the four equivalent decoders, rccdfs2enc + its two decoders
size_t rccdfsenc(unsigned char *in, size_t inlen, unsigned char *out, cdf_t *cdf, unsigned cdfnum) { return host_encode(TRC_RCS1, in, inlen, out, cdf, (int)cdfnum); }
#define TRC_EXPORT_RCS1DEC(name) size_t name(unsigned char *in, size_t outlen, unsigned char *out, cdf_t *cdf, unsigned cdfnum) { return host_decode(TRC_RCS1, in, outlen, out, cdf, (int)cdfnum); }
TRC_EXPORT_RCS1DEC(rccdfsldec)
TRC_EXPORT_RCS1DEC(rccdfsbdec)
TRC_EXPORT_RCS1DEC(rccdfsvldec)
TRC_EXPORT_RCS1DEC(rccdfsvbdec)
// one stream, 32-bit range / 16-bit I/O (reference rccdf.c:648-694; turborc -e44) -- SURVEY 8f rank 1
size_t rccdfsmenc(unsigned char *in, size_t inlen, unsigned char *out, cdf_t *cdf, unsigned cdfnum) { return host_encode(TRC_RCSM, in, inlen, out, cdf, (int)cdfnum); }
size_t rccdfsmbdec(unsigned char *in, size_t outlen, unsigned char *out, cdf_t *cdf, unsigned cdfnum) { return host_decode(TRC_RCSM, in, outlen, out, cdf, (int)cdfnum); }
size_t rccdfsmldec(unsigned char *in, size_t outlen, unsigned char *out, cdf_t *cdf, unsigned cdfnum) { return host_decode(TRC_RCSM, in, outlen, out, cdf, (int)cdfnum); }
size_t rccdfs2enc(unsigned char *in, size_t inlen, unsigned char *out, cdf_t *cdf, unsigned cdfnum) { return host_encode(TRC_RCS2, in, inlen, out, cdf, (int)cdfnum); }
size_t rccdfsl2dec(unsigned char *in, size_t outlen, unsigned char *out, cdf_t *cdf, unsigned cdfnum) { return host_decode(TRC_RCS2, in, outlen, out, cdf, (int)cdfnum); }
size_t rccdfsb2dec(unsigned char *in, size_t outlen, unsigned char *out, cdf_t *cdf, unsigned cdfnum) { return host_decode(TRC_RCS2, in, outlen, out, cdf, (int)cdfnum); }

// bitwise order-0 range coder, "s" predictor (reference rc_.c:37-58; turborc -e1 / file codec 1)
size_t rcsenc(unsigned char *in, size_t inlen, unsigned char *out) { return host_encode(TRC_RCB, in, inlen, out, nullptr, 0); }
size_t rcsdec(unsigned char *in, size_t outlen, unsigned char *out) { return host_decode(TRC_RCB, in, outlen, out, nullptr, 0); }

// adaptive-CDF byte range coder (reference rccdf.c:187-211; turborc -e46)
size_t rccdfenc(unsigned char *in, size_t inlen, unsigned char *out) { return host_encode(TRC_RCA, in, inlen, out, nullptr, 0); }
size_t rccdfdec(unsigned char *in, size_t outlen, unsigned char *out) { return host_decode(TRC_RCA, in, outlen, out, nullptr, 0); }

// interleaved adaptive-CDF byte range coder (reference rccdf.c:213-249; turborc -e47) -- SURVEY 8f rank 1
size_t rccdfienc(unsigned char *in, size_t inlen, unsigned char *out) { return host_encode(TRC_RCAI, in, inlen, out, nullptr, 0); }
size_t rccdfidec(unsigned char *in, size_t outlen, unsigned char *out) { return host_decode(TRC_RCAI, in, outlen, out, nullptr, 0); }

// "vnibble" adaptive-CDF range coders (reference rccdf.c:326-390; turborc -e48 / -e49)
size_t rccdfenc8(unsigned char *in, size_t inlen, unsigned char *out) { return host_encode(TRC_RCV8, in, inlen, out, nullptr, 0); }
size_t rccdfdec8(unsigned char *in, size_t outlen, unsigned char *out) { return host_decode(TRC_RCV8, in, outlen, out, nullptr, 0); }
size_t rccdfienc8(unsigned char *in, size_t inlen, unsigned char *out) { return host_encode(TRC_RCVI8, in, inlen, out, nullptr, 0); }
size_t rccdfidec8(unsigned char *in, size_t outlen, unsigned char *out) { return host_decode(TRC_RCVI8, in, outlen, out, nullptr, 0); }

// adaptive-CDF byte rANS (reference anscdf.c:567-605, dispatch :816-817; turborc -e56 / -e57 / -e58)
#define TRC_EXPORT_ANSA(sfx) \
    size_t anscdfenc##sfx(unsigned char *in, size_t inlen, unsigned char *out) { return host_encode(TRC_ANSA, in, inlen, out, nullptr, 0); } \
    size_t anscdfdec##sfx(unsigned char *in, size_t outlen, unsigned char *out) { return host_decode(TRC_ANSA, in, outlen, out, nullptr, 0); }
TRC_EXPORT_ANSA()
TRC_EXPORT_ANSA(0)
TRC_EXPORT_ANSA(s)
TRC_EXPORT_ANSA(x)
// the `turborc -n` coders on values 0..15 -- SURVEY 8f rank 1
// adaptive-CDF nibble range coder, one stream (reference rccdf.c:250-275) and interleaved (rccdf.c:277-323)
size_t rccdf4enc(unsigned char *in, size_t inlen, unsigned char *out) { return host_encode(TRC_RCA4, in, inlen, out, nullptr, 0); }
size_t rccdf4dec(unsigned char *in, size_t outlen, unsigned char *out) { return host_decode(TRC_RCA4, in, outlen, out, nullptr, 0); }
size_t rccdf4ienc(unsigned char *in, size_t inlen, unsigned char *out) { return host_encode(TRC_RCAI4, in, inlen, out, nullptr, 0); }
size_t rccdf4idec(unsigned char *in, size_t outlen, unsigned char *out) { return host_decode(TRC_RCAI4, in, outlen, out, nullptr, 0); }
// adaptive-CDF nibble rANS (reference anscdf.c:87-133, dispatch :814-815)
#define TRC_EXPORT_ANSA4(sfx) \
    size_t anscdf4enc##sfx(unsigned char *in, size_t inlen, unsigned char *out) { return host_encode(TRC_ANSA4, in, inlen, out, nullptr, 0); } \
    size_t anscdf4dec##sfx(unsigned char *in, size_t outlen, unsigned char *out) { return host_decode(TRC_ANSA4, in, outlen, out, nullptr, 0); }
TRC_EXPORT_ANSA4()
TRC_EXPORT_ANSA4(0)
TRC_EXPORT_ANSA4(s)
TRC_EXPORT_ANSA4(x)

// order-1 adaptive-CDF byte rANS (reference anscdf.c:607-645, dispatch :818-819; turborc -e64) -- SURVEY 8f rank 2
#define TRC_EXPORT_ANSO1(sfx) \
    size_t anscdf1enc##sfx(unsigned char *in, size_t inlen, unsigned char *out) { return host_encode(TRC_ANSO1, in, inlen, out, nullptr, 0); } \
    size_t anscdf1dec##sfx(unsigned char *in, size_t outlen, unsigned char *out) { return host_decode(TRC_ANSO1, in, outlen, out, nullptr, 0); }
TRC_EXPORT_ANSO1()
TRC_EXPORT_ANSO1(0)
TRC_EXPORT_ANSO1(s)
TRC_EXPORT_ANSO1(x)

// bitwise order-0 rANS (reference anscdf.c:672-731; turborc -e66) -- SURVEY 8f rank 2.  Chunks above one reference
// block (8192 bytes) are not supported by the kernels: the call uses min(configured chunk, 8192).
size_t ansbc(unsigned char *in, size_t inlen, unsigned char *out) { return host_encode(TRC_ANSB, in, inlen, out, nullptr, 0); }
size_t ansbd(unsigned char *in, size_t outlen, unsigned char *out) { return host_decode(TRC_ANSB, in, outlen, out, nullptr, 0); }

// Turbo-VLC integer coders over the adaptive CDF range coder (reference rccdf.c:391-632; turborc -e50/52/53 with 16- or
// 32-bit input) -- SURVEY 8f rank 3.  Lengths are byte counts, as in the reference.
#define TRC_EXPORT_VLC(name, codec) \
    size_t name##enc16(unsigned char *in, size_t inlen, unsigned char *out) { return host_encode(codec##16, in, inlen, out, nullptr, 0); } \
    size_t name##dec16(unsigned char *in, size_t outlen, unsigned char *out) { return host_decode(codec##16, in, outlen, out, nullptr, 0); } \
    size_t name##enc32(unsigned char *in, size_t inlen, unsigned char *out) { return host_encode(codec##32, in, inlen, out, nullptr, 0); } \
    size_t name##dec32(unsigned char *in, size_t outlen, unsigned char *out) { return host_decode(codec##32, in, outlen, out, nullptr, 0); }
TRC_EXPORT_VLC(rccdfu, TRC_VLCU)
TRC_EXPORT_VLC(rccdfv, TRC_VLCV)
TRC_EXPORT_VLC(rccdfvz, TRC_VLCVZ)
// ... and over the adaptive CDF rANS (reference anscdf.c:139-483, dispatch :820-833; turborc -e60..63)
#define TRC_EXPORT_VLA(name, bits, codec) \
    size_t name##enc##bits(unsigned char *in, size_t inlen, unsigned char *out) { return host_encode(codec, in, inlen, out, nullptr, 0); } \
    size_t name##dec##bits(unsigned char *in, size_t outlen, unsigned char *out) { return host_decode(codec, in, outlen, out, nullptr, 0); } \
    size_t name##enc##bits##0(unsigned char *in, size_t inlen, unsigned char *out) { return host_encode(codec, in, inlen, out, nullptr, 0); } \
    size_t name##dec##bits##0(unsigned char *in, size_t outlen, unsigned char *out) { return host_decode(codec, in, outlen, out, nullptr, 0); } \
    size_t name##enc##bits##s(unsigned char *in, size_t inlen, unsigned char *out) { return host_encode(codec, in, inlen, out, nullptr, 0); } \
    size_t name##dec##bits##s(unsigned char *in, size_t outlen, unsigned char *out) { return host_decode(codec, in, outlen, out, nullptr, 0); } \
    size_t name##enc##bits##x(unsigned char *in, size_t inlen, unsigned char *out) { return host_encode(codec, in, inlen, out, nullptr, 0); } \
    size_t name##dec##bits##x(unsigned char *in, size_t outlen, unsigned char *out) { return host_decode(codec, in, outlen, out, nullptr, 0); }
TRC_EXPORT_VLA(anscdfu, 16, TRC_VLAU16)
TRC_EXPORT_VLA(anscdfuz, 16, TRC_VLAUZ16)
TRC_EXPORT_VLA(anscdfv, 16, TRC_VLAV16)
TRC_EXPORT_VLA(anscdfvz, 16, TRC_VLAVZ16)
TRC_EXPORT_VLA(anscdfv, 32, TRC_VLAV32)
TRC_EXPORT_VLA(anscdfvz, 32, TRC_VLAVZ32)

typedef size_t (*fanscdfenc)(unsigned char *in, size_t inlen, unsigned char *out);
typedef size_t (*fanscdfdec)(unsigned char *in, size_t inlen, unsigned char *out);
fanscdfenc _anscdfenc = anscdfenc;      // the reference's dispatch globals (include/anscdf.h:32-35)
fanscdfdec _anscdfdec = anscdfdec;
fanscdfenc _anscdf4enc = anscdf4enc;
fanscdfdec _anscdf4dec = anscdf4dec;

}  // extern "C"
