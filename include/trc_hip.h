/*
 * trc_hip.h -- C-ABI of libturborc_hip.so, the MI355X (gfx950) entropy-coding core.
 *
 * Two layers, both plain C (no torch / C++ types in any signature):
 *
 *  (1) the reference's own prototypes -- include/anscdf.h and include/turborc.h in this repo carry
 *      the same signatures as the reference's include/anscdf.h:40-52,70-96 and
 *      include/turborc.h:62-63,497-519 -- host pointers in, host pointers out, so a TurboRC-style
 *      bench harness links unchanged;
 *  (2) the device-resident entry points below (the "*_dev" extension SURVEY.md section 8b allows):
 *      everything stays in HBM, the caller owns all buffers and the HIP stream.  bench.py, the
 *      parity tests and the multi-GPU path use this layer.
 *
 * Unit of parallelism = CHUNK.  The input is cut into `chunk`-byte slices; chunk c is coded by the
 * reference algorithm exactly as if the reference function had been called on that slice alone:
 *
 *        payload(c) == reference_fn(in + c*chunk, len_c)          (bit-exact, incl. raw fallback)
 *
 * clen[c] is the reference function's return value for the slice (== len_c means "stored raw",
 * include/turborc.h:50-53).  Payloads are concatenated without padding in chunk order.
 *
 * Host-pointer calls wrap this in a self-describing container:
 *        trc_container_hdr (32 B) | uint32 clen[nchunks] | payload bytes
 * and keep the reference's return convention (== inlen  =>  out is a raw copy of in).
 *
 * How a host-pointer call runs (csrc/trc_host.inc): a PCIe pipeline per device -- the chunk chosen for the
 * stored size (trc_auto_chunk_codec), slices coded concurrently on two coder streams, pageable memory staged in 8 MB pieces by copy
 * threads, page-locked caller buffers (trc_host_pin) read and written by DMA directly.  For rccdf / rccdfi / anscdf / rcs the input
 * is delivered in striped passes (the k-th part of every chunk per 2-D copy) to an encoder that is already waiting at an arrival
 * gate, and a decoder's output is fetched part by part while it is still running; trc_set_devices / TRC_DEVICES spread a call over
 * several GPUs.  None of this changes a byte of the container.
 */
#ifndef TRC_HIP_H_
#define TRC_HIP_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* coder ids (container `codec` byte).  Reference function each one reproduces per chunk: */
enum trc_codec {
    TRC_ANS4S = 1,  /* anscdf4senc / anscdf4sdec   static-CDF rANS, 2 states      anscdf.c:57-85   (-e65) */
    TRC_RCS1  = 2,  /* rccdfsenc   / rccdfs*dec    static-CDF RC, 1 stream        rccdf.c:71-122   (-e42/43) */
    TRC_RCS2  = 3,  /* rccdfs2enc  / rccdfs*2dec   static-CDF RC, 2 streams       rccdf.c:125-184  (-e45) */
    TRC_RCA   = 4,  /* rccdfenc    / rccdfdec      adaptive-CDF byte RC           rccdf.c:187-211  (-e46) */
    TRC_ANSA  = 5,  /* anscdfenc   / anscdfdec     adaptive-CDF byte rANS, 4 st.  anscdf.c:567-605 (-e56) */
    TRC_RCB   = 6,  /* rcsenc      / rcsdec        bitwise order-0 RC             rc_.c:37-58      (-e1)  */
    TRC_RCAI  = 7,  /* rccdfienc   / rccdfidec     adaptive-CDF byte RC, 2 streams rccdf.c:213-249 (-e47) */
    /* the `turborc -n` coders: input values 0..15, one CDF16 table (harness gate m<16, turborc.c:499-520) */
    TRC_RCA4  = 8,  /* rccdf4enc   / rccdf4dec     adaptive-CDF nibble RC         rccdf.c:250-275  (-n -e46) */
    TRC_RCAI4 = 9,  /* rccdf4ienc  / rccdf4idec    ... on 2 interleaved streams   rccdf.c:277-323  (-n -e47) */
    TRC_ANSA4 = 10, /* anscdf4enc  / anscdf4dec    adaptive-CDF nibble rANS, 2 st. anscdf.c:87-133 (-n -e56) */
    TRC_RCSM  = 11, /* rccdfsmenc  / rccdfsm*dec   static-CDF RC, 32-bit range, 16-bit I/O rccdf.c:648-694 (-e44) */
    TRC_ANSO1 = 12, /* anscdf1enc  / anscdf1dec    order-1 adaptive-CDF byte rANS  anscdf.c:607-645 (-e64); 136 KiB of
                       model per chunk in the workspace: use chunks of 4 KiB and more */
    TRC_ANSB  = 13, /* ansbc       / ansbd         bitwise order-0 rANS, 4 states  anscdf.c:672-731 (-e66); chunk <= 8192
                       (one reference block) */
    /* Turbo-VLC integer coders over the adaptive CDF range coder, 16- / 32-bit elements (rccdf.c:391-632; -e50/52/53) */
    TRC_VLCU16 = 14,  TRC_VLCU32 = 15,   /* rccdfuenc16/32, rccdfudec16/32     6-bit exponent */
    TRC_VLCV16 = 16,  TRC_VLCV32 = 17,   /* rccdfvenc16/32, rccdfvdec16/32     7-bit exponent */
    TRC_VLCVZ16 = 18, TRC_VLCVZ32 = 19,  /* rccdfvzenc16/32, rccdfvzdec16/32   7-bit exponent on zigzag deltas */
    /* ... and over the adaptive CDF rANS (anscdf.c:139-483; -e60..63) */
    TRC_VLAU16 = 20,  TRC_VLAUZ16 = 21,  /* anscdfuenc16 / anscdfuzenc16 (+dec)        6-bit exponent, plain / zigzag deltas */
    TRC_VLAV16 = 22,  TRC_VLAVZ16 = 23,  /* anscdfvenc16 / anscdfvzenc16 (+dec)        7-bit exponent */
    TRC_VLAV32 = 24,  TRC_VLAVZ32 = 25,  /* anscdfvenc32 / anscdfvzenc32 (+dec) */
    /* "vnibble" coders: a byte becomes 1-3 CDF16 symbols on three adaptive tables (rccdf.c:326-390, rccdf_.h:76-98) */
    TRC_RCV8 = 26,    /* rccdfenc8  / rccdfdec8    one stream    (-e48) */
    TRC_RCVI8 = 27    /* rccdfienc8 / rccdfidec8   two streams   (-e49) */
};

#define TRC_MAGIC        0x31435254u   /* "TRC1" */
#define TRC_CHUNK_MIN    256u
#define TRC_CHUNK_MAX    65536u        /* chunk must be a multiple of 64 in [MIN, MAX] */
#define TRC_CHUNK_AUTO_MIN 512u        /* the parallel unit is the chunk: at 100 MB per GPU 4096 leaves 1.5 waves per CU (static rANS
                                          163 GB/s), 1024 six (449 GB/s), 512 twelve (590 GB/s); payload ratio on text 63.50 / 63.94 /
                                          64.52 %.  Gigabyte inputs fill the chip at 4096 too, so host-pointer calls pick the size from
                                          the input length (trc_auto_chunk) unless the caller fixes it. */
#define TRC_ANSB_CHUNK_MAX 8192u       /* TRC_ANSB only: one 8192-byte block of the reference per chunk */
#define TRC_PAD          256u          /* readable slack the device entry points need after every buffer */

typedef struct trc_container_hdr {
    uint32_t magic;      /* TRC_MAGIC */
    uint8_t  codec;      /* enum trc_codec */
    uint8_t  version;    /* 1 */
    uint16_t cdfnum;     /* static coders: alphabet size the CDF was built for, else 0 */
    uint32_t chunk;      /* chunk size in bytes */
    uint32_t nchunks;    /* ceil(n / chunk) */
    uint64_t n;          /* original length */
    uint64_t payload;    /* total payload bytes (sum of clen[]) */
} trc_container_hdr;     /* 32 bytes, little endian */

/* error codes of the *_dev layer (0 = ok) */
enum { TRC_OK = 0, TRC_E_ARG = -1, TRC_E_HIP = -2, TRC_E_WORK = -3, TRC_E_CDF = -4, TRC_E_NODEV = -5 };

/* last error text of the calling thread's most recent failing call ("" if none) */
const char *trc_last_error(void);

/* number of visible HIP devices (0 if the runtime cannot initialise -- no CPU fallback exists) */
int trc_device_count(void);

/* chunk size of the host-pointer (reference-signature) calls, process-wide.  0 = automatic (the default): every call
 * takes trc_auto_chunk_codec(its coder, its input length) -- round 6: the LARGEST chunk of the ladder 512 .. 16 384 whose
 * one-wave time still hides behind the call's PCIe time (budget max(1.7 ms, 0.35 n / 50 GB/s)); what the caller of these
 * functions sees is the stored size and a PCIe-bound rate, and every chunk costs coder state, a directory entry and, for
 * the adaptive coders, a model that starts from scratch.  100 MB: 4096 for the static and the adaptive byte coders, 2048
 * for the bitwise ones; 1 GB: 16 384 (rccdfenc on drift: 26.9 % stored against 26.7 % for one whole-buffer call of the
 * reference; at chunk 512 it was 38.7 %).  Static coders stop at 4096, the bitwise rANS at one reference block (8192), the
 * order-1 rANS never goes below 4096.  trc_set_chunk(c) or TRC_CHUNK=c in the environment fix it; trc_set_chunk(0)
 * returns to automatic.  Decoders take the size from the container. */
int      trc_set_chunk(uint32_t chunk);
uint32_t trc_get_chunk(void);
uint32_t trc_auto_chunk(size_t n);                   /* = trc_auto_chunk_codec(TRC_ANS4S, n): the static coders' rule */
uint32_t trc_auto_chunk_codec(int codec, size_t n);
/* The devices of the host-pointer calls.  Default (ndev = 0, TRC_DEVICES unset): the caller's current device.  With a list --
 * trc_set_devices, or TRC_DEVICES="all" / "0,1,2,3" in the environment -- every call is cut into contiguous shards of whole
 * chunk groups, one per list entry, coded at the same time (one pipeline and one host thread per entry) and written straight
 * to their places in the caller's buffer: the result is byte-identical to the one-device container, and it is how a
 * single-threaded caller such as the reference harness (turborc.c:420-579) uses all GPUs of a node.  An entry may repeat (two
 * pipelines on one device: what the tests do on a one-GPU box).  trc_get_devices returns the list length. */
int trc_set_devices(const int *devs, int ndev);
int trc_get_devices(int *devs, int cap);
/* Diagnostic (needs no device): how a host-pointer call of n bytes would run on one pipeline.  chunk 0 = the automatic one; decode 0 / 1;
 * page_locked: the caller's input (encode) or output (decode) buffer is page-locked.  first_chunk[0 .. slices] <- the first chunk of every
 * slice (launch), the last entry = the number of chunks (at most `cap` entries are written); *part_bytes <- bytes of a chunk per pass when
 * the call is striped (encode) / streamed (decode), else 0.  Returns the number of slices, or a negative error code. */
int trc_host_plan(int codec, size_t n, uint32_t chunk, int decode, int page_locked, size_t *first_chunk, int cap, uint32_t *part_bytes);
/* the chunk for a DEVICE-RESIDENT call of n bytes (one launch over the whole input): the largest multiple of 64 <= 4096 that
 * makes the input a whole number of residency rounds of the coder's lanes, barely (a launch lasts rounds x one wave's time:
 * 100 MB of the model-per-lane coders at 1280 instead of 1536 is half the throughput).  What bench.py runs every coder at. */
uint32_t trc_round_chunk(int codec, size_t n);

/* ---- device-resident layer --------------------------------------------------------------------
 * All d_* pointers are device pointers on the current HIP device, 16-byte aligned, with TRC_PAD
 * readable/writable bytes of slack behind the stated size.  `stream` is a hipStream_t (NULL =
 * default stream).  Calls only enqueue work; they never synchronise.                            */

/* bytes of device workspace trc_encode_dev / trc_decode_dev need for (codec, n, chunk) */
size_t trc_work_bytes(int codec, size_t n, uint32_t chunk);

/* cdfini on device (reference: rccdf.c:50-68): byte histogram of d_in[0..n) -> 15-bit CDF
 * d_cdf[0..cdfnum] (uint16).  d_status (int32, device) receives (int)n or -1 where the reference
 * would die().  d_work: >= trc_work_bytes(0, 0, 0) bytes.                                        */
int trc_cdfini_dev(const void *d_in, size_t n, uint16_t *d_cdf, unsigned cdfnum,
                   int32_t *d_status, void *d_work, void *stream);

/* The two halves of trc_cdfini_dev, for sharded inputs: every rank histograms its shard
 * (d_hist: uint64[256], zeroed by the call), the histograms are summed across ranks (one RCCL
 * all-reduce of 2 KiB), then every rank builds the same CDF from the global histogram. */
int trc_hist_dev(const void *d_in, size_t n, uint64_t *d_hist, void *stream);
int trc_cdf_from_hist_dev(const uint64_t *d_hist, size_t n_total, uint16_t *d_cdf, unsigned cdfnum,
                          int32_t *d_status, void *stream);

/* Static coders derive their symbol tables (44 KiB at the start of the workspace) from the CDF at every
 * trc_encode_dev / trc_decode_dev call.  A caller that codes many buffers against one CDF -- the reference harness
 * builds its CDF once, untimed, before the timed calls (turborc.c:429-433) -- can build them once with
 * trc_tables_dev and pass `codec | TRC_TABLES_READY` afterwards; the tables stay valid until the CDF or the
 * workspace changes. */
#define TRC_TABLES_READY 0x100
int trc_tables_dev(const uint16_t *d_cdf, unsigned cdfnum, void *d_work, size_t work_bytes, void *stream);

/* Encode n bytes at d_in with `codec`.
 *   d_cdf/cdfnum : static coders only (uint16[cdfnum+1], cdf[cdfnum] == 32768), else NULL/0
 *   d_clen       : uint32[nchunks]  <- per-chunk compressed length (== chunk length: raw)
 *   d_payload    : >= n bytes       <- concatenated payloads
 *   d_total      : uint64           <- sum of clen[]
 *   d_work       : trc_work_bytes() bytes of scratch                                             */
int trc_encode_dev(int codec, const void *d_in, size_t n, uint32_t chunk,
                   const uint16_t *d_cdf, unsigned cdfnum,
                   uint32_t *d_clen, void *d_payload, uint64_t *d_total,
                   void *d_work, size_t work_bytes, void *stream);

/* Decode: inverse of trc_encode_dev; d_out receives n bytes.
 * The decoders find a chunk's payload through per-group sums of d_clen, which a small kernel derives at every call.
 * trc_encode_dev leaves the very same sums in the workspace as a by-product, and so does every trc_decode_dev: a
 * caller that decodes the directory the workspace last saw -- encode followed by decode on one workspace, or the
 * same container decoded repeatedly, as the reference harness does -- may pass `codec | TRC_DIR_READY` to skip that
 * kernel (same (codec, n, chunk), d_clen contents unchanged since; not checked). */
#define TRC_DIR_READY 0x200
int trc_decode_dev(int codec, const uint32_t *d_clen, const void *d_payload, size_t n, uint32_t chunk,
                   const uint16_t *d_cdf, unsigned cdfnum,
                   void *d_out, void *d_work, size_t work_bytes, void *stream);

/* ---- multi-GPU: the gather of results over RCCL (xGMI), plain C ------------------------------------------------
 * One process per GPU; every rank codes a contiguous range of whole chunks with the calls above (no data-path
 * collective).  Static coders first agree on one CDF: trc_hist_dev on the shard, trc_hist_allreduce_dev (256 x u64
 * summed in place), trc_cdf_from_hist_dev with the total length -- the gathered container then equals the single-GPU
 * container of the whole input bit for bit.
 * trc_exchange_dev gathers `nbatch` consecutive results at once, batch j onto rank j % world (nbatch = 1: the plain
 * gather onto rank 0; nbatch = world: every directed xGMI link carries one payload, all at the same time).  One
 * all-gather of the sizes, a host sync to read them, then ONE grouped ncclSend/ncclRecv call.
 *   nccl_comm   an ncclComm_t (passed as void*: RCCL is resolved at run time, this header needs no RCCL header)
 *   b[j]        batch j: this rank's result (d_clen[nchunks], d_payload, d_total as trc_encode_dev left them) and, on the
 *               batch's root, the receive buffers: d_clen_all (all ranks' directory slices in rank order) and
 *               d_payload_all (all ranks' payloads in rank order = the container's payload area)
 *   h_sizes     host, uint64[world * nbatch * 2] <- {payload bytes, chunks} of rank r, batch j at [(r*nbatch + j)*2]
 *   d_meta      device scratch, 16 * nbatch * (world + 1) bytes                                                    */
#define TRC_EXCHANGE_MAX_BATCH 64
typedef struct trc_batch {
    const uint32_t *d_clen; size_t nchunks; const void *d_payload; const uint64_t *d_total;
    uint32_t *d_clen_all; void *d_payload_all;
} trc_batch;
int trc_exchange_dev(void *nccl_comm, int nbatch, const trc_batch *b, uint64_t *h_sizes, void *d_meta, void *stream);
int trc_hist_allreduce_dev(void *nccl_comm, uint64_t *d_hist, void *stream);

/* Host-pointer encode that ALWAYS returns the TRC1 container, with an explicit chunk size -- for callers that repackage the
 * per-chunk payloads themselves (harness/trcfile.c writes the reference's file format from it) and therefore must not get
 * the reference convention "return == n means out is a raw copy" applied to the container as a whole.  out must hold
 * trc_container_bound(n, chunk) bytes (header + directory + n: every chunk stored raw).  Returns the container size, 0 on
 * error.  Decode with the reference-named decoder of the codec, or trc_decode_dev. */
size_t trc_container_bound(size_t n, uint32_t chunk);
size_t trc_encode_host(int codec, const void *in, size_t n, uint32_t chunk, void *out, size_t outcap,
                       const uint16_t *cdf, unsigned cdfnum);

/* Bounded decode for untrusted input: the reference-named decoders carry no input length, this one does -- the container is
 * validated against `inlen` (trc_container_check) before anything is read; inlen == outlen means "stored raw" and is copied.
 * cdf / cdfnum: static coders only (cdfnum 0: derived from the CDF's terminating 32768).  Returns outlen, 0 on error. */
size_t trc_decode_host(int codec, const void *in, size_t inlen, void *out, size_t outlen,
                       const uint16_t *cdf, unsigned cdfnum);

/* Host-pointer calls and page-locked memory.  A pageable caller buffer travels through pinned staging slots (copy threads
 * + DMA: 38-40 GB/s per direction on the MI355X box); a page-locked one -- hipHostMalloc, or registered with the pair
 * below -- is read / written by DMA directly, detected per call with hipPointerGetAttributes.  Registration costs ~55 us
 * per MB, so it pays for buffers that are reused, as the reference harness reuses (in, out) for every timed repetition. */
int trc_host_pin(void *p, size_t len);
int trc_host_unpin(void *p);

/* Validate a TRC1 container held in buf[0..buflen) BEFORE handing it to a reference-named decoder: those prototypes
 * carry no input length, so a caller reading untrusted files must check that everything the decoder will touch lies
 * inside its buffer.  Checks header fields, codec (0 = any), the original length (outlen, (size_t)-1 = any), that the
 * directory fits, and that the directory's lengths add up to exactly the stated payload, which must end inside
 * buflen.  Host-only (no GPU needed).  Returns TRC_OK or TRC_E_ARG (text in trc_last_error()). */
int trc_container_check(const void *buf, size_t buflen, int codec, size_t outlen);

/* Optional timing of the coder kernels: every coder launch of a call carries a HIP event pair (hipExtLaunchKernel
 * start/stop events on the caller's stream), so the durations are the kernels' own -- BOTH passes of the two-pass
 * rANS encoders and the order-1 model fill included (the directory/gather kernels are not coder kernels).
 * enable(1) resets the counters; read() waits for the recorded events and returns the summed duration and the
 * number of encode (decode) CALLS measured: total_ms / launches = coder-kernel time of one call (at most 4096
 * kernel launches per direction between two enable() calls).  `decode` = 2 reads the third class: the encode path's own
 * directory work (group scan on large inputs, payload gather), so that (N + C) / (coder + gather) is a measured quantity.
 * Thread-safe. */
int trc_timing_enable(int on);
int trc_timing_pause(int paused);   /* suspend (1) / resume (0) the event pairs without resetting what was collected: time a SAMPLE of the calls */
int trc_timing_read(int decode, double *total_ms, int *launches);

/* name of the dominant kernel the last encode/decode of `codec` launched (for rocprof lookups) */
const char *trc_kernel_name(int codec, int decode);

#ifdef __cplusplus
}
#endif
#endif
