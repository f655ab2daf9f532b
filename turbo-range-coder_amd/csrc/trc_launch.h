// trc_launch.h -- host-side launch entry points of the kernel translation units (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

// Optional timing of the coder kernels of a call: while the API layer has timing armed for the calling thread, every
// coder launch of the call (both passes of the two-pass rANS encoders, the order-1 model fill) carries its own event
// pair (hipExtLaunchKernelGGL: timestamps of the dispatch itself, no extra barrier packets in the queue -- separate
// hipEventRecord calls around a launch cost ~6 us of queue time each); otherwise it is a plain launch.  `kern` goes in
// parentheses when it is a template instance.
bool trc_tm_next(hipEvent_t *start, hipEvent_t *stop);      // trc_api.hip: hands out the next pair, false = not timing
#define TRC_LAUNCH_TIMED(kern, grid, block, lds, stream, ...)                                                      \
    do {                                                                                                         \
        hipEvent_t tm_a_, tm_b_;                                                                                 \
        if (trc_tm_next(&tm_a_, &tm_b_)) hipExtLaunchKernelGGL(kern, grid, block, (uint32_t)(lds), stream, tm_a_, tm_b_, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);                                    \
    } while (0)

// Kernels that need more than 64 KiB of dynamic LDS raise their limit once per DEVICE (the attribute is per device:
// a process that moves to another GPU must set it there too).
bool trc_first_use_on_device(unsigned long long *mask);
#define TRC_RAISE_LDS_ONCE(kern, bytes)                                                                            \
    do {                                                                                                         \
        static unsigned long long seen_ = 0;                                                                     \
        if (trc_first_use_on_device(&seen_))                                                                     \
            (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
    } while (0)

// a dynamic-LDS request above half of a CU's 160 KiB: at most one such workgroup per CU (the large workgroups whose waves keep each
// other's pace, TrcPace: their waves are meant to be the only ones on their SIMDs)
#define TRC_LDS_ONE_PER_CU (82u * 1024u)
// the nibble coders (48-byte model rows: many waves per CU) take the large workgroup shape when the launch is one residency round of
// twelve waves per CU (TRC_NIB_WPG, trc_dev.h); TRC_NIB_BIG=0 / 1 forces the shape (tuning aid, tests)
static inline bool trc_nib_big(uint32_t ngroups)
{
    static const int env = getenv("TRC_NIB_BIG") ? atoi(getenv("TRC_NIB_BIG")) : -1;
    return env >= 0 ? env != 0 : (ngroups >= 2048u && ngroups <= 12u * 256u);
}

// The arrival gate of the NEXT encode launched by this thread (trc_io.h, WaveChunks::gate): set by the host layer around its
// trc_encode_dev call, read by the launchers of the encoders that honour it (trc_gate_ok), null otherwise.
struct TrcGate { const uint32_t *flag; uint32_t part; };
extern thread_local TrcGate trc_gate_tls;
struct TrcProg { uint32_t *counters; uint32_t *host_flags; uint32_t part; };  // ... and the progress counters of the next DECODE (WaveChunks::prog)
extern thread_local TrcProg trc_prog_tls;
bool trc_prog_ok(int codec);                                   // the default decoder form of `codec` reports its progress
bool trc_rca_dec_prog_ok();
bool trc_ansa_dec_prog_ok();
bool trc_rcb_dec_prog_ok();
bool trc_gate_ok(int codec);                                   // the default encoder form of `codec` waits at the gate (trc_api.hip)
bool trc_rca_enc_gate_ok();
bool trc_ansa_enc_gate_ok();
bool trc_rcb_enc_gate_ok();

// Workspace carve-up shared by encode and decode (all offsets 256-byte aligned).
struct TrcWork {
    uint8_t  *tables;    // per-call coder tables derived from the CDF (static coders)
    uint32_t *gsum;      // per-group (64 chunks) payload bytes
    uint64_t *goff;      // exclusive prefix of gsum (ngroups+1 entries) when the scan kernel runs; NULL = kernels sum gsum themselves
    uint64_t *goff_area; // where that prefix lives in the workspace, always.  Round 4: the gather of an encode leaves every group's base
                         // there (it has just computed it), and a decode of that very directory (TRC_DIR_READY) reads it instead of
                         // having each of its waves add up the group sums below its own
    uint8_t  *scratch;   // encode only: per-chunk private output regions
    uint32_t  stride;    // bytes per scratch region
    uint8_t  *scratch2;  // second region array (RCS2: stream 1)
    uint32_t  stride2;
    uint32_t  nchunks, ngroups;
    uint8_t  *model;     // ANSO1 only: one 136 KiB order-1 model per chunk
    uint32_t *aux;       // Turbo-VLC coders: two u32 per chunk (length of the first payload piece; mantissa bits)
};
#define TRC_O1_MODEL_BYTES (256u * 17u * 32u)

// table area layout (bytes from TrcWork::tables)
#define TRC_TAB_ENC   0          // uint4[256]   encoder symbol table
#define TRC_TAB_DEC   4096       // u32[256]     decoder symbol table
#define TRC_TAB_LUT   8192       // u8[32768]    slot -> symbol
#define TRC_TAB_CDF   40960      // u16[260]     sanitised CDF copy
#define TRC_TAB_SYNC  41984      // 2112 B       sync area of the encoders that gather their own payload (trc_gather.h): zero between calls
#define TRC_TAB_BYTES 45056

// static-table prep (ANS4S / RCS1 / RCS2)
void trc_launch_static_prep(const uint16_t *d_cdf, unsigned cdfnum, uint8_t *tables, hipStream_t s);

// directory scan + payload gather
void trc_launch_group_sums(const uint32_t *d_clen, uint32_t nchunks, size_t n, uint32_t chunk, uint32_t *gsum, hipStream_t s);
void trc_launch_scan_groups(const uint32_t *gsum, uint32_t ngroups, uint64_t *goff, uint64_t *d_total, hipStream_t s);
// mode 0: payload at the START of the chunk's scratch region; mode 1: at the END of it;
// mode 2: [4 + len0 bytes at the start of region A][rest at the start of region B] (RCS2), len0 = u32 at region A.
// Raw chunks (clen == chunk length) are copied from the input instead.
void trc_launch_gather(const uint8_t *d_in, size_t n, uint32_t chunk, const TrcWork &w, int from_end,
                       const uint32_t *d_clen, uint8_t *d_payload, uint64_t *d_total, hipStream_t s);

// ANS4S: static-CDF rANS (anscdf4senc / anscdf4sdec)
// returns true when the payload is already in place (the encoder's waves gathered it: trc_gather.h) -- no trc_launch_gather then
bool trc_launch_ans4s_enc(const uint8_t *d_in, size_t n, uint32_t chunk, const TrcWork &w,
                          uint32_t *d_clen, uint8_t *d_payload, uint64_t *d_total, hipStream_t s);
void trc_launch_ans4s_dec(const uint8_t *d_payload, const uint32_t *d_clen, size_t n, uint32_t chunk,
                          const TrcWork &w, uint8_t *d_out, hipStream_t s);

// RCS1 / RCS2: static-CDF range coder, 1 or 2 streams (rccdfsenc / rccdfs2enc and their decoders);
// nstreams == -1: RCSM, one stream with the 32-bit range / 16-bit I/O geometry (rccdfsmenc / rccdfsm*dec)
void trc_launch_rcs_enc(int nstreams, const uint8_t *d_in, size_t n, uint32_t chunk, const TrcWork &w,
                        uint32_t *d_clen, hipStream_t s);
void trc_launch_rcs_dec(int nstreams, const uint8_t *d_payload, const uint32_t *d_clen, size_t n, uint32_t chunk,
                        const TrcWork &w, uint8_t *d_out, hipStream_t s);

// RCB: bitwise order-0 range coder (rcsenc / rcsdec)
void trc_launch_rcb_enc(const uint8_t *d_in, size_t n, uint32_t chunk, const TrcWork &w, uint32_t *d_clen, hipStream_t s);
void trc_launch_rcb_dec(const uint8_t *d_payload, const uint32_t *d_clen, size_t n, uint32_t chunk,
                        const TrcWork &w, uint8_t *d_out, hipStream_t s);

// RCA / RCAI: adaptive-CDF byte range coder, 1 stream (rccdfenc / rccdfdec) or hi/lo nibbles on 2 streams (rccdfienc / rccdfidec);
// nibble != 0: the `turborc -n` coders on values 0..15 (rccdf4enc/dec, rccdf4ienc/idec)
void trc_launch_rca_enc(int nstreams, int nibble, const uint8_t *d_in, size_t n, uint32_t chunk, const TrcWork &w, uint32_t *d_clen, hipStream_t s);
void trc_launch_rca_dec(int nstreams, int nibble, const uint8_t *d_payload, const uint32_t *d_clen, size_t n, uint32_t chunk,
                        const TrcWork &w, uint8_t *d_out, hipStream_t s);

// RCV8 / RCVI8: "vnibble" adaptive-CDF range coders, 1 or 2 streams (rccdfenc8 / rccdfienc8 and their decoders)
void trc_launch_rcv_enc(int nstreams, const uint8_t *d_in, size_t n, uint32_t chunk, const TrcWork &w, uint32_t *d_clen, hipStream_t s);
void trc_launch_rcv_dec(int nstreams, const uint8_t *d_payload, const uint32_t *d_clen, size_t n, uint32_t chunk,
                        const TrcWork &w, uint8_t *d_out, hipStream_t s);

// ANSA: adaptive-CDF byte rANS (anscdfenc / anscdfdec); scratch2 holds the 8 B/byte record stack
// nibble != 0: anscdf4enc / anscdf4dec on values 0..15 (2 states, 4 B/byte record stack)
void trc_launch_ansa_enc(int nibble, const uint8_t *d_in, size_t n, uint32_t chunk, const TrcWork &w, uint32_t *d_clen, hipStream_t s);
void trc_launch_ansa_dec(int nibble, const uint8_t *d_payload, const uint32_t *d_clen, size_t n, uint32_t chunk,
                         const TrcWork &w, uint8_t *d_out, hipStream_t s);

void trc_launch_ansa_code(int nibble, size_t n, uint32_t chunk, const TrcWork &w, uint32_t *d_clen, hipStream_t s);   // pass 2 alone

// ANSO1: order-1 adaptive-CDF byte rANS (anscdf1enc / anscdf1dec): pass 1 with the models in HBM (w.model), then
// trc_launch_ansa_code(0, ...); scratch2 holds the same 8 B/byte record stack as ANSA
bool trc_launch_anso1_model(const uint8_t *d_in, size_t n, uint32_t chunk, const TrcWork &w, hipStream_t s);   // true: records are in the planar space
void trc_launch_ansa_code_planar(size_t n, uint32_t chunk, const TrcWork &w, uint32_t *d_clen, hipStream_t s);   // pass 2 over the planar record space (four lanes per chunk)
void trc_launch_anso1_dec(const uint8_t *d_payload, const uint32_t *d_clen, size_t n, uint32_t chunk,
                          const TrcWork &w, uint8_t *d_out, hipStream_t s);

// ANSB: bitwise order-0 rANS (ansbc / ansbd); chunks of at most 8192 bytes (one reference block); scratch2 holds the
// 16 B/byte record stack
void trc_launch_ansb_enc(const uint8_t *d_in, size_t n, uint32_t chunk, const TrcWork &w, uint32_t *d_clen, hipStream_t s);
void trc_launch_ansb_dec(const uint8_t *d_payload, const uint32_t *d_clen, size_t n, uint32_t chunk,
                         const TrcWork &w, uint8_t *d_out, hipStream_t s);

// Turbo-VLC integer coders (rccdf{u,v,vz}{enc,dec}{16,32}): variant 0 = u, 1 = v, 2 = vz; elem = 2 or 4 bytes;
// aux[2c] = length of the range-coder piece (gather mode 3)
void trc_launch_vlc_enc(int variant, int elem, const uint8_t *d_in, size_t n, uint32_t chunk, const TrcWork &w, uint32_t *d_clen, hipStream_t s);
void trc_launch_vlc_dec(int variant, int elem, const uint8_t *d_payload, const uint32_t *d_clen, size_t n, uint32_t chunk,
                        const TrcWork &w, uint8_t *d_out, hipStream_t s);

// ... over the adaptive CDF rANS (anscdf{u,uz,v,vz}{enc,dec}{16,32}): variant 0 = u, 1 = v; zz = zigzag-delta form;
// scratch2 holds, per chunk, the record stack (8 B per element) and, at the end of the slot, the mantissa bytes
void trc_launch_vla_enc(int variant, int zz, int elem, const uint8_t *d_in, size_t n, uint32_t chunk, const TrcWork &w, uint32_t *d_clen, hipStream_t s);
void trc_launch_vla_dec(int variant, int zz, int elem, const uint8_t *d_payload, const uint32_t *d_clen, size_t n, uint32_t chunk,
                        const TrcWork &w, uint8_t *d_out, hipStream_t s);

// cdfini on device
void trc_launch_hist(const uint8_t *d_in, size_t n, uint64_t *d_hist, hipStream_t s);
void trc_launch_cdf_build(const uint64_t *d_hist, size_t n_total, uint16_t *d_cdf, unsigned cdfnum, int32_t *d_status, hipStream_t s);
void trc_launch_cdfini(const uint8_t *d_in, size_t n, uint16_t *d_cdf, unsigned cdfnum,
                       int32_t *d_status, uint64_t *d_hist /*256 u64*/, hipStream_t s);
