/*
 * turborc.h -- drop-in prototypes for the range-coder side of the hot path, served by
 * libturborc_hip.so (MI355X / gfx950).  Own text; the prototypes mirror the reference's
 * include/turborc.h so that a TurboRC-style harness compiles and links unchanged:
 *
 *   cdf_t                          reference include/turborc.h:497
 *   cdfini                         reference include/turborc.h:500   (rccdf.c:50-68)
 *   rccdfsenc / rccdfs{b,l,vb,vl}dec   include/turborc.h:502-506     (rccdf.c:71-122)
 *   rccdfs2enc / rccdfs{l,b}2dec   include/turborc.h:508-510         (rccdf.c:125-184)
 *   rccdfenc / rccdfdec            include/turborc.h:513-514         (rccdf.c:187-211)
 *   rcsenc / rcsdec                include/turborc.h:62-63           (rc_.c:37-58)
 *
 * Calling convention (reference include/turborc.h:46-59), unchanged:
 *   encoders: `out` holds at least inlen bytes (+ the harness's usual slack); the return value is
 *             the compressed length, or exactly inlen when the data is incompressible, in which
 *             case out[0..inlen) is a copy of the input and the caller must memcpy instead of
 *             calling the decoder;
 *   decoders: return outlen.
 * Stream format: the TRC1 chunk container of include/trc_hip.h -- every chunk's payload is
 * bit-identical to what the reference function returns for that chunk alone.
 * Errors (no HIP device, HIP failure, malformed container): message on stderr, trc_last_error(),
 * return value 0 (cdfini: -1).  The library never calls exit() and has no CPU coding path.
 */
#ifndef TURBORC_H_
#define TURBORC_H_
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef unsigned short cdf_t;

int cdfini(unsigned char *in, size_t inlen, cdf_t *cdf, unsigned cdfnum);

/* static-CDF range coder, one stream (reference rccdf.c:71-122).  The four reference decoders differ
 * only in how they search the CDF (linear / binary / division+linear / division+binary) and decode
 * the same stream; all four names are served by the same kernel. */
size_t rccdfsenc(unsigned char *src, size_t srclen, unsigned char *dst, cdf_t *cdf, unsigned cdfnum);
size_t rccdfsldec(unsigned char *src, size_t dstlen, unsigned char *dst, cdf_t *cdf, unsigned cdfnum);
size_t rccdfsbdec(unsigned char *src, size_t dstlen, unsigned char *dst, cdf_t *cdf, unsigned cdfnum);
size_t rccdfsvldec(unsigned char *src, size_t dstlen, unsigned char *dst, cdf_t *cdf, unsigned cdfnum);
size_t rccdfsvbdec(unsigned char *src, size_t dstlen, unsigned char *dst, cdf_t *cdf, unsigned cdfnum);

/* the one-stream static coder with a 32-bit range and 16-bit I/O (reference rccdf.c:648-694, include/turborc.h:521-526;
 * `turborc -e44`).  A different bitstream from rccdfsenc.  The reference decoders divide through a reciprocal table;
 * both names decode the same stream here (exact division). */
size_t rccdfsmenc(unsigned char *src, size_t srclen, unsigned char *dst, cdf_t *cdf, unsigned cdfnum);
size_t rccdfsmldec(unsigned char *src, size_t dstlen, unsigned char *dst, cdf_t *cdf, unsigned cdfnum);
size_t rccdfsmbdec(unsigned char *src, size_t dstlen, unsigned char *dst, cdf_t *cdf, unsigned cdfnum);

/* static-CDF range coder, two interleaved streams (reference rccdf.c:125-184; `turborc -e45`) */
size_t rccdfs2enc(unsigned char *src, size_t srclen, unsigned char *dst, cdf_t *cdf, unsigned cdfnum);
size_t rccdfsl2dec(unsigned char *src, size_t dstlen, unsigned char *dst, cdf_t *cdf, unsigned cdfnum);
size_t rccdfsb2dec(unsigned char *src, size_t dstlen, unsigned char *dst, cdf_t *cdf, unsigned cdfnum);

/* adaptive-CDF byte range coder (reference rccdf.c:187-211; `turborc -e46`) */
size_t rccdfenc(unsigned char *src, size_t srclen, unsigned char *dst);
size_t rccdfdec(unsigned char *src, size_t dstlen, unsigned char *dst);

/* adaptive-CDF byte range coder, hi nibbles on stream 0 / lo nibbles on stream 1 (reference rccdf.c:213-249,
 * include/turborc.h:515-516; `turborc -e47`) */
size_t rccdfienc(unsigned char *src, size_t srclen, unsigned char *dst);
size_t rccdfidec(unsigned char *src, size_t dstlen, unsigned char *dst);

/* the `turborc -n` coders: adaptive-CDF range coder on values 0..15, one CDF16 table (reference rccdf.c:250-275 and,
 * two interleaved streams, rccdf.c:277-323; include/turborc.h:517-519,528-529; harness ids 46/47 when the data is
 * nibble-valued, turborc.c:499-501).  Values above 15 are outside the reference's contract; their low nibble is coded. */
size_t rccdf4enc(unsigned char *src, size_t srclen, unsigned char *dst);
size_t rccdf4dec(unsigned char *src, size_t dstlen, unsigned char *dst);
size_t rccdf4ienc(unsigned char *src, size_t srclen, unsigned char *dst);
size_t rccdf4idec(unsigned char *src, size_t dstlen, unsigned char *dst);

/* Turbo-VLC integer coders over the adaptive CDF range coder (reference rccdf.c:391-632, include/turborc.h:536-549;
 * `turborc -e50/52/53` on 16- or 32-bit input): u = 6-bit exponent, v = 7-bit exponent, vz = v on the zigzag of the
 * delta to the previous element.  inlen/outlen are BYTES (multiples of the element size). */
size_t rccdfuenc16(unsigned char *src, size_t srclen, unsigned char *dst);   size_t rccdfudec16(unsigned char *src, size_t dstlen, unsigned char *dst);
size_t rccdfuenc32(unsigned char *src, size_t srclen, unsigned char *dst);   size_t rccdfudec32(unsigned char *src, size_t dstlen, unsigned char *dst);
size_t rccdfvenc16(unsigned char *src, size_t srclen, unsigned char *dst);   size_t rccdfvdec16(unsigned char *src, size_t dstlen, unsigned char *dst);
size_t rccdfvenc32(unsigned char *src, size_t srclen, unsigned char *dst);   size_t rccdfvdec32(unsigned char *src, size_t dstlen, unsigned char *dst);
size_t rccdfvzenc16(unsigned char *src, size_t srclen, unsigned char *dst);  size_t rccdfvzdec16(unsigned char *src, size_t dstlen, unsigned char *dst);
size_t rccdfvzenc32(unsigned char *src, size_t srclen, unsigned char *dst);  size_t rccdfvzdec32(unsigned char *src, size_t dstlen, unsigned char *dst);

/* "vnibble" coders (reference rccdf.c:326-390, include/turborc.h:531-534; `turborc -e48 / -e49`): a byte becomes one to
 * three CDF16 symbols on three adaptive tables (0-12 | 13,14 + nibble | 15 + two nibbles) -- for data that is mostly small
 * values; the `i` form codes the middle symbols on a second interleaved stream */
size_t rccdfenc8(unsigned char *src, size_t srclen, unsigned char *dst);     size_t rccdfdec8(unsigned char *src, size_t dstlen, unsigned char *dst);
size_t rccdfienc8(unsigned char *src, size_t srclen, unsigned char *dst);    size_t rccdfidec8(unsigned char *src, size_t dstlen, unsigned char *dst);

/* bitwise order-0 range coder, "s" predictor (reference rc_.c:37-58; `turborc -e1`, file codec 1) */
size_t rcsenc(unsigned char *src, size_t srclen, unsigned char *dst);
size_t rcsdec(unsigned char *src, size_t dstlen, unsigned char *dst);

#ifdef __cplusplus
}
#endif
#endif
