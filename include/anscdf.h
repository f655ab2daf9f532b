/*
 * anscdf.h -- drop-in prototypes for the rANS side of the hot path, served by libturborc_hip.so
 * (MI355X / gfx950).  Own text; prototypes mirror the reference's include/anscdf.h (include
 * turborc.h first: cdf_t comes from there, as in the reference, anscdf.c:30-31):
 *
 *   anscdfini                      reference include/anscdf.h:40      (anscdf.c:759-808: CPU ISA dispatch; no-op here)
 *   anscdf4senc / anscdf4sdec      reference include/anscdf.h:42-43   (anscdf.c:57-85, 810-811)
 *   ...0 / ...s / ...x             reference include/anscdf.h:70-75   direct per-ISA entry points of the
 *                                  reference (scalar / SSE / AVX2 builds of the same function, identical
 *                                  bitstreams: SURVEY F7); all three names reach the same HIP kernel here.
 *
 * Conventions: see turborc.h.  anscdf4sdec decodes the full byte alphabet (the reference decoder is
 * limited to 16 symbols).
 */
#ifndef ANSCDF_H_
#define ANSCDF_H_
#include <stddef.h>
#include "turborc.h"

#define LIBAPI

#ifdef __cplusplus
extern "C" {
#endif

void anscdfini(unsigned id);

LIBAPI size_t anscdf4senc(unsigned char *src, size_t srclen, unsigned char *dst, cdf_t *cdf);
LIBAPI size_t anscdf4sdec(unsigned char *src, size_t dstlen, unsigned char *dst, cdf_t *cdf);

LIBAPI size_t anscdf4senc0(unsigned char *src, size_t srclen, unsigned char *dst, cdf_t *cdf);
LIBAPI size_t anscdf4sdec0(unsigned char *src, size_t dstlen, unsigned char *dst, cdf_t *cdf);
LIBAPI size_t anscdf4sencs(unsigned char *src, size_t srclen, unsigned char *dst, cdf_t *cdf);
LIBAPI size_t anscdf4sdecs(unsigned char *src, size_t dstlen, unsigned char *dst, cdf_t *cdf);
LIBAPI size_t anscdf4sencx(unsigned char *src, size_t srclen, unsigned char *dst, cdf_t *cdf);
LIBAPI size_t anscdf4sdecx(unsigned char *src, size_t dstlen, unsigned char *dst, cdf_t *cdf);

/* adaptive-CDF byte rANS, 4 states (reference include/anscdf.h:46-47,76-81; anscdf.c:567-605;
 * `turborc -e56` auto, -e57 "s" build, -e58 "x" build -- identical bitstreams, one kernel here) */
LIBAPI size_t anscdfenc(unsigned char *src, size_t srclen, unsigned char *dst);
LIBAPI size_t anscdfdec(unsigned char *src, size_t dstlen, unsigned char *dst);
LIBAPI size_t anscdfenc0(unsigned char *src, size_t srclen, unsigned char *dst);
LIBAPI size_t anscdfdec0(unsigned char *src, size_t dstlen, unsigned char *dst);
LIBAPI size_t anscdfencs(unsigned char *src, size_t srclen, unsigned char *dst);
LIBAPI size_t anscdfdecs(unsigned char *src, size_t dstlen, unsigned char *dst);
LIBAPI size_t anscdfencx(unsigned char *src, size_t srclen, unsigned char *dst);
LIBAPI size_t anscdfdecx(unsigned char *src, size_t dstlen, unsigned char *dst);

/* order-1 adaptive-CDF byte rANS (reference include/anscdf.h:51-52,84-89; anscdf.c:607-645; `turborc -e64`):
 * anscdfenc with the tables selected by the previous byte */
LIBAPI size_t anscdf1enc(unsigned char *src, size_t srclen, unsigned char *dst);
LIBAPI size_t anscdf1dec(unsigned char *src, size_t dstlen, unsigned char *dst);
LIBAPI size_t anscdf1enc0(unsigned char *src, size_t srclen, unsigned char *dst);
LIBAPI size_t anscdf1dec0(unsigned char *src, size_t dstlen, unsigned char *dst);
LIBAPI size_t anscdf1encs(unsigned char *src, size_t srclen, unsigned char *dst);
LIBAPI size_t anscdf1decs(unsigned char *src, size_t dstlen, unsigned char *dst);
LIBAPI size_t anscdf1encx(unsigned char *src, size_t srclen, unsigned char *dst);
LIBAPI size_t anscdf1decx(unsigned char *src, size_t dstlen, unsigned char *dst);

/* bitwise order-0 rANS, 4 states (reference anscdf.c:672-731; `turborc -e66`).  The reference keeps these two
 * prototypes commented out in its header (include/anscdf.h:140-141) and calls them from turborc.c:536. */
LIBAPI size_t ansbc(unsigned char *src, size_t srclen, unsigned char *dst);
LIBAPI size_t ansbd(unsigned char *src, size_t dstlen, unsigned char *dst);

/* Turbo-VLC integer coders over the adaptive CDF rANS (reference include/anscdf.h:53-68,97-139; anscdf.c:139-483;
 * `turborc -e60..63` on 16/32-bit input): u/uz = 6-bit exponent (16-bit elements), v/vz = 7-bit exponent, "z" = on the
 * zigzag of the delta to the previous element.  inlen/outlen are BYTES.  The reference decoders return the compressed
 * size + 4 (its harness ignores it); these return outlen like every other decoder. */
LIBAPI size_t anscdfuenc16(unsigned char *src, size_t srclen, unsigned char *dst);
LIBAPI size_t anscdfudec16(unsigned char *src, size_t dstlen, unsigned char *dst);
LIBAPI size_t anscdfuenc160(unsigned char *src, size_t srclen, unsigned char *dst);
LIBAPI size_t anscdfudec160(unsigned char *src, size_t dstlen, unsigned char *dst);
LIBAPI size_t anscdfuenc16s(unsigned char *src, size_t srclen, unsigned char *dst);
LIBAPI size_t anscdfudec16s(unsigned char *src, size_t dstlen, unsigned char *dst);
LIBAPI size_t anscdfuenc16x(unsigned char *src, size_t srclen, unsigned char *dst);
LIBAPI size_t anscdfudec16x(unsigned char *src, size_t dstlen, unsigned char *dst);
LIBAPI size_t anscdfuzenc16(unsigned char *src, size_t srclen, unsigned char *dst);
LIBAPI size_t anscdfuzdec16(unsigned char *src, size_t dstlen, unsigned char *dst);
LIBAPI size_t anscdfuzenc160(unsigned char *src, size_t srclen, unsigned char *dst);
LIBAPI size_t anscdfuzdec160(unsigned char *src, size_t dstlen, unsigned char *dst);
LIBAPI size_t anscdfuzenc16s(unsigned char *src, size_t srclen, unsigned char *dst);
LIBAPI size_t anscdfuzdec16s(unsigned char *src, size_t dstlen, unsigned char *dst);
LIBAPI size_t anscdfuzenc16x(unsigned char *src, size_t srclen, unsigned char *dst);
LIBAPI size_t anscdfuzdec16x(unsigned char *src, size_t dstlen, unsigned char *dst);
LIBAPI size_t anscdfvenc16(unsigned char *src, size_t srclen, unsigned char *dst);
LIBAPI size_t anscdfvdec16(unsigned char *src, size_t dstlen, unsigned char *dst);
LIBAPI size_t anscdfvenc160(unsigned char *src, size_t srclen, unsigned char *dst);
LIBAPI size_t anscdfvdec160(unsigned char *src, size_t dstlen, unsigned char *dst);
LIBAPI size_t anscdfvenc16s(unsigned char *src, size_t srclen, unsigned char *dst);
LIBAPI size_t anscdfvdec16s(unsigned char *src, size_t dstlen, unsigned char *dst);
LIBAPI size_t anscdfvenc16x(unsigned char *src, size_t srclen, unsigned char *dst);
LIBAPI size_t anscdfvdec16x(unsigned char *src, size_t dstlen, unsigned char *dst);
LIBAPI size_t anscdfvzenc16(unsigned char *src, size_t srclen, unsigned char *dst);
LIBAPI size_t anscdfvzdec16(unsigned char *src, size_t dstlen, unsigned char *dst);
LIBAPI size_t anscdfvzenc160(unsigned char *src, size_t srclen, unsigned char *dst);
LIBAPI size_t anscdfvzdec160(unsigned char *src, size_t dstlen, unsigned char *dst);
LIBAPI size_t anscdfvzenc16s(unsigned char *src, size_t srclen, unsigned char *dst);
LIBAPI size_t anscdfvzdec16s(unsigned char *src, size_t dstlen, unsigned char *dst);
LIBAPI size_t anscdfvzenc16x(unsigned char *src, size_t srclen, unsigned char *dst);
LIBAPI size_t anscdfvzdec16x(unsigned char *src, size_t dstlen, unsigned char *dst);
LIBAPI size_t anscdfvenc32(unsigned char *src, size_t srclen, unsigned char *dst);
LIBAPI size_t anscdfvdec32(unsigned char *src, size_t dstlen, unsigned char *dst);
LIBAPI size_t anscdfvenc320(unsigned char *src, size_t srclen, unsigned char *dst);
LIBAPI size_t anscdfvdec320(unsigned char *src, size_t dstlen, unsigned char *dst);
LIBAPI size_t anscdfvenc32s(unsigned char *src, size_t srclen, unsigned char *dst);
LIBAPI size_t anscdfvdec32s(unsigned char *src, size_t dstlen, unsigned char *dst);
LIBAPI size_t anscdfvenc32x(unsigned char *src, size_t srclen, unsigned char *dst);
LIBAPI size_t anscdfvdec32x(unsigned char *src, size_t dstlen, unsigned char *dst);
LIBAPI size_t anscdfvzenc32(unsigned char *src, size_t srclen, unsigned char *dst);
LIBAPI size_t anscdfvzdec32(unsigned char *src, size_t dstlen, unsigned char *dst);
LIBAPI size_t anscdfvzenc320(unsigned char *src, size_t srclen, unsigned char *dst);
LIBAPI size_t anscdfvzdec320(unsigned char *src, size_t dstlen, unsigned char *dst);
LIBAPI size_t anscdfvzenc32s(unsigned char *src, size_t srclen, unsigned char *dst);
LIBAPI size_t anscdfvzdec32s(unsigned char *src, size_t dstlen, unsigned char *dst);
LIBAPI size_t anscdfvzenc32x(unsigned char *src, size_t srclen, unsigned char *dst);
LIBAPI size_t anscdfvzdec32x(unsigned char *src, size_t dstlen, unsigned char *dst);

/* adaptive-CDF nibble rANS on values 0..15, 2 states (reference include/anscdf.h:44-45,70-75; anscdf.c:87-133;
 * `turborc -n -e56/57/58`).  The decoder takes the n%4 tail from the state the encoder used (the reference's
 * decoder does not round-trip such lengths). */
LIBAPI size_t anscdf4enc(unsigned char *src, size_t srclen, unsigned char *dst);
LIBAPI size_t anscdf4dec(unsigned char *src, size_t dstlen, unsigned char *dst);
LIBAPI size_t anscdf4enc0(unsigned char *src, size_t srclen, unsigned char *dst);
LIBAPI size_t anscdf4dec0(unsigned char *src, size_t dstlen, unsigned char *dst);
LIBAPI size_t anscdf4encs(unsigned char *src, size_t srclen, unsigned char *dst);
LIBAPI size_t anscdf4decs(unsigned char *src, size_t dstlen, unsigned char *dst);
LIBAPI size_t anscdf4encx(unsigned char *src, size_t srclen, unsigned char *dst);
LIBAPI size_t anscdf4decx(unsigned char *src, size_t dstlen, unsigned char *dst);

#ifdef __cplusplus
}
#endif

/* dispatch globals of the reference (include/anscdf.h:27-35); they point at the functions above */
typedef LIBAPI size_t (*fanscdfenc)(unsigned char *src, size_t srclen, unsigned char *dst);
typedef LIBAPI size_t (*fanscdfdec)(unsigned char *src, size_t srclen, unsigned char *dst);
/* static-CDF forms (reference include/anscdf.h:29-30) */
typedef LIBAPI size_t (*fanscdf4senc)(unsigned char *src, size_t srclen, unsigned char *dst, cdf_t *cdf);
typedef LIBAPI size_t (*fanscdf4sdec)(unsigned char *src, size_t srclen, unsigned char *dst, cdf_t *cdf);
#ifdef __cplusplus
extern "C" {
#endif
extern fanscdfenc _anscdfenc;
extern fanscdfdec _anscdfdec;
extern fanscdfenc _anscdf4enc;
extern fanscdfdec _anscdf4dec;
#ifdef __cplusplus
}
#endif
#endif
