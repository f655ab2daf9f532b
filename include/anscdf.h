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

LIBAPI size_t anscdf4senc(unsigned char *in, size_t inlen, unsigned char *out, cdf_t *cdf);
LIBAPI size_t anscdf4sdec(unsigned char *in, size_t outlen, unsigned char *out, cdf_t *cdf);

LIBAPI size_t anscdf4senc0(unsigned char *in, size_t inlen, unsigned char *out, cdf_t *cdf);
LIBAPI size_t anscdf4sdec0(unsigned char *in, size_t outlen, unsigned char *out, cdf_t *cdf);
LIBAPI size_t anscdf4sencs(unsigned char *in, size_t inlen, unsigned char *out, cdf_t *cdf);
LIBAPI size_t anscdf4sdecs(unsigned char *in, size_t outlen, unsigned char *out, cdf_t *cdf);
LIBAPI size_t anscdf4sencx(unsigned char *in, size_t inlen, unsigned char *out, cdf_t *cdf);
LIBAPI size_t anscdf4sdecx(unsigned char *in, size_t outlen, unsigned char *out, cdf_t *cdf);

#ifdef __cplusplus
}
#endif
#endif
