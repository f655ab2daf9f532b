/*
 * trc_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY).  See trc_oracle.h for the rules.
 *
 * Own scalar restatement of the arithmetic described in SURVEY.md section 8a; nothing here is
 * copied from the reference, each function cites the reference lines whose behaviour it follows.
 * Parity pinned against oracle/_ref (the reference compiled in place) and tests/golden/.
 */
#include "trc_oracle.h"
#include <stdlib.h>
#include <string.h>

#define PROB_BITS 15u
#define PROB_ONE  (1u << PROB_BITS)
#define TOP32     ((uint64_t)1 << 32)

static inline uint32_t ld32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint16_t ld16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }
static inline void st32(uint8_t *p, uint32_t v) { memcpy(p, &v, 4); }
static inline void st16(uint8_t *p, uint16_t v) { memcpy(p, &v, 2); }

/* "incompressible" limit shared by the range coders: OVERFLOW, rcutil_.h:129-131.
 * The reference compares pointers (op >= out + inlen*255/256 - 8); for tiny inlen the size_t
 * expression wraps, which on a flat address space is the signed comparison below. */
static inline int rc_overflow(size_t written, size_t inlen)
{
    int64_t lim = (int64_t)((inlen * 255) / 256) - 8;
    return (int64_t)written >= lim;
}

/* ------------------------------------------------------------------------------------------ */
/* M0  cdfini  (rccdf.c:50-68)                                                                 */
int orc_cdfini(const uint8_t *in, size_t inlen, uint16_t *cdf, unsigned cdfnum)
{
    uint64_t cnt[256] = {0}, best = 0, sum = 0;
    unsigned besti = 0, i;
    if (!inlen || !cdfnum || cdfnum > 256) return -1;
    for (size_t k = 0; k < inlen; k++) cnt[in[k]]++;
    for (i = 0; i < cdfnum; i++) {
        uint64_t f = (cnt[i] << PROB_BITS) / inlen;
        if (!f) f = 1;
        cnt[i] = f;
        sum += f;
        if (f > best) { best = f; besti = i; }           /* strict '>' : lowest index wins */
    }
    cnt[besti] -= sum - PROB_ONE;                        /* modular, like the size_t original */
    cdf[0] = 0;
    for (i = 0; i < cdfnum; i++) cdf[i + 1] = (uint16_t)(cdf[i] + cnt[i]);
    for (i = 0; i < cdfnum; i++) if (cdf[i] >= cdf[i + 1]) return -1;
    if (cdf[cdfnum] != PROB_ONE) return -1;
    return (int)inlen;
}

/* ------------------------------------------------------------------------------------------ */
/* M2  64-bit range coder core, 32-bit I/O, 15-bit probabilities (turborc_.h:103-158,215-229)  */
typedef struct { uint64_t range, low, mark; uint8_t *op; } rce_t;   /* mark = low at last renorm */
typedef struct { uint64_t range, code; const uint8_t *ip; } rcd_t;

static inline void rce_start(rce_t *e, uint8_t *op) { e->range = ~(uint64_t)0; e->low = e->mark = 0; e->op = op; }

/* deferred carry: if low wrapped since the last renorm, +1 ripples into the emitted words */
static inline void rce_carry(rce_t *e)
{
    if (e->mark > e->low) {
        uint8_t *p = e->op;
        uint32_t w;
        do { p -= 4; w = ld32(p) + 1; st32(p, w); } while (w == 0);
    }
}
static inline void rce_put(rce_t *e, uint32_t w) { st32(e->op, w); e->op += 4; }

static inline void rce_renorm(rce_t *e)                  /* single 'if': RC_IO=32 (_LOOP = if) */
{
    if (e->range < TOP32) {
        rce_carry(e);
        rce_put(e, (uint32_t)(e->low >> 32));
        e->low <<= 32; e->range <<= 32; e->mark = e->low;
    }
}
static inline void rce_sym(rce_t *e, uint32_t c0, uint32_t c1)      /* _rccdfenc_ + renorm */
{
    e->range >>= PROB_BITS;
    e->low += e->range * c0;
    e->range *= (c1 - c0);
    rce_renorm(e);
}
static inline void rce_finish(rce_t *e)                  /* rceflush, turborc_.h:118-128 */
{
    rce_renorm(e);
    if (e->range > ((uint64_t)1 << 33)) {
        e->low += TOP32; rce_carry(e);
        rce_put(e, (uint32_t)(e->low >> 32));
    } else {
        e->low += 1; rce_carry(e);
        rce_put(e, (uint32_t)(e->low >> 32));
        rce_put(e, (uint32_t)e->low);
    }
}
static inline void rcd_start(rcd_t *d, const uint8_t *ip)
{
    d->range = ~(uint64_t)0;
    d->code = ((uint64_t)ld32(ip) << 32) | ld32(ip + 4);
    d->ip = ip + 8;
}
static inline void rcd_renorm(rcd_t *d)
{
    if (d->range < TOP32) { d->range <<= 32; d->code = (d->code << 32) | ld32(d->ip); d->ip += 4; }
}
static inline void rcd_consume(rcd_t *d, uint32_t c0, uint32_t c1)  /* _rccdfupdate (+renorm) */
{
    uint64_t rp = d->range * c0;
    d->range = d->range * c1 - rp;
    d->code -= rp;
    rcd_renorm(d);
}
/* symbol search, identical result for the l/b/vl/vb reference decoders (turborc_.h:245,307-315) */
static inline unsigned rcd_find(const rcd_t *d, const uint16_t *cdf, unsigned cdfnum)
{
    unsigned x = 0, hi = cdfnum;
    while (x + 1 < hi) {
        unsigned mid = (x + hi) >> 1;
        if ((uint64_t)cdf[mid] * d->range > d->code) hi = mid; else x = mid;
    }
    return x;
}

/* ------------------------------------------------------------------------------------------ */
/* M3  rccdfsenc / rccdfs{l,b,vl,vb}dec  (rccdf.c:71-122)                                       */
size_t orc_rccdfsenc(const uint8_t *in, size_t inlen, uint8_t *out, const uint16_t *cdf, unsigned cdfnum)
{
    rce_t e; (void)cdfnum;
    rce_start(&e, out);
    for (size_t i = 0; i < inlen; i++) {
        unsigned x = in[i];
        rce_sym(&e, cdf[x], cdf[x + 1]);
        if (rc_overflow((size_t)(e.op - out), inlen)) { memcpy(out, in, inlen); return inlen; }
    }
    rce_finish(&e);
    return (size_t)(e.op - out);
}
size_t orc_rccdfsdec(const uint8_t *in, size_t outlen, uint8_t *out, const uint16_t *cdf, unsigned cdfnum)
{
    rcd_t d; rcd_start(&d, in);
    for (size_t i = 0; i < outlen; i++) {
        d.range >>= PROB_BITS;
        unsigned x = rcd_find(&d, cdf, cdfnum);
        rcd_consume(&d, cdf[x], cdf[x + 1]);
        out[i] = (uint8_t)x;
    }
    return outlen;
}

/* ------------------------------------------------------------------------------------------ */
/* rccdfsmenc / rccdfsm{b,l}dec (rccdf.c:648-694), `turborc -e44`: the same coder with a   */
/* 32-bit range, 16-bit I/O and 15-bit probabilities (RC_SIZE 32, RC_IO 16, turborc_.h:62-68,103-128).       */
/* Renorm below 2^16 (one step is enough: range >= 2 after the scale), flush adds 2^16 and emits one word    */
/* when range > 2^17, else adds 1 and emits two.  The reference decoders take code/range through a           */
/* reciprocal table (turborc_.h:172-190); the plain quotient below is what that table approximates, and the  */
/* symbol is the largest x with cdf[x] <= quotient (quotients >= 32768 -- possible because range>>15         */
/* truncates -- select the last symbol, as the reference's searches do).                                     */
size_t orc_rccdfsmenc(const uint8_t *in, size_t inlen, uint8_t *out, const uint16_t *cdf, unsigned cdfnum)
{
    uint32_t range = ~0u, low = 0, mark = 0;
    uint8_t *op = out;
    (void)cdfnum;
#define SM_CARRY() do { if (mark > low) { uint8_t *p_ = op; uint16_t w_; do { p_ -= 2; w_ = (uint16_t)(ld16(p_) + 1); st16(p_, w_); } while (w_ == 0); } } while (0)
#define SM_RENORM() do { if (range < (1u << 16)) { SM_CARRY(); st16(op, (uint16_t)(low >> 16)); op += 2; low <<= 16; range <<= 16; mark = low; } } while (0)
    for (size_t i = 0; i < inlen; i++) {
        unsigned x = in[i];
        range >>= PROB_BITS;
        low += range * cdf[x];
        range *= (uint32_t)cdf[x + 1] - cdf[x];
        SM_RENORM();
        if (rc_overflow((size_t)(op - out), inlen)) { memcpy(out, in, inlen); return inlen; }
    }
    SM_RENORM();
    if (range > (1u << 17)) {
        low += 1u << 16; SM_CARRY();
        st16(op, (uint16_t)(low >> 16)); op += 2;
    } else {
        low += 1; SM_CARRY();
        st16(op, (uint16_t)(low >> 16)); op += 2;
        st16(op, (uint16_t)low); op += 2;
    }
#undef SM_RENORM
#undef SM_CARRY
    return (size_t)(op - out);
}
size_t orc_rccdfsmdec(const uint8_t *in, size_t outlen, uint8_t *out, const uint16_t *cdf, unsigned cdfnum)
{
    const uint8_t *ip = in + 4;
    uint32_t range = ~0u, code = (uint32_t)ld16(in) << 16 | ld16(in + 2);
    for (size_t i = 0; i < outlen; i++) {
        range >>= PROB_BITS;
        uint32_t q = code / range;
        unsigned x = 0, hi = cdfnum;
        while (x + 1 < hi) { unsigned mid = (x + hi) >> 1; if (cdf[mid] > q) hi = mid; else x = mid; }
        uint32_t rp = range * cdf[x];
        range = range * cdf[x + 1] - rp;
        code -= rp;
        if (range < (1u << 16)) { range <<= 16; code = code << 16 | ld16(ip); ip += 2; }
        out[i] = (uint8_t)x;
    }
    return outlen;
}

/* ------------------------------------------------------------------------------------------ */
/* M4  rccdfs2enc / rccdfs{l,b}2dec  (rccdf.c:125-184)                                          */
size_t orc_rccdfs2enc(const uint8_t *in, size_t inlen, uint8_t *out, const uint16_t *cdf, unsigned cdfnum)
{
    (void)cdfnum;
    /* inlen < 4 makes the reference compute a wild stream-1 pointer ((inlen-4) wraps); every
     * input of 2..9 bytes falls into its raw branch anyway, 0/1 crash it: all of <10 is raw here */
    if (inlen < 10) { memcpy(out, in, inlen); return inlen; }
    uint8_t *base0 = out + 4, *base1 = out + 4 + ((inlen - 4) * 37) / 64;
    rce_t e0, e1; rce_start(&e0, base0); rce_start(&e1, base1);
    size_t i = 0, pairs = inlen & ~(size_t)1;
    for (; i < pairs; i += 2) {
        unsigned x0 = in[i], x1 = in[i + 1];
        rce_sym(&e0, cdf[x0], cdf[x0 + 1]);
        rce_sym(&e1, cdf[x1], cdf[x1 + 1]);
        if (rc_overflow((size_t)(e1.op - out), inlen) || e0.op >= base1) { memcpy(out, in, inlen); return inlen; }
    }
    for (; i < inlen; i++) { unsigned x = in[i]; rce_sym(&e0, cdf[x], cdf[x + 1]); }
    rce_finish(&e0);
    rce_finish(&e1);
    size_t len0 = (size_t)(e0.op - base0), len1 = (size_t)(e1.op - base1);
    st32(out, (uint32_t)len0);
    memmove(e0.op, base1, len1);
    size_t total = 4 + len0 + len1;
    if (rc_overflow(total, inlen)) { memcpy(out, in, inlen); return inlen; }
    return total;
}
size_t orc_rccdfs2dec(const uint8_t *in, size_t outlen, uint8_t *out, const uint16_t *cdf, unsigned cdfnum)
{
    rcd_t d0, d1;
    rcd_start(&d0, in + 4);
    rcd_start(&d1, in + 4 + ld32(in));
    size_t i = 0, pairs = outlen & ~(size_t)1;
    for (; i < pairs; i += 2) {
        d0.range >>= PROB_BITS; d1.range >>= PROB_BITS;
        unsigned x0 = rcd_find(&d0, cdf, cdfnum), x1 = rcd_find(&d1, cdf, cdfnum);
        rcd_consume(&d0, cdf[x0], cdf[x0 + 1]);
        rcd_consume(&d1, cdf[x1], cdf[x1 + 1]);
        out[i] = (uint8_t)x0; out[i + 1] = (uint8_t)x1;
    }
    for (; i < outlen; i++) {
        d0.range >>= PROB_BITS;
        unsigned x = rcd_find(&d0, cdf, cdfnum);
        rcd_consume(&d0, cdf[x], cdf[x + 1]);
        out[i] = (uint8_t)x;
    }
    return outlen;
}

/* ------------------------------------------------------------------------------------------ */
/* M1  adaptive 16-symbol CDF, rate 7 (cdf_.h:25-41 init, :46-50 / :87-97 SIMD update rule).   */
typedef struct { uint16_t hi[17]; uint16_t lo[16][17]; } nibmodel_t;

static void nib_reset(nibmodel_t *m)
{
    for (int j = 0; j <= 16; j++) {
        m->hi[j] = (uint16_t)(j << 11);
        for (int i = 0; i < 16; i++) m->lo[i][j] = (uint16_t)(j << 11);
    }
}
/* after coding symbol x whose lower bound was c = t[x] (read BEFORE the update):
 * every entry moves 1/128 of the way to 10*i (entries <= c) or 10*i + 32736 (entries > c),
 * int16 lanes, arithmetic shift. */
static inline void nib_adapt(uint16_t *t, unsigned x)
{
    int16_t c = (int16_t)t[x];
    for (int i = 0; i < 16; i++) {
        int16_t v = (int16_t)t[i];
        int16_t d = (int16_t)((int16_t)(10 * i) - v);
        if (v > c) d = (int16_t)(d + 32736);
        t[i] = (uint16_t)(int16_t)(v + (d >> 7));
    }
}

/* ------------------------------------------------------------------------------------------ */
/* M5  rccdfenc / rccdfdec  (rccdf.c:187-211, rccdf_.h:28-34,48-54, turborc_.h:259-304)         */
size_t orc_rccdfenc(const uint8_t *in, size_t inlen, uint8_t *out)
{
    nibmodel_t m; nib_reset(&m);
    rce_t e; rce_start(&e, out);
    for (size_t i = 0; i < inlen; i++) {
        unsigned h = in[i] >> 4, l = in[i] & 15;
        rce_sym(&e, m.hi[h], m.hi[h + 1]);       nib_adapt(m.hi, h);
        uint16_t *t = m.lo[h];
        rce_sym(&e, t[l], t[l + 1]);             nib_adapt(t, l);
        if (rc_overflow((size_t)(e.op - out), inlen)) { memcpy(out, in, inlen); return inlen; }
    }
    rce_finish(&e);
    return (size_t)(e.op - out);
}
static inline unsigned rcd_nibble(rcd_t *d, uint16_t *t)
{
    unsigned x = 0;
    d->range >>= PROB_BITS;
    while (x < 15 && (uint64_t)t[x + 1] * d->range <= d->code) x++;  /* first i with t[i+1]*r > code, else 15 */
    rcd_consume(d, t[x], t[x + 1]);
    nib_adapt(t, x);
    return x;
}
size_t orc_rccdfdec(const uint8_t *in, size_t outlen, uint8_t *out)
{
    nibmodel_t m; nib_reset(&m);
    rcd_t d; rcd_start(&d, in);
    for (size_t i = 0; i < outlen; i++) {
        unsigned h = rcd_nibble(&d, m.hi);
        unsigned l = rcd_nibble(&d, m.lo[h]);
        out[i] = (uint8_t)(h << 4 | l);
    }
    return outlen;
}

/* ------------------------------------------------------------------------------------------ */
/* rccdfienc / rccdfidec (rccdf.c:213-249; cdf8e2/cdf8d2 rccdf_.h:35-40,64-76), */
/* `turborc -e47`: hi nibbles -> stream 0, lo nibbles -> stream 1 (base out+4+inlen/2), one model. */
/* OVERFLOWI after every full group of 4 bytes only; both streams are written into `out` itself   */
/* exactly as the reference does (out needs inlen + 64 bytes).                                     */
size_t orc_rccdfienc(const uint8_t *in, size_t inlen, uint8_t *out)
{
    nibmodel_t m; nib_reset(&m);
    uint8_t *base0 = out + 4, *base1 = out + 4 + inlen / 2;
    rce_t e0, e1; rce_start(&e0, base0); rce_start(&e1, base1);
    size_t i = 0, groups = inlen & ~(size_t)3;
    for (; i < inlen; i++) {
        unsigned h = in[i] >> 4, l = in[i] & 15;
        rce_sym(&e0, m.hi[h], m.hi[h + 1]);       nib_adapt(m.hi, h);
        uint16_t *t = m.lo[h];
        rce_sym(&e1, t[l], t[l + 1]);             nib_adapt(t, l);
        if (i < groups && (i & 3) == 3)
            if (rc_overflow((size_t)(e1.op - out), inlen) || e0.op >= base1) { memcpy(out, in, inlen); return inlen; }
    }
    rce_finish(&e0);
    rce_finish(&e1);
    size_t len0 = (size_t)(e0.op - base0), len1 = (size_t)(e1.op - base1);
    st32(out, (uint32_t)len0);
    memmove(e0.op, base1, len1);
    size_t total = 4 + len0 + len1;
    if (rc_overflow(total, inlen)) { memcpy(out, in, inlen); return inlen; }
    return total;
}
size_t orc_rccdfidec(const uint8_t *in, size_t outlen, uint8_t *out)
{
    nibmodel_t m; nib_reset(&m);
    rcd_t d0, d1;
    rcd_start(&d0, in + 4);
    rcd_start(&d1, in + 4 + ld32(in));
    for (size_t i = 0; i < outlen; i++) {
        unsigned h = rcd_nibble(&d0, m.hi);
        unsigned l = rcd_nibble(&d1, m.lo[h]);
        out[i] = (uint8_t)(h << 4 | l);
    }
    return outlen;
}

/* ------------------------------------------------------------------------------------------ */
/* "vnibble" coders rccdfenc8 / rccdfdec8 (`turborc -e48`) and rccdfienc8 / rccdfidec8 (`-e49`):    */
/* rccdf.c:326-390, symbol split cdfe8 / cdfd8 rccdf_.h:76-98.  A byte x becomes 1-3 CDF16 symbols */
/* on three adaptive tables m0, m1, m2:                                                            */
/*     x < 13        : m0 <- x                                                                     */
/*     13 <= x < 45  : m0 <- 13 + ((x-13) >> 4)   (13 or 14),   m1 <- (x-13) & 15                   */
/*     x >= 45       : m0 <- 15,   m1 <- (x-45) >> 4 (0..13),   m2 <- (x-45) & 15                   */
/* One-stream form: all symbols on one range coder, OVERFLOW after every byte.  Interleaved form:   */
/* the m0 and m2 symbols on stream 0 (base out+4), the m1 symbols on stream 1 (base out+4+inlen*37/64),*/
/* OVERFLOWI after every full group of 4 bytes, header = u32 len0, OVERFLOW on the total; both     */
/* streams are written into `out` itself as the reference does (out needs inlen + 64 bytes).        */
typedef struct { uint16_t t[3][17]; } vnibmodel_t;
static void vnib_reset(vnibmodel_t *m) { for (int k = 0; k < 3; k++) for (int j = 0; j <= 16; j++) m->t[k][j] = (uint16_t)(j << 11); }
static inline void vnib_put(rce_t *e, uint16_t *t, unsigned y) { rce_sym(e, t[y], t[y + 1]); nib_adapt(t, y); }
static inline void vnib_enc_byte(vnibmodel_t *m, rce_t *e0, rce_t *e1, unsigned x)
{
    if (x < 13) vnib_put(e0, m->t[0], x);
    else if (x < 13 + 32) { x -= 13; vnib_put(e0, m->t[0], (x >> 4) + 13); vnib_put(e1, m->t[1], x & 15); }
    else { x -= 13 + 32; vnib_put(e0, m->t[0], 15); vnib_put(e1, m->t[1], x >> 4); vnib_put(e0, m->t[2], x & 15); }
}
static inline unsigned vnib_dec_byte(vnibmodel_t *m, rcd_t *d0, rcd_t *d1)
{
    unsigned x = rcd_nibble(d0, m->t[0]);
    if (x >= 13) {
        unsigned y = rcd_nibble(d1, m->t[1]);
        if (x != 15) x = (((x - 13) << 4) | y) + 13;
        else { x = rcd_nibble(d0, m->t[2]); x = ((y << 4) | x) + 13 + 32; }
    }
    return x;
}
size_t orc_rccdfenc8(const uint8_t *in, size_t inlen, uint8_t *out)
{
    vnibmodel_t m; vnib_reset(&m);
    rce_t e; rce_start(&e, out);
    for (size_t i = 0; i < inlen; i++) {
        vnib_enc_byte(&m, &e, &e, in[i]);
        if (rc_overflow((size_t)(e.op - out), inlen)) { memcpy(out, in, inlen); return inlen; }
    }
    rce_finish(&e);
    return (size_t)(e.op - out);
}
size_t orc_rccdfdec8(const uint8_t *in, size_t outlen, uint8_t *out)
{
    vnibmodel_t m; vnib_reset(&m);
    rcd_t d; rcd_start(&d, in);
    for (size_t i = 0; i < outlen; i++) out[i] = (uint8_t)vnib_dec_byte(&m, &d, &d);
    return outlen;
}
size_t orc_rccdfienc8(const uint8_t *in, size_t inlen, uint8_t *out)
{
    vnibmodel_t m; vnib_reset(&m);
    uint8_t *base0 = out + 4, *base1 = out + 4 + inlen * 37 / 64;
    rce_t e0, e1; rce_start(&e0, base0); rce_start(&e1, base1);
    size_t groups = inlen & ~(size_t)3;
    for (size_t i = 0; i < inlen; i++) {
        vnib_enc_byte(&m, &e0, &e1, in[i]);
        if (i < groups && (i & 3) == 3)
            if (rc_overflow((size_t)(e1.op - out), inlen) || e0.op >= base1) { memcpy(out, in, inlen); return inlen; }
    }
    rce_finish(&e0);
    rce_finish(&e1);
    size_t len0 = (size_t)(e0.op - base0), len1 = (size_t)(e1.op - base1);
    /* the reference lets stream 0's last bytes run into stream 1's region when stream 0 ends within ~11 bytes past it (the
     * tail and the flush are not tested): what it returns then cannot be decoded.  Stored raw here, like every other
     * chunk the coder cannot represent (and the kernels, which keep the streams apart, do the same). */
    if (e0.op > base1) { memcpy(out, in, inlen); return inlen; }
    st32(out, (uint32_t)len0);
    memmove(e0.op, base1, len1);
    size_t total = 4 + len0 + len1;
    if (rc_overflow(total, inlen)) { memcpy(out, in, inlen); return inlen; }
    return total;
}
size_t orc_rccdfidec8(const uint8_t *in, size_t outlen, uint8_t *out)
{
    vnibmodel_t m; vnib_reset(&m);
    rcd_t d0, d1;
    rcd_start(&d0, in + 4);
    rcd_start(&d1, in + 4 + ld32(in));
    for (size_t i = 0; i < outlen; i++) out[i] = (uint8_t)vnib_dec_byte(&m, &d0, &d1);
    return outlen;
}

/* ------------------------------------------------------------------------------------------ */
/* M6  32-bit rANS core, 16-bit renorm, 15-bit scale (anscdf_.h:33-48,90-94)                   */
#define ANS_LO (1u << 15)
static inline void ans_put(uint32_t *st, uint32_t c0, uint32_t f, uint8_t **ep)
{
    uint32_t s = *st;
    if (s >= (f << 16)) { *ep -= 2; st16(*ep, (uint16_t)s); s >>= 16; }
    uint32_t q = s / f;
    *st = s + (q << PROB_BITS) - q * f + c0;
}

/* ------------------------------------------------------------------------------------------ */
/* M7  anscdf4senc / anscdf4sdec  (anscdf.c:57-85).  The pointer test of anscdf.c:63,66 (ep vs  */
/* in) is not reproduced: with `out` above `in` it never fires.                      */
size_t orc_anscdf4senc(const uint8_t *in, size_t inlen, uint8_t *out, const uint16_t *cdf)
{
    size_t cap = 2 * inlen + 16;                       /* <= 1 word per symbol + 2 states */
    uint8_t *buf = (uint8_t *)malloc(cap), *ep = buf + cap;
    uint32_t st[2] = { ANS_LO, ANS_LO };
    size_t ip = inlen, body = inlen & ~(size_t)3;
    if (!buf) return 0;
    while (ip > body) { unsigned x = in[--ip]; ans_put(&st[0], cdf[x], (uint32_t)cdf[x + 1] - cdf[x], &ep); }
    while (ip > 0) {
        unsigned x;
        x = in[--ip]; ans_put(&st[1], cdf[x], (uint32_t)cdf[x + 1] - cdf[x], &ep);
        x = in[--ip]; ans_put(&st[0], cdf[x], (uint32_t)cdf[x + 1] - cdf[x], &ep);
        x = in[--ip]; ans_put(&st[1], cdf[x], (uint32_t)cdf[x + 1] - cdf[x], &ep);
        x = in[--ip]; ans_put(&st[0], cdf[x], (uint32_t)cdf[x + 1] - cdf[x], &ep);
    }
    ep -= 4; st32(ep, st[0]);
    ep -= 4; st32(ep, st[1]);
    size_t l = (size_t)(buf + cap - ep);
    if (l >= inlen) { memcpy(out, in, inlen); l = inlen; } else memcpy(out, ep, l);
    free(buf);
    return l;
}
static inline unsigned ans_get(uint32_t *st, const uint16_t *cdf, unsigned cdfnum, const uint8_t **ip)
{
    uint32_t s = *st, slot = s & (PROB_ONE - 1);
    unsigned x = 0, hi = cdfnum;                      /* largest x with cdf[x] <= slot */
    while (x + 1 < hi) { unsigned mid = (x + hi) >> 1; if (cdf[mid] > slot) hi = mid; else x = mid; }
    s = ((uint32_t)cdf[x + 1] - cdf[x]) * (s >> PROB_BITS) + slot - cdf[x];
    if (s < ANS_LO) { s = s << 16 | ld16(*ip); *ip += 2; }
    *st = s;
    return x;
}
size_t orc_anscdf4sdec(const uint8_t *in, size_t outlen, uint8_t *out, const uint16_t *cdf, unsigned cdfnum)
{
    const uint8_t *ip = in;
    uint32_t sa = ld32(ip), sb = ld32(ip + 4);        /* sa = encoder state 1, sb = encoder state 0 */
    ip += 8;
    size_t i = 0, body = outlen & ~(size_t)3;
    for (; i < body; i += 4) {
        out[i]     = (uint8_t)ans_get(&sb, cdf, cdfnum, &ip);
        out[i + 1] = (uint8_t)ans_get(&sa, cdf, cdfnum, &ip);
        out[i + 2] = (uint8_t)ans_get(&sb, cdf, cdfnum, &ip);
        out[i + 3] = (uint8_t)ans_get(&sa, cdf, cdfnum, &ip);
    }
    for (; i < outlen; i++) out[i] = (uint8_t)ans_get(&sb, cdf, cdfnum, &ip);   /* encoder coded the tail on its state 0 */
    return outlen;
}

/* ------------------------------------------------------------------------------------------ */
/* M8  anscdfenc / anscdfdec  (anscdf.c:567-605; anscdf_.h:106-162)                             */
#define ANS_BLOCK ((size_t)1 << 22)
size_t orc_anscdfenc(const uint8_t *in, size_t inlen, uint8_t *out)
{
    size_t blk = inlen < ANS_BLOCK ? inlen : ANS_BLOCK;
    uint32_t *stack = (uint32_t *)malloc((blk + 1) * 2 * sizeof(uint32_t) + 64);
    uint8_t *op = out, *oend = out + inlen;
    size_t pos = 0;
    if (!stack) return 0;
    while (pos < inlen) {
        size_t len = inlen - pos < blk ? inlen - pos : blk, k, ns = 0;
        nibmodel_t m; nib_reset(&m);
        /* pass 1 (forward): record {state id, cdf_lo, freq} per nibble, adapting as we go */
        for (k = 0; k < len + (len & 1); k += 2) {
            unsigned x0 = in[pos + k], x1 = (k + 1 < len) ? in[pos + k + 1] : 0;   /* odd tail pairs with a coded dummy 0 */
            unsigned xs[2] = { x0, x1 };
            for (int b = 0; b < 2; b++) {
                unsigned h = xs[b] >> 4, l = xs[b] & 15, sid = 3u - 2u * (unsigned)b;
                stack[ns++] = sid << 30 | (uint32_t)m.hi[h] << 15 | (uint32_t)(m.hi[h + 1] - m.hi[h]);
                nib_adapt(m.hi, h);
                uint16_t *t = m.lo[h];
                stack[ns++] = (sid - 1) << 30 | (uint32_t)t[l] << 15 | (uint32_t)(t[l + 1] - t[l]);
                nib_adapt(t, l);
            }
        }
        /* pass 2 (backward): rANS steps, words grow downward from out+inlen */
        uint32_t st[4] = { ANS_LO, ANS_LO, ANS_LO, ANS_LO };
        uint8_t *ep = oend;
        while (ns) {
            uint32_t r = stack[--ns];
            if (ep <= op + 2 + 16) goto raw;
            ans_put(&st[r >> 30], (r >> 15) & 0x7fff, r & 0x7fff, &ep);
        }
        for (k = 0; k < 4; k++) { ep -= 4; st32(ep, st[k]); }
        if (ep <= op) goto raw;
        size_t l = (size_t)(oend - ep);
        if (op + l >= oend) goto raw;
        memmove(op, ep, l); op += l;
        pos += len;
    }
    free(stack);
    return (size_t)(op - out);
raw:
    /* reference quirk (anscdf.c:573,583): for a multi-block input its raw copy starts at the
     * ADVANCED in pointer and over-reads; we copy the whole input from its true start. */
    free(stack);
    memcpy(out, in, inlen);
    return inlen;
}
static inline unsigned ansd_nibble(uint32_t *st, uint16_t *t)   /* cdf16ansdec: search + state + model */
{
    uint32_t s = *st, slot = s & (PROB_ONE - 1);
    unsigned x = 0;
    while (x < 15 && t[x + 1] <= slot) x++;
    *st = ((uint32_t)t[x + 1] - t[x]) * (s >> PROB_BITS) + slot - t[x];
    nib_adapt(t, x);
    return x;
}
static inline void ansd_renorm(uint32_t *st, const uint8_t **ip)
{
    if (*st < ANS_LO) { *st = *st << 16 | ld16(*ip); *ip += 2; }
}
size_t orc_anscdfdec(const uint8_t *in, size_t outlen, uint8_t *out)
{
    size_t blk = outlen < ANS_BLOCK ? outlen : ANS_BLOCK, pos = 0;
    const uint8_t *ip = in;
    while (pos < outlen) {
        size_t len = outlen - pos < blk ? outlen - pos : blk, k;
        nibmodel_t m; nib_reset(&m);
        uint32_t st[4];
        for (k = 0; k < 4; k++) { st[k] = ld32(ip); ip += 4; }
        for (k = 0; k < len; k += 2) {
            unsigned h0 = ansd_nibble(&st[0], m.hi), l0 = ansd_nibble(&st[1], m.lo[h0]);
            unsigned h1 = ansd_nibble(&st[2], m.hi), l1 = ansd_nibble(&st[3], m.lo[h1]);
            ansd_renorm(&st[0], &ip); ansd_renorm(&st[1], &ip);
            ansd_renorm(&st[2], &ip); ansd_renorm(&st[3], &ip);
            out[pos + k] = (uint8_t)(h0 << 4 | l0);
            if (k + 1 < len) out[pos + k + 1] = (uint8_t)(h1 << 4 | l1);
        }
        pos += len;
    }
    return outlen;
}

/* ------------------------------------------------------------------------------------------ */
/* the nibble coders (`turborc -n`, input values 0..15; harness gate m<16,     */
/* turborc.c:499-501,514-520).  One CDF16 table; values above 15 are outside the contract (the   */
/* reference indexes past its table), we code their low nibble.                                  */

/* rccdf4enc / rccdf4dec (rccdf.c:250-275; cdf4e/cdf4d rccdf_.h:28,48), `turborc -n -e46` */
size_t orc_rccdf4enc(const uint8_t *in, size_t inlen, uint8_t *out)
{
    nibmodel_t m; nib_reset(&m);
    rce_t e; rce_start(&e, out);
    for (size_t i = 0; i < inlen; i++) {
        unsigned x = in[i] & 15;
        rce_sym(&e, m.hi[x], m.hi[x + 1]);       nib_adapt(m.hi, x);
        if (rc_overflow((size_t)(e.op - out), inlen)) { memcpy(out, in, inlen); return inlen; }
    }
    rce_finish(&e);
    return (size_t)(e.op - out);
}
size_t orc_rccdf4dec(const uint8_t *in, size_t outlen, uint8_t *out)
{
    nibmodel_t m; nib_reset(&m);
    rcd_t d; rcd_start(&d, in);
    for (size_t i = 0; i < outlen; i++) out[i] = (uint8_t)rcd_nibble(&d, m.hi);
    return outlen;
}

/* rccdf4ienc / rccdf4idec (rccdf.c:277-323), `turborc -n -e47`: even positions -> stream 0, odd positions ->
 * stream 1 (base out+4+inlen/2), odd tail on stream 0.  BOTH symbols of a pair are coded with the table as it
 * was before the pair, then the table adapts to x0 and to x1 (rccdf.c:311-314).
 * Deviations from the reference, both where the reference itself does not round-trip:
 *  - its in-loop OVERFLOW is handed op1 but the function returns op0-out (rccdf.c:314,322), i.e. a
 *    meaningless length over a raw copy (small / incompressible inputs; it then crashes in the decoder);
 *  - it never tests stream 0 running into stream 1's region (short inputs: the two flushes alone can need
 *    more than inlen/2 bytes).
 *  In both cases we return inlen with a raw copy, the convention of include/turborc.h:46-59. */
size_t orc_rccdf4ienc(const uint8_t *in, size_t inlen, uint8_t *out)
{
    nibmodel_t m; nib_reset(&m);
    size_t off1 = 4 + inlen / 2, pairs = inlen & ~(size_t)1, i;
    uint8_t *s0 = (uint8_t *)malloc(2 * inlen + 32), *s1 = (uint8_t *)malloc(2 * inlen + 32);
    rce_t e0, e1;
    if (!s0 || !s1) { free(s0); free(s1); return 0; }
    rce_start(&e0, s0); rce_start(&e1, s1);
    for (i = 0; i < pairs; i += 2) {
        unsigned x0 = in[i] & 15, x1 = in[i + 1] & 15;
        rce_sym(&e0, m.hi[x0], m.hi[x0 + 1]);
        rce_sym(&e1, m.hi[x1], m.hi[x1 + 1]);
        nib_adapt(m.hi, x0);
        nib_adapt(m.hi, x1);
        if (rc_overflow(off1 + (size_t)(e1.op - s1), inlen)) goto raw;
    }
    if (i < inlen) { unsigned x = in[i] & 15; rce_sym(&e0, m.hi[x], m.hi[x + 1]); nib_adapt(m.hi, x); }
    rce_finish(&e0);
    rce_finish(&e1);
    {
        size_t len0 = (size_t)(e0.op - s0), len1 = (size_t)(e1.op - s1), total = 4 + len0 + len1;
        if (4 + len0 > off1 || rc_overflow(total, inlen)) goto raw;
        st32(out, (uint32_t)len0);
        memcpy(out + 4, s0, len0);
        memcpy(out + 4 + len0, s1, len1);
        free(s0); free(s1);
        return total;
    }
raw:
    free(s0); free(s1);
    memcpy(out, in, inlen);
    return inlen;
}
size_t orc_rccdf4idec(const uint8_t *in, size_t outlen, uint8_t *out)
{
    nibmodel_t m; nib_reset(&m);
    rcd_t d0, d1;
    size_t pairs = outlen & ~(size_t)1, i;
    rcd_start(&d0, in + 4);
    rcd_start(&d1, in + 4 + ld32(in));
    for (i = 0; i < pairs; i += 2) {
        unsigned x0 = 0, x1 = 0;
        d0.range >>= PROB_BITS; d1.range >>= PROB_BITS;
        while (x0 < 15 && (uint64_t)m.hi[x0 + 1] * d0.range <= d0.code) x0++;
        while (x1 < 15 && (uint64_t)m.hi[x1 + 1] * d1.range <= d1.code) x1++;
        rcd_consume(&d0, m.hi[x0], m.hi[x0 + 1]);
        rcd_consume(&d1, m.hi[x1], m.hi[x1 + 1]);
        nib_adapt(m.hi, x0);
        nib_adapt(m.hi, x1);
        out[i] = (uint8_t)x0; out[i + 1] = (uint8_t)x1;
    }
    if (i < outlen) out[i] = (uint8_t)rcd_nibble(&d0, m.hi);
    return outlen;
}

/* anscdf4enc / anscdf4dec (anscdf.c:87-133; mnenc4/mnflush/mndec4 anscdf_.h:106,128-141), `turborc -n -e56`:
 * adaptive nibble rANS, 2 states, 4 MiB blocks.  Groups of 4: positions 0,2 -> state 1, positions 1,3 -> state 0;
 * the n%4 tail -> state 0.  The decoder here reads the tail from the state the ENCODER used; the reference
 * decoder takes its st[0] (= encoder state 1, mnfill reverses the order), so the reference does not round-trip
 * when the block length is not a multiple of 4 -- same defect as anscdf4sdec (see orc_anscdf4sdec). */
size_t orc_anscdf4enc(const uint8_t *in, size_t inlen, uint8_t *out)
{
    size_t blk = inlen < ANS_BLOCK ? inlen : ANS_BLOCK;
    uint32_t *stack = (uint32_t *)malloc((blk + 1) * sizeof(uint32_t) + 64);
    uint8_t *op = out, *oend = out + inlen;
    size_t pos = 0;
    if (!stack) return 0;
    while (pos < inlen) {
        size_t len = inlen - pos < blk ? inlen - pos : blk, body = len & ~(size_t)3, k, ns = 0;
        nibmodel_t m; nib_reset(&m);
        for (k = 0; k < len; k++) {
            unsigned x = in[pos + k] & 15, sid = (k < body) ? 1u - (unsigned)(k & 1) : 0u;
            stack[ns++] = sid << 30 | (uint32_t)m.hi[x] << 15 | (uint32_t)(m.hi[x + 1] - m.hi[x]);
            nib_adapt(m.hi, x);
        }
        uint32_t st[2] = { ANS_LO, ANS_LO };
        uint8_t *ep = oend;
        while (ns) {
            uint32_t r = stack[--ns];
            if (ep <= op + 2 + 8) goto raw;
            ans_put(&st[r >> 30], (r >> 15) & 0x7fff, r & 0x7fff, &ep);
        }
        for (k = 0; k < 2; k++) { ep -= 4; st32(ep, st[k]); }
        if (ep <= op) goto raw;
        size_t l = (size_t)(oend - ep);
        if (op + l >= oend) goto raw;
        memmove(op, ep, l); op += l;
        pos += len;
    }
    free(stack);
    return (size_t)(op - out);
raw:
    free(stack);
    memcpy(out, in, inlen);                            /* whole input from its true start (cf. orc_anscdfenc) */
    return inlen;
}
size_t orc_anscdf4dec(const uint8_t *in, size_t outlen, uint8_t *out)
{
    size_t blk = outlen < ANS_BLOCK ? outlen : ANS_BLOCK, pos = 0;
    const uint8_t *ip = in;
    while (pos < outlen) {
        size_t len = outlen - pos < blk ? outlen - pos : blk, body = len & ~(size_t)3, k;
        nibmodel_t m; nib_reset(&m);
        uint32_t sa = ld32(ip), sb = ld32(ip + 4);     /* sa = encoder state 1, sb = encoder state 0 */
        ip += 8;
        for (k = 0; k < len; k++) {
            uint32_t *s = (k < body && !(k & 1)) ? &sa : &sb;
            out[pos + k] = (uint8_t)ansd_nibble(s, m.hi);
            ansd_renorm(s, &ip);
        }
        pos += len;
    }
    return outlen;
}

/* ------------------------------------------------------------------------------------------ */
/* anscdf1enc / anscdf1dec (anscdf.c:607-645; mnenc8x2x/mndec8x2x anscdf_.h:121-126,164-174), */
/* `turborc -e64`: the byte rANS of M8 with an order-1 model -- hi table selected by the previous byte, lo table  */
/* by (previous byte, hi nibble); 256 x (17 + 16 x 17) u16 = 148 KB, reset per 4 MiB block.  The context byte     */
/* itself is NOT reset between blocks (declared outside the block loop); an odd tail pairs with a coded dummy 0,   */
/* which also becomes the context.                                                                               */
typedef struct { uint16_t hi[256][17]; uint16_t lo[256][16][17]; } o1model_t;
static void o1_reset(o1model_t *m)
{
    for (int c = 0; c < 256; c++)
        for (int j = 0; j <= 16; j++) {
            m->hi[c][j] = (uint16_t)(j << 11);
            for (int i = 0; i < 16; i++) m->lo[c][i][j] = (uint16_t)(j << 11);
        }
}
size_t orc_anscdf1enc(const uint8_t *in, size_t inlen, uint8_t *out)
{
    size_t blk = inlen < ANS_BLOCK ? inlen : ANS_BLOCK;
    uint32_t *stack = (uint32_t *)malloc((blk + 1) * 2 * sizeof(uint32_t) + 64);
    o1model_t *m = (o1model_t *)malloc(sizeof *m);
    uint8_t *op = out, *oend = out + inlen;
    size_t pos = 0;
    unsigned cx = 0;
    if (!stack || !m) { free(stack); free(m); return 0; }
    while (pos < inlen) {
        size_t len = inlen - pos < blk ? inlen - pos : blk, k, ns = 0;
        o1_reset(m);
        for (k = 0; k < len + (len & 1); k += 2) {
            unsigned xs[2] = { in[pos + k], (k + 1 < len) ? in[pos + k + 1] : 0u };
            for (int b = 0; b < 2; b++) {
                unsigned h = xs[b] >> 4, l = xs[b] & 15, sid = 3u - 2u * (unsigned)b;
                uint16_t *th = m->hi[cx], *tl = m->lo[cx][h];
                stack[ns++] = sid << 30 | (uint32_t)th[h] << 15 | (uint32_t)(th[h + 1] - th[h]);
                nib_adapt(th, h);
                stack[ns++] = (sid - 1) << 30 | (uint32_t)tl[l] << 15 | (uint32_t)(tl[l + 1] - tl[l]);
                nib_adapt(tl, l);
                cx = xs[b];
            }
        }
        uint32_t st[4] = { ANS_LO, ANS_LO, ANS_LO, ANS_LO };
        uint8_t *ep = oend;
        while (ns) {
            uint32_t r = stack[--ns];
            if (ep <= op + 2 + 16) goto raw;
            ans_put(&st[r >> 30], (r >> 15) & 0x7fff, r & 0x7fff, &ep);
        }
        for (k = 0; k < 4; k++) { ep -= 4; st32(ep, st[k]); }
        if (ep <= op) goto raw;
        size_t l = (size_t)(oend - ep);
        if (op + l >= oend) goto raw;
        memmove(op, ep, l); op += l;
        pos += len;
    }
    free(stack); free(m);
    return (size_t)(op - out);
raw:
    free(stack); free(m);
    memcpy(out, in, inlen);                            /* whole input from its true start (cf. orc_anscdfenc) */
    return inlen;
}
size_t orc_anscdf1dec(const uint8_t *in, size_t outlen, uint8_t *out)
{
    size_t blk = outlen < ANS_BLOCK ? outlen : ANS_BLOCK, pos = 0;
    const uint8_t *ip = in;
    o1model_t *m = (o1model_t *)malloc(sizeof *m);
    unsigned cx = 0;
    if (!m) return 0;
    while (pos < outlen) {
        size_t len = outlen - pos < blk ? outlen - pos : blk, k;
        uint32_t st[4];
        o1_reset(m);
        for (k = 0; k < 4; k++) { st[k] = ld32(ip); ip += 4; }
        for (k = 0; k < len; k += 2) {
            unsigned h0 = ansd_nibble(&st[0], m->hi[cx]), l0 = ansd_nibble(&st[1], m->lo[cx][h0]);
            cx = h0 << 4 | l0;
            unsigned h1 = ansd_nibble(&st[2], m->hi[cx]), l1 = ansd_nibble(&st[3], m->lo[cx][h1]);
            cx = h1 << 4 | l1;
            ansd_renorm(&st[0], &ip); ansd_renorm(&st[1], &ip);
            ansd_renorm(&st[2], &ip); ansd_renorm(&st[3], &ip);
            out[pos + k] = (uint8_t)(h0 << 4 | l0);
            if (k + 1 < len) out[pos + k + 1] = (uint8_t)cx;
        }
        pos += len;
    }
    free(m);
    return outlen;
}

/* ------------------------------------------------------------------------------------------ */
/* ansbc / ansbd (anscdf.c:672-731; ecbe/ecbd anscdf.c:659-668), `turborc -e66`: bitwise order-0  */
/* rANS.  255-node bit tree of 15-bit probabilities P(bit=1) (init 2^14; bit 1: p += (2^15-p)>>5, bit 0: p -= p>>5),  */
/* kept across blocks; blocks of 8192 bytes: forward pass records {bit<<15 | p} per bit, backward pass codes them   */
/* on 4 states (last bit first; a byte's bits b0..b7 go to states 0,1,2,3,0,1,2,3), states restart at every block.  */
/* Block payload [st3][st2][st1][st0][u16 words in decode order].  The decoder renormalises BEFORE each bit.        */
/* Raw rule: the reference tests op+l > out+inlen after each block, so a total of exactly inlen comes back as a     */
/* coded stream that its caller then treats as raw (it cannot round-trip); here totals >= inlen are stored raw.     */
#define ANSB_BLOCK 8192u
size_t orc_ansbc(const uint8_t *in, size_t inlen, uint8_t *out)
{
    uint16_t mb[256], *rec = (uint16_t *)malloc(ANSB_BLOCK * 8 * sizeof(uint16_t));
    uint8_t *tmp = (uint8_t *)malloc(2 * ANSB_BLOCK * 8 + 64), *op = out;
    if (!rec || !tmp) { free(rec); free(tmp); return 0; }
    for (int i = 0; i < 256; i++) mb[i] = PROB_ONE >> 1;
    for (size_t pos = 0; pos < inlen; pos += ANSB_BLOCK) {
        size_t len = inlen - pos < ANSB_BLOCK ? inlen - pos : ANSB_BLOCK, nr = 0;
        for (size_t k = 0; k < len; k++) {
            unsigned cx = 0x100u | in[pos + k];
            for (int s = 8; s >= 1; s--) {
                uint16_t *m = &mb[cx >> s];
                unsigned q = *m, b = (cx >> (s - 1)) & 1;
                rec[nr++] = (uint16_t)(b << PROB_BITS | q);
                *m = (uint16_t)(b ? q + ((PROB_ONE - q) >> 5) : q - (q >> 5));
            }
        }
        uint32_t st[4] = { ANS_LO, ANS_LO, ANS_LO, ANS_LO };
        uint8_t *eend = tmp + 2 * ANSB_BLOCK * 8 + 64, *ep = eend;
        unsigned si = 0;
        while (nr) {
            unsigned r = rec[--nr], b = r >> PROB_BITS, p0 = r & (PROB_ONE - 1);
            uint32_t ls = b ? p0 : PROB_ONE - p0, s = st[si];
            if (s >= (ls << 16)) { ep -= 2; st16(ep, (uint16_t)s); s >>= 16; }
            st[si] = s + (s / ls) * (PROB_ONE - ls) + (b ? 0 : p0);
            si = (si + 1) & 3;
        }
        for (int k = 0; k < 4; k++) { ep -= 4; st32(ep, st[k]); }
        size_t l = (size_t)(eend - ep);
        if ((size_t)(op - out) + l >= inlen) { free(rec); free(tmp); memcpy(out, in, inlen); return inlen; }
        memcpy(op, ep, l); op += l;
    }
    free(rec); free(tmp);
    return (size_t)(op - out);
}
size_t orc_ansbd(const uint8_t *in, size_t outlen, uint8_t *out)
{
    uint16_t mb[256];
    const uint8_t *ip = in;
    for (int i = 0; i < 256; i++) mb[i] = PROB_ONE >> 1;
    for (size_t pos = 0; pos < outlen; pos += ANSB_BLOCK) {
        size_t len = outlen - pos < ANSB_BLOCK ? outlen - pos : ANSB_BLOCK;
        uint32_t st[4];
        for (int k = 0; k < 4; k++) { st[k] = ld32(ip); ip += 4; }
        for (size_t k = 0; k < len; k++) {
            unsigned cx = 1;
            for (int j = 0; j < 8; j++) {
                uint32_t *s = &st[j & 3];
                if (*s < ANS_LO) { *s = *s << 16 | ld16(ip); ip += 2; }
                uint16_t *m = &mb[cx];
                uint32_t p0 = *m, r = *s & (PROB_ONE - 1), rcx = (*s >> PROB_BITS) * p0;
                if (r < p0) { *s = rcx + r; *m = (uint16_t)(p0 + ((PROB_ONE - p0) >> 5)); cx = cx * 2 + 1; }
                else { *s -= rcx + p0; *m = (uint16_t)(p0 - (p0 >> 5)); cx = cx * 2; }
            }
            out[pos + k] = (uint8_t)cx;
        }
    }
    return outlen;
}

/* ------------------------------------------------------------------------------------------ */
/* Turbo-VLC integer coders over the adaptive CDF range coder (rccdf.c:391-632; cdfe7/cdfd7,       */
/* cdfe6/cdfd6 rccdf_.h:100-123; vlcenc/vlcdec/bitvrput/bitvrget include_/vlcbit.h:24-63; reverse bit I/O             */
/* rcutil_.h:163-189; zigzag rcutil_.h:142-149): 16- or 32-bit integers, `turborc -e50/52/53`.                        */
/*   u (vn = 1, "vlc6") / v (vn = 2, "vlc7") / vz (v on the zigzag of the delta to the previous element).             */
/* An element x >= 2^(vn+1) is split into an exponent symbol and f = bsr(x) - vn mantissa bits:                       */
/*   expo = ((f+1) << vn) + bits [f, f+vn) of x;  mantissa = low f bits of x.                                         */
/* The (small) value or the exponent is coded with two CDF16 tables: below T (8 for vn = 2, 12 for vn = 1) one symbol */
/* with table 0, else symbol ((x-T)>>4)+T with table 0 and (x-T)&15 with table 1.  Mantissas go MSB-first into a bit  */
/* string that grows DOWN from the end of the output (byte k of the string at out+inlen-1-k), the range coder grows up */
/* from out+4; raw as soon as the two come within 8 bytes of each other (test after every element, with the bit side  */
/* at out+inlen-8-floor(bits/8)).  Result [u32 total][range-coder words][bit bytes], OVERFLOW on the total.           */
/* A trailing partial element is zero-extended here; the reference reads the bytes that follow its input (and its      */
/* decoder writes a whole element): callers pass multiples of the element size, and that is where parity is pinned.  */
static inline unsigned bsr32(uint32_t x) { return 31u - (unsigned)__builtin_clz(x); }
static size_t vlc_enc(const uint8_t *in, size_t inlen, uint8_t *out, unsigned es, unsigned vn, int zz)
{
    const unsigned T = vn == 2 ? 8u : 12u;
    size_t nel = (inlen + es - 1) / es, bits = 0, nbytes;
    uint8_t *bitbuf = (uint8_t *)calloc(nel * 4 + 16, 1);      /* the bit string, byte k = bits 8k .. 8k+7 */
    uint8_t *rcbuf = (uint8_t *)malloc(2 * inlen + 64);
    nibmodel_t m; nib_reset(&m);                               /* tables 0 and 1 = m.hi and m.lo[0] */
    rce_t e; uint32_t prev = 0;
    if (!bitbuf || !rcbuf) { free(bitbuf); free(rcbuf); return 0; }
    rce_start(&e, rcbuf);
    for (size_t i = 0; i < nel; i++) {
        uint32_t v = 0, x;
        memcpy(&v, in + i * es, inlen - i * es < es ? inlen - i * es : es);
        if (zz) {
            uint32_t d = v - prev;
            x = es == 2 ? (uint16_t)(((int16_t)d << 1) ^ ((int16_t)d >> 15)) : (uint32_t)(((int32_t)d << 1) ^ ((int32_t)d >> 31));
            prev = v;
        } else x = v;
        if (x >= (1u << (vn + 1))) {
            unsigned f = bsr32(x) - vn, expo = ((f + 1) << vn) + ((x >> f) & ((1u << vn) - 1));
            uint32_t ma = x & ((1u << f) - 1);
            for (unsigned k = 0; k < f; k++, bits++)           /* MSB first */
                if ((ma >> (f - 1 - k)) & 1) bitbuf[bits >> 3] |= (uint8_t)(0x80u >> (bits & 7));
            x = expo;
        }
        if (x < T) { rce_sym(&e, m.hi[x], m.hi[x + 1]); nib_adapt(m.hi, x); }
        else {
            unsigned y = ((x - T) >> 4) + T, z = (x - T) & 15;
            rce_sym(&e, m.hi[y], m.hi[y + 1]); nib_adapt(m.hi, y);
            rce_sym(&e, m.lo[0][z], m.lo[0][z + 1]); nib_adapt(m.lo[0], z);
        }
        if ((int64_t)(4 + (e.op - rcbuf)) + 8 >= (int64_t)inlen - 8 - (int64_t)(bits >> 3)) goto raw;
    }
    rce_finish(&e);
    nbytes = (bits + 7) >> 3;
    {
        size_t rc = (size_t)(e.op - rcbuf), total = 4 + rc + nbytes;
        if (rc_overflow(total, inlen)) goto raw;
        st32(out, (uint32_t)total);
        memcpy(out + 4, rcbuf, rc);
        for (size_t k = 0; k < nbytes; k++) out[total - 1 - k] = bitbuf[k];
        free(bitbuf); free(rcbuf);
        return total;
    }
raw:
    free(bitbuf); free(rcbuf);
    memcpy(out, in, inlen);
    return inlen;
}
static size_t vlc_dec(const uint8_t *in, size_t outlen, uint8_t *out, unsigned es, unsigned vn, int zz)
{
    const unsigned T = vn == 2 ? 8u : 12u;
    size_t nel = (outlen + es - 1) / es, bits = 0, total = ld32(in);
    nibmodel_t m; nib_reset(&m);
    rcd_t d; rcd_start(&d, in + 4);
    uint32_t prev = 0;
    for (size_t i = 0; i < nel; i++) {
        uint32_t x = rcd_nibble(&d, m.hi), v;
        if (x >= T) { unsigned z = rcd_nibble(&d, m.lo[0]); x = ((x - T) << 4 | z) + T; }
        if (x >= (1u << (vn + 1))) {
            unsigned f = (x >> vn) - 1;
            uint32_t ma = 0;
            for (unsigned k = 0; k < f; k++, bits++)
                ma = ma << 1 | ((in[total - 1 - (bits >> 3)] >> (7 - (bits & 7))) & 1);
            x = (((1u << vn) + (x & ((1u << vn) - 1))) << f) + ma;
        }
        if (zz) {
            v = es == 2 ? (uint16_t)(prev + (uint16_t)((x >> 1) ^ (0u - (x & 1)))) : prev + ((x >> 1) ^ (0u - (x & 1)));
            prev = v;
        } else v = x;
        size_t nb = outlen - i * es < es ? outlen - i * es : es;   /* a partial last element: only its valid bytes */
        memcpy(out + i * es, &v, nb);
    }
    return outlen;
}
#define VLC_PAIR(name, es, vn, zz) \
    size_t orc_##name##enc##es(const uint8_t *in, size_t inlen, uint8_t *out) { return vlc_enc(in, inlen, out, es / 8, vn, zz); } \
    size_t orc_##name##dec##es(const uint8_t *in, size_t outlen, uint8_t *out) { return vlc_dec(in, outlen, out, es / 8, vn, zz); }
VLC_PAIR(rccdfu, 16, 1, 0)
VLC_PAIR(rccdfu, 32, 1, 0)
VLC_PAIR(rccdfv, 16, 2, 0)
VLC_PAIR(rccdfv, 32, 2, 0)
VLC_PAIR(rccdfvz, 16, 2, 1)
VLC_PAIR(rccdfvz, 32, 2, 1)

/* ------------------------------------------------------------------------------------------ */
/* SURVEY 8f rank 3, second half: the same Turbo-VLC integer coders over the adaptive CDF rANS (anscdf.c:139-483;     */
/* cdfenc6/7, cdfdec6/7 anscdf_.h:205-230; mnflush :128-138), `turborc -e60..63`: anscdfu/uz (vn = 1) on 16-bit,       */
/* anscdfv/vz (vn = 2) on 16- and 32-bit elements; "z" = zigzag of the delta to the previous element.                 */
/* Per block of 4 Mi ELEMENTS: tables reset, a forward pass appends the mantissas to the bit string (which, like the   */
/* previous element of the z variants, runs on across blocks) and records one or two symbols per element -- the first */
/* on rANS state 1, the second on state 0 -- then the records are coded backwards, words growing down from 16 bytes    */
/* below the bit string's current end, and the block payload [st1][st0][u16 words] is moved up behind the previous     */
/* one.  Result [u32 total][block payloads][bit bytes].  Raw rules (all pointer tests of the reference, as offsets):   */
/* before every record words + 30 + floor(bits/8) >= inlen; after a block 4 + payloads + 16 + floor(bits/8) >= inlen;  */
/* at the end total >= inlen.  A trailing partial element is zero-extended (cf. vlc_enc).                             */
static size_t vlca_enc(const uint8_t *in, size_t inlen, uint8_t *out, unsigned es, unsigned vn, int zz)
{
    const unsigned T = vn == 2 ? 8u : 12u;
    size_t nel = (inlen + es - 1) / es, blk = nel < ANS_BLOCK ? nel : ANS_BLOCK, bits = 0, op = 4, nbytes;
    uint8_t *bitbuf = (uint8_t *)calloc(nel * 4 + 16, 1);
    uint8_t *body = (uint8_t *)malloc(inlen + 64), *tmp = (uint8_t *)malloc(4 * blk + 64);
    uint32_t *stack = (uint32_t *)malloc((2 * blk + 2) * sizeof(uint32_t)), prev = 0;
    if (!bitbuf || !body || !tmp || !stack) { free(bitbuf); free(body); free(tmp); free(stack); return 0; }
    for (size_t pos = 0; pos < nel; pos += blk) {
        size_t cnt = nel - pos < blk ? nel - pos : blk, ns = 0;
        nibmodel_t m; nib_reset(&m);                               /* tables 0 and 1 = m.hi and m.lo[0] */
        for (size_t i = pos; i < pos + cnt; i++) {
            uint32_t v = 0, x;
            memcpy(&v, in + i * es, inlen - i * es < es ? inlen - i * es : es);
            if (zz) {
                uint32_t d = v - prev;
                x = es == 2 ? (uint16_t)(((int16_t)d << 1) ^ ((int16_t)d >> 15)) : (uint32_t)(((int32_t)d << 1) ^ ((int32_t)d >> 31));
                prev = v;
            } else x = v;
            if (x >= (1u << (vn + 1))) {
                unsigned f = bsr32(x) - vn, expo = ((f + 1) << vn) + ((x >> f) & ((1u << vn) - 1));
                uint32_t ma = x & ((1u << f) - 1);
                for (unsigned k = 0; k < f; k++, bits++)
                    if ((ma >> (f - 1 - k)) & 1) bitbuf[bits >> 3] |= (uint8_t)(0x80u >> (bits & 7));
                x = expo;
            }
            if (x < T) { stack[ns++] = 1u << 30 | (uint32_t)m.hi[x] << 15 | (uint32_t)(m.hi[x + 1] - m.hi[x]); nib_adapt(m.hi, x); }
            else {
                unsigned y = ((x - T) >> 4) + T, z = (x - T) & 15;
                stack[ns++] = 1u << 30 | (uint32_t)m.hi[y] << 15 | (uint32_t)(m.hi[y + 1] - m.hi[y]); nib_adapt(m.hi, y);
                stack[ns++] = (uint32_t)m.lo[0][z] << 15 | (uint32_t)(m.lo[0][z + 1] - m.lo[0][z]); nib_adapt(m.lo[0], z);
            }
        }
        /* mnflush(op, bp - 8, ...) with bp = out + inlen - 8 - floor(bits/8): everything as offsets from out */
        int64_t top = (int64_t)inlen - 16 - (int64_t)(bits >> 3), ep = top;
        uint32_t st[2] = { ANS_LO, ANS_LO };
        uint8_t *tend = tmp + 4 * blk + 64, *tp = tend;
        while (ns) {
            uint32_t r = stack[--ns];
            if (ep <= (int64_t)op + 2 + 8) goto raw;
            uint8_t *before = tp;
            ans_put(&st[r >> 30], (r >> 15) & 0x7fff, r & 0x7fff, &tp);
            ep -= before - tp;
        }
        for (int k = 0; k < 2; k++) { tp -= 4; st32(tp, st[k]); ep -= 4; }
        if (ep <= (int64_t)op) goto raw;
        size_t l = (size_t)(tend - tp);
        if ((int64_t)op + (int64_t)l >= top) goto raw;
        memcpy(body + op, tp, l); op += l;
    }
    nbytes = (bits + 7) >> 3;
    if (op + nbytes >= inlen) goto raw;
    {
        size_t total = op + nbytes;
        memcpy(out + 4, body + 4, op - 4);
        st32(out, (uint32_t)total);
        for (size_t k = 0; k < nbytes; k++) out[total - 1 - k] = bitbuf[k];
        free(bitbuf); free(body); free(tmp); free(stack);
        return total;
    }
raw:
    free(bitbuf); free(body); free(tmp); free(stack);
    memcpy(out, in, inlen);
    return inlen;
}
static size_t vlca_dec(const uint8_t *in, size_t outlen, uint8_t *out, unsigned es, unsigned vn, int zz)
{
    const unsigned T = vn == 2 ? 8u : 12u;
    size_t nel = (outlen + es - 1) / es, blk = nel < ANS_BLOCK ? nel : ANS_BLOCK, bits = 0, total = ld32(in);
    const uint8_t *ip = in + 4;
    uint32_t prev = 0;
    for (size_t pos = 0; pos < nel; pos += blk) {
        size_t cnt = nel - pos < blk ? nel - pos : blk;
        nibmodel_t m; nib_reset(&m);
        uint32_t sa = ld32(ip), sb = ld32(ip + 4);                 /* sa = encoder state 1 (first symbols), sb = state 0 */
        ip += 8;
        for (size_t i = pos; i < pos + cnt; i++) {
            uint32_t x = ansd_nibble(&sa, m.hi), v;
            ansd_renorm(&sa, &ip);
            if (x >= T) { unsigned z = ansd_nibble(&sb, m.lo[0]); ansd_renorm(&sb, &ip); x = ((x - T) << 4 | z) + T; }
            if (x >= (1u << (vn + 1))) {
                unsigned f = (x >> vn) - 1;
                uint32_t ma = 0;
                for (unsigned k = 0; k < f; k++, bits++)
                    ma = ma << 1 | ((in[total - 1 - (bits >> 3)] >> (7 - (bits & 7))) & 1);
                x = (((1u << vn) + (x & ((1u << vn) - 1))) << f) + ma;
            }
            if (zz) {
                v = es == 2 ? (uint16_t)(prev + (uint16_t)((x >> 1) ^ (0u - (x & 1)))) : prev + ((x >> 1) ^ (0u - (x & 1)));
                prev = v;
            } else v = x;
            size_t nb = outlen - i * es < es ? outlen - i * es : es;
            memcpy(out + i * es, &v, nb);
        }
    }
    return outlen;
}
#define VLCA_PAIR(name, es, vn, zz) \
    size_t orc_##name##enc##es(const uint8_t *in, size_t inlen, uint8_t *out) { return vlca_enc(in, inlen, out, es / 8, vn, zz); } \
    size_t orc_##name##dec##es(const uint8_t *in, size_t outlen, uint8_t *out) { return vlca_dec(in, outlen, out, es / 8, vn, zz); }
VLCA_PAIR(anscdfu, 16, 1, 0)
VLCA_PAIR(anscdfuz, 16, 1, 1)
VLCA_PAIR(anscdfv, 16, 2, 0)
VLCA_PAIR(anscdfvz, 16, 2, 1)
VLCA_PAIR(anscdfv, 32, 2, 0)
VLCA_PAIR(anscdfvz, 32, 2, 1)

/* ------------------------------------------------------------------------------------------ */
/* M9  rcsenc / rcsdec  (rc_.c:37-58, mb_o0.h:27-41,89-112, turborc_.h:417-452, mbc_s.h:53-55)  */
static inline uint16_t bit_adapt(uint32_t p, uint32_t bit)
{
    return (uint16_t)(p - (((p - (bit ? PROB_ONE : 0u)) >> 5) + bit));   /* 32-bit unsigned, logical shift */
}
size_t orc_rcsenc(const uint8_t *in, size_t inlen, uint8_t *out)
{
    uint16_t mb[256];
    rce_t e; rce_start(&e, out);
    for (int i = 0; i < 256; i++) mb[i] = PROB_ONE >> 1;
    for (size_t i = 0; i < inlen; i++) {
        unsigned x = in[i], ctx = 1;
        for (int b = 7; b >= 0; b--) {
            if (b & 1) rce_renorm(&e);                 /* renorm only before bits 7,5,3,1 */
            uint32_t p = mb[ctx], bit = (x >> b) & 1;
            uint64_t cut = (e.range >> PROB_BITS) * p;
            if (bit) e.range = cut; else { e.low += cut; e.range -= cut; }
            mb[ctx] = bit_adapt(p, bit);
            ctx = ctx * 2 + bit;
        }
        if (rc_overflow((size_t)(e.op - out), inlen)) { memcpy(out, in, inlen); return inlen; }
    }
    rce_finish(&e);
    return (size_t)(e.op - out);
}
size_t orc_rcsdec(const uint8_t *in, size_t outlen, uint8_t *out)
{
    uint16_t mb[256];
    rcd_t d; rcd_start(&d, in);
    for (int i = 0; i < 256; i++) mb[i] = PROB_ONE >> 1;
    for (size_t i = 0; i < outlen; i++) {
        unsigned ctx = 1;
        for (int b = 7; b >= 0; b--) {
            if (b & 1) rcd_renorm(&d);
            uint32_t p = mb[ctx], bit;
            uint64_t cut = (d.range >> PROB_BITS) * p;
            if (d.code < cut) { d.range = cut; bit = 1; } else { d.range -= cut; d.code -= cut; bit = 0; }
            mb[ctx] = bit_adapt(p, bit);
            ctx = ctx * 2 + bit;
        }
        out[i] = (uint8_t)ctx;
    }
    return outlen;
}

/* ------------------------------------------------------------------------------------------ */
/* Per-chunk drivers (the product's unit of parallelism: payload(c) == coder(chunk c)).        */
static size_t enc_one(int codec, const uint8_t *in, size_t n, uint8_t *out, const uint16_t *cdf, unsigned cdfnum)
{
    switch (codec) {
    case ORC_ANS4S: return orc_anscdf4senc(in, n, out, cdf);
    case ORC_RCS1:  return orc_rccdfsenc(in, n, out, cdf, cdfnum);
    case ORC_RCS2:  return orc_rccdfs2enc(in, n, out, cdf, cdfnum);
    case ORC_RCA:   return orc_rccdfenc(in, n, out);
    case ORC_ANSA:  return orc_anscdfenc(in, n, out);
    case ORC_RCB:   return orc_rcsenc(in, n, out);
    case ORC_RCAI:  return orc_rccdfienc(in, n, out);
    case ORC_RCA4:  return orc_rccdf4enc(in, n, out);
    case ORC_RCAI4: return orc_rccdf4ienc(in, n, out);
    case ORC_ANSA4: return orc_anscdf4enc(in, n, out);
    case ORC_RCSM:  return orc_rccdfsmenc(in, n, out, cdf, cdfnum);
    case ORC_ANSO1: return orc_anscdf1enc(in, n, out);
    case ORC_ANSB:  return orc_ansbc(in, n, out);
    case ORC_VLCU16:  return orc_rccdfuenc16(in, n, out);
    case ORC_VLCU32:  return orc_rccdfuenc32(in, n, out);
    case ORC_VLCV16:  return orc_rccdfvenc16(in, n, out);
    case ORC_VLCV32:  return orc_rccdfvenc32(in, n, out);
    case ORC_VLCVZ16: return orc_rccdfvzenc16(in, n, out);
    case ORC_VLCVZ32: return orc_rccdfvzenc32(in, n, out);
    case ORC_VLAU16:  return orc_anscdfuenc16(in, n, out);
    case ORC_VLAUZ16: return orc_anscdfuzenc16(in, n, out);
    case ORC_VLAV16:  return orc_anscdfvenc16(in, n, out);
    case ORC_VLAVZ16: return orc_anscdfvzenc16(in, n, out);
    case ORC_VLAV32:  return orc_anscdfvenc32(in, n, out);
    case ORC_VLAVZ32: return orc_anscdfvzenc32(in, n, out);
    case ORC_RCV8:    return orc_rccdfenc8(in, n, out);
    case ORC_RCVI8:   return orc_rccdfienc8(in, n, out);
    }
    return 0;
}
static void dec_one(int codec, const uint8_t *in, size_t n, uint8_t *out, const uint16_t *cdf, unsigned cdfnum)
{
    switch (codec) {
    case ORC_ANS4S: orc_anscdf4sdec(in, n, out, cdf, cdfnum); break;
    case ORC_RCS1:  orc_rccdfsdec(in, n, out, cdf, cdfnum); break;
    case ORC_RCS2:  orc_rccdfs2dec(in, n, out, cdf, cdfnum); break;
    case ORC_RCA:   orc_rccdfdec(in, n, out); break;
    case ORC_ANSA:  orc_anscdfdec(in, n, out); break;
    case ORC_RCB:   orc_rcsdec(in, n, out); break;
    case ORC_RCAI:  orc_rccdfidec(in, n, out); break;
    case ORC_RCA4:  orc_rccdf4dec(in, n, out); break;
    case ORC_RCAI4: orc_rccdf4idec(in, n, out); break;
    case ORC_ANSA4: orc_anscdf4dec(in, n, out); break;
    case ORC_RCSM:  orc_rccdfsmdec(in, n, out, cdf, cdfnum); break;
    case ORC_ANSO1: orc_anscdf1dec(in, n, out); break;
    case ORC_ANSB:  orc_ansbd(in, n, out); break;
    case ORC_VLCU16:  orc_rccdfudec16(in, n, out); break;
    case ORC_VLCU32:  orc_rccdfudec32(in, n, out); break;
    case ORC_VLCV16:  orc_rccdfvdec16(in, n, out); break;
    case ORC_VLCV32:  orc_rccdfvdec32(in, n, out); break;
    case ORC_VLCVZ16: orc_rccdfvzdec16(in, n, out); break;
    case ORC_VLCVZ32: orc_rccdfvzdec32(in, n, out); break;
    case ORC_VLAU16:  orc_anscdfudec16(in, n, out); break;
    case ORC_VLAUZ16: orc_anscdfuzdec16(in, n, out); break;
    case ORC_VLAV16:  orc_anscdfvdec16(in, n, out); break;
    case ORC_VLAVZ16: orc_anscdfvzdec16(in, n, out); break;
    case ORC_VLAV32:  orc_anscdfvdec32(in, n, out); break;
    case ORC_VLAVZ32: orc_anscdfvzdec32(in, n, out); break;
    case ORC_RCV8:    orc_rccdfdec8(in, n, out); break;
    case ORC_RCVI8:   orc_rccdfidec8(in, n, out); break;
    }
}
size_t orc_chunked_enc(int codec, const uint8_t *in, size_t n, size_t chunk,
                       const uint16_t *cdf, unsigned cdfnum,
                       uint8_t *payload, uint32_t *clen, uint64_t *poff)
{
    size_t nch = chunk ? (n + chunk - 1) / chunk : 0, off = 0;
    uint8_t *tmp = (uint8_t *)malloc(chunk + 64);
    if (!tmp) return 0;
    for (size_t c = 0; c < nch; c++) {
        size_t len = n - c * chunk < chunk ? n - c * chunk : chunk;
        size_t l = enc_one(codec, in + c * chunk, len, tmp, cdf, cdfnum);
        memcpy(payload + off, tmp, l);
        clen[c] = (uint32_t)l; poff[c] = off; off += l;
    }
    poff[nch] = off;
    free(tmp);
    return off;
}
size_t orc_chunked_dec(int codec, const uint8_t *payload, const uint32_t *clen, size_t n, size_t chunk,
                       const uint16_t *cdf, unsigned cdfnum, uint8_t *out)
{
    size_t nch = chunk ? (n + chunk - 1) / chunk : 0, off = 0;
    uint8_t *tmp = (uint8_t *)malloc(chunk + 64);
    if (!tmp) return 0;
    for (size_t c = 0; c < nch; c++) {
        size_t len = n - c * chunk < chunk ? n - c * chunk : chunk;
        if (clen[c] == len) memcpy(out + c * chunk, payload + off, len);      /* stored raw */
        else {
            memcpy(tmp, payload + off, clen[c]); memset(tmp + clen[c], 0, 16);  /* decoders over-read <= 8 B */
            dec_one(codec, tmp, len, out + c * chunk, cdf, cdfnum);
        }
        off += clen[c];
    }
    free(tmp);
    return n;
}
