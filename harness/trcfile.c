/* trcfile.c -- minimal file compressor on top of the drop-in headers ("makes the chunked format a
 * usable file compressor").  Plain C, links only against libturborc_hip.so.
 *
 *   trcfile c <id> <in> <out>     compress   (id: TurboRC -e numbers 1, 42, 44, 45, 46, 47, 56, 64, 65, 66)
 *   trcfile d <in> <out>          decompress
 *
 * File = "TRCF" | u8 id | u8 cdfnum-1 | u16 0 | u64 raw length | u64 stored length | [cdf: (cdfnum+1) x u16, static coders]
 *        | stored bytes (the library's TRC1 container, or the raw input when it does not compress: the reference's
 *        "returned length == input length means stored" convention, include/turborc.h:46-59).
 * The reference's own file mode (hd_t / hdb_t, turborc.c:666-733,1044-1167) writes whole-buffer streams that only its
 * serial decoders can read; this tool does not read or write that format. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../include/turborc.h"
#include "../include/anscdf.h"
#include "../include/trc_hip.h"

typedef size_t (*fn3)(unsigned char *, size_t, unsigned char *);
typedef size_t (*fn5)(unsigned char *, size_t, unsigned char *, cdf_t *, unsigned);
static size_t e65(unsigned char *i, size_t n, unsigned char *o, cdf_t *c, unsigned m) { (void)m; return anscdf4senc(i, n, o, c); }
static size_t d65(unsigned char *i, size_t n, unsigned char *o, cdf_t *c, unsigned m) { (void)m; return anscdf4sdec(i, n, o, c); }

static int pick(int id, fn3 *e3, fn3 *d3, fn5 *e5, fn5 *d5)
{
    *e3 = *d3 = 0; *e5 = *d5 = 0;
    switch (id) {
    case 1:  *e3 = rcsenc; *d3 = rcsdec; return 0;
    case 46: *e3 = rccdfenc; *d3 = rccdfdec; return 0;
    case 47: *e3 = rccdfienc; *d3 = rccdfidec; return 0;
    case 56: *e3 = anscdfenc; *d3 = anscdfdec; return 0;
    case 64: *e3 = anscdf1enc; *d3 = anscdf1dec; return 0;
    case 66: *e3 = ansbc; *d3 = ansbd; return 0;
    case 42: *e5 = rccdfsenc; *d5 = rccdfsbdec; return 0;
    case 44: *e5 = rccdfsmenc; *d5 = rccdfsmbdec; return 0;
    case 45: *e5 = rccdfs2enc; *d5 = rccdfsb2dec; return 0;
    case 65: *e5 = e65; *d5 = d65; return 0;
    }
    return -1;
}
static unsigned char *slurp(const char *path, size_t *n)
{
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); return 0; }
    fseek(f, 0, SEEK_END); *n = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
    unsigned char *p = malloc(*n + 1024);
    if (!p || fread(p, 1, *n, f) != *n) { perror("read"); fclose(f); free(p); return 0; }
    fclose(f);
    return p;
}

int main(int argc, char **argv)
{
    fn3 e3, d3; fn5 e5, d5;
    if (argc == 5 && !strcmp(argv[1], "c")) {
        const int id = atoi(argv[2]);
        size_t n;
        if (pick(id, &e3, &d3, &e5, &d5)) { fprintf(stderr, "unknown id %d\n", id); return 2; }
        unsigned char *in = slurp(argv[3], &n);
        if (!in) return 2;
        unsigned char *out = malloc(n + n / 3 + 1024);
        if (!out) { perror("malloc"); return 2; }
        cdf_t cdf[257];
        unsigned m = 0;
        size_t l = n;
        if (n && e5) {
            for (size_t i = 0; i < n; i++) if (in[i] > m) m = in[i];
            if (cdfini(in, n, cdf, m + 1) < 0) { e5 = 0; e3 = 0; }        /* distribution the 15-bit CDF cannot hold: store */
        }
        if (n && (e3 || e5)) {
            l = e3 ? e3(in, n, out) : e5(in, n, out, cdf, m + 1);
            if (!l) { fprintf(stderr, "encode failed: %s\n", trc_last_error()); return 1; }
        }
        FILE *f = fopen(argv[4], "wb");
        if (!f) { perror(argv[4]); return 2; }
        const uint8_t hdr[8] = { 'T', 'R', 'C', 'F', (uint8_t)id, (uint8_t)m, 0, 0 };
        const uint64_t raw = n, stored = l;
        fwrite(hdr, 1, 8, f); fwrite(&raw, 8, 1, f); fwrite(&stored, 8, 1, f);
        if (e5) fwrite(cdf, sizeof(cdf_t), m + 2, f);
        fwrite(l == n ? in : out, 1, l, f);
        fclose(f);
        printf("%zu -> %zu bytes (%.2f%%)%s\n", n, l, n ? 100.0 * l / n : 0.0, l == n ? "  stored" : "");
        return 0;
    }
    if (argc == 4 && !strcmp(argv[1], "d")) {
        size_t fl;
        unsigned char *fb = slurp(argv[2], &fl);
        if (!fb) return 2;
        if (fl < 24 || memcmp(fb, "TRCF", 4)) { fprintf(stderr, "not a TRCF file\n"); return 2; }
        const int id = fb[4];
        const unsigned m = fb[5];
        uint64_t raw, stored;
        memcpy(&raw, fb + 8, 8); memcpy(&stored, fb + 16, 8);
        if (pick(id, &e3, &d3, &e5, &d5)) { fprintf(stderr, "unknown id %d\n", id); return 2; }
        size_t pos = 24;
        cdf_t cdf[257];
        if (d5) {
            if (pos + (m + 2) * sizeof(cdf_t) > fl) { fprintf(stderr, "truncated file\n"); return 2; }
            memcpy(cdf, fb + pos, (m + 2) * sizeof(cdf_t)); pos += (m + 2) * sizeof(cdf_t);
        }
        if (stored > fl - pos || stored != fl - pos || stored > raw) { fprintf(stderr, "truncated or padded file\n"); return 2; }
        /* untrusted input: the decoders take no input length, so the container is validated against what was read */
        if (stored != raw && trc_container_check(fb + pos, (size_t)stored, 0, (size_t)raw)) { fprintf(stderr, "corrupt file: %s\n", trc_last_error()); return 2; }
        unsigned char *out = malloc(raw + 1024);
        if (!out) { perror("malloc"); return 2; }
        if (stored == raw) memcpy(out, fb + pos, raw);                    /* stored: the caller copies (CCPY) */
        else if ((d3 ? d3(fb + pos, raw, out) : d5(fb + pos, raw, out, cdf, m + 1)) != raw) { fprintf(stderr, "decode failed: %s\n", trc_last_error()); return 1; }
        FILE *f = fopen(argv[3], "wb");
        if (!f) { perror(argv[3]); return 2; }
        fwrite(out, 1, raw, f);
        fclose(f);
        return 0;
    }
    fprintf(stderr, "usage: trcfile c <id> <in> <out> | trcfile d <in> <out>\n");
    return 2;
}
