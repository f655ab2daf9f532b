/*
 * trcbench.c -- a TurboRC-style bench harness in plain C, linked against libturborc_hip.so through
 * include/turborc.h + include/anscdf.h ONLY (the drop-in boundary).  It re-creates what the
 * reference's bench() does for the hot-path ids (turborc.c:420-579; timing policy time_.h:174-213):
 *   untimed cdfini for the static coders (turborc.c:429-433); `cpy` poisoned with ~in (:427);
 *   timed encode; timed decode -- or memcpy when the encoder returned inlen (CCPY, :434);
 *   memcheck (:287-295); min over runs; MB = 10^6.
 * The timed calls take HOST pointers, so these numbers include PCIe both ways; the
 * device-resident numbers come from bench.py.
 *
 *   trcbench [-e id[,id..]] [-I runs] [-c chunk] (file | --zipf N | --text N | --uniform N | --nibble N | --int16 N | --int32 N)
 * ids: 1 rcs | 42 cdfsb | 43 cdfsv | 45 cdfs2 | 46 cdf | 47 cdfi | 56 ans | 57 ans(s) | 58 ans(x) | 65 ans4s | 79 memcpy
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "../include/turborc.h"
#include "../include/anscdf.h"

int trc_set_chunk(unsigned chunk);           /* from include/trc_hip.h */
int trc_host_pin(void *p, size_t len);       /* page-lock a buffer: host-pointer calls then DMA straight from / to it */
const char *trc_last_error(void);

static int g_elem = 0;      /* element bytes of integer input (2 / 4), 0 = bytes */
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static unsigned long long sm64(unsigned long long *s)
{
    unsigned long long z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
/* kind 0: Zipf(1.1) over 256 symbols, 1: text-like (Zipf(1.6) over 96 printable bytes), 2: uniform,
 * 3: nibble values (Zipf(1.1) over 16 symbols, the `turborc -n` coders) */
static void gen(unsigned char *p, size_t n, int kind)
{
    unsigned long long s = 12345;
    if (kind == 4) {                                   /* slow random walk of 16/32-bit integers (N bytes) */
        unsigned v = 1u << (8 * g_elem - 2);
        for (size_t k = 0; k + g_elem <= n; k += g_elem) {
            v += (unsigned)(sm64(&s) % 61) - 30;
            memcpy(p + k, &v, g_elem);
        }
        return;
    }
    double cum[256], tot = 0;
    int nsym = kind == 1 ? 96 : kind == 3 ? 16 : 256;
    for (int i = 0; i < nsym; i++) { tot += kind == 2 ? 1.0 : 1.0 / pow(i + 1.0, kind == 1 ? 1.6 : 1.1); cum[i] = tot; }
    for (size_t k = 0; k < n; k++) {
        double u = (double)(sm64(&s) >> 11) * (1.0 / 9007199254740992.0) * tot;
        int lo = 0, hi = nsym - 1;
        while (lo < hi) { int mid = (lo + hi) >> 1; if (cum[mid] <= u) lo = mid + 1; else hi = mid; }
        p[k] = (unsigned char)(kind == 1 ? 32 + lo : lo);
    }
}
static size_t memcheck(const unsigned char *a, const unsigned char *b, size_t n)
{
    for (size_t i = 0; i < n; i++) if (a[i] != b[i]) { printf("ERROR in[%zu]=%x dec[%zu]=%x\n", i, a[i], i, b[i]); return i + 1; }
    return 0;
}

typedef size_t (*enc3)(unsigned char *, size_t, unsigned char *);
typedef size_t (*enc4)(unsigned char *, size_t, unsigned char *, cdf_t *);
typedef size_t (*enc5)(unsigned char *, size_t, unsigned char *, cdf_t *, unsigned);

static int bench(unsigned char *in, size_t n, unsigned char *out, unsigned char *cpy, int id, int runs)
{
    cdf_t cdf[257];
    unsigned m = 0;
    const char *name = "?";
    enc3 e3 = 0, d3 = 0; enc4 e4 = 0, d4 = 0; enc5 e5 = 0, d5 = 0;
    for (size_t i = 0; i < n; i++) if (in[i] > m) m = in[i];
    /* nibble-valued input (`turborc -n`): ids 46/47/56-58 run the one-table nibble coders, as the reference
     * harness does under its m<16 gate (turborc.c:499-501,514-520) */
    if (m < 16) switch (id) {
    case 46: name = "cdf4 nibble adaptive (rccdf4enc/rccdf4dec)"; e3 = rccdf4enc; d3 = rccdf4dec; break;
    case 47: name = "cdf4i nibble adaptive interleaved (rccdf4ienc/rccdf4idec)"; e3 = rccdf4ienc; d3 = rccdf4idec; break;
    case 56: name = "ans auto nibble (anscdf4enc/anscdf4dec)"; e3 = anscdf4enc; d3 = anscdf4dec; break;
    case 57: name = "ans s nibble (anscdf4encs/anscdf4decs)"; e3 = anscdf4encs; d3 = anscdf4decs; break;
    case 58: name = "ans x nibble (anscdf4encx/anscdf4decx)"; e3 = anscdf4encx; d3 = anscdf4decx; break;
    }
    /* 16/32-bit integer input (`turborc -Os2 / -Os4` style, here: --int16 / --int32): ids 50/52/53 are the Turbo-VLC coders */
    if (!e3 && g_elem) switch (id) {
    case 50: name = g_elem == 2 ? "cdf-16 Turbo vlc6 (rccdfuenc16/rccdfudec16)" : "cdf-32 Turbo vlc6 (rccdfuenc32/rccdfudec32)";
             e3 = g_elem == 2 ? rccdfuenc16 : rccdfuenc32; d3 = g_elem == 2 ? rccdfudec16 : rccdfudec32; break;
    case 52: name = g_elem == 2 ? "cdf-16 Turbo vlc7 (rccdfvenc16/rccdfvdec16)" : "cdf-32 Turbo vlc7 (rccdfvenc32/rccdfvdec32)";
             e3 = g_elem == 2 ? rccdfvenc16 : rccdfvenc32; d3 = g_elem == 2 ? rccdfvdec16 : rccdfvdec32; break;
    case 53: name = g_elem == 2 ? "cdf-16 Turbo vlc7 zigzag (rccdfvzenc16/rccdfvzdec16)" : "cdf-32 Turbo vlc7 zigzag (rccdfvzenc32/rccdfvzdec32)";
             e3 = g_elem == 2 ? rccdfvzenc16 : rccdfvzenc32; d3 = g_elem == 2 ? rccdfvzdec16 : rccdfvzdec32; break;
    /* ... over rANS (the reference has the 6-bit forms for 16-bit input only: turborc.c:526-529) */
    case 60: if (g_elem == 2) { name = "anscdf-16 Turbo vlc6 (anscdfuenc16/anscdfudec16)"; e3 = anscdfuenc16; d3 = anscdfudec16; } break;
    case 61: if (g_elem == 2) { name = "anscdf-16 Turbo vlc6 zigzag (anscdfuzenc16/anscdfuzdec16)"; e3 = anscdfuzenc16; d3 = anscdfuzdec16; } break;
    case 62: name = g_elem == 2 ? "anscdf-16 Turbo vlc7 (anscdfvenc16/anscdfvdec16)" : "anscdf-32 Turbo vlc7 (anscdfvenc32/anscdfvdec32)";
             e3 = g_elem == 2 ? anscdfvenc16 : anscdfvenc32; d3 = g_elem == 2 ? anscdfvdec16 : anscdfvdec32; break;
    case 63: name = g_elem == 2 ? "anscdf-16 Turbo vlc7 zigzag (anscdfvzenc16/anscdfvzdec16)" : "anscdf-32 Turbo vlc7 zigzag (anscdfvzenc32/anscdfvzdec32)";
             e3 = g_elem == 2 ? anscdfvzenc16 : anscdfvzenc32; d3 = g_elem == 2 ? anscdfvzdec16 : anscdfvzdec32; break;
    }
    if (!e3) switch (id) {
    case 1:  name = "rc o0 (rcsenc/rcsdec)"; e3 = rcsenc; d3 = rcsdec; break;
    case 42: name = "cdfsb (rccdfsenc/rccdfsbdec)"; e5 = rccdfsenc; d5 = rccdfsbdec; break;
    case 43: name = "cdfsv (rccdfsenc/rccdfsvbdec)"; e5 = rccdfsenc; d5 = rccdfsvbdec; break;
    case 44: name = "cdfsm 32-bit range (rccdfsmenc/rccdfsmbdec)"; e5 = rccdfsmenc; d5 = rccdfsmbdec; break;
    case 45: name = "cdfsb interleaved (rccdfs2enc/rccdfsb2dec)"; e5 = rccdfs2enc; d5 = rccdfsb2dec; break;
    case 46: name = "cdf byte adaptive (rccdfenc/rccdfdec)"; e3 = rccdfenc; d3 = rccdfdec; break;
    case 47: name = "cdfi byte adaptive interleaved (rccdfienc/rccdfidec)"; e3 = rccdfienc; d3 = rccdfidec; break;
    case 48: name = "cdf-8 variable-length (rccdfenc8/rccdfdec8)"; e3 = rccdfenc8; d3 = rccdfdec8; break;
    case 49: name = "cdfi-8 variable-length interleaved (rccdfienc8/rccdfidec8)"; e3 = rccdfienc8; d3 = rccdfidec8; break;
    case 56: name = "ans auto (anscdfenc/anscdfdec)"; e3 = anscdfenc; d3 = anscdfdec; break;
    case 57: name = "ans s (anscdfencs/anscdfdecs)"; e3 = anscdfencs; d3 = anscdfdecs; break;
    case 58: name = "ans x (anscdfencx/anscdfdecx)"; e3 = anscdfencx; d3 = anscdfdecx; break;
    case 64: name = "ans o1 (anscdf1enc/anscdf1dec)"; e3 = anscdf1enc; d3 = anscdf1dec; break;
    case 65: name = "ans static (anscdf4senc/anscdf4sdec)"; e4 = anscdf4senc; d4 = anscdf4sdec; break;
    case 66: name = "ansb bitwise ans (ansbc/ansbd)"; e3 = ansbc; d3 = ansbd; break;
    case 79: name = "memcpy"; break;
    default: return 0;
    }
    if (e4 || e5) {                                    /* untimed, as in the reference harness */
        if (cdfini(in, n, cdf, m + 1) < 0) { printf("%2d: cdfini failed: %s\n", id, trc_last_error()); return 1; }
    }
    for (size_t i = 0; i < n; i++) cpy[i] = (unsigned char)~in[i];
    size_t l = 0;
    double te = 1e30, td = 1e30;
    for (int r = 0; r < runs; r++) {
        double t0 = now();
        if (e3) l = e3(in, n, out); else if (e4) l = e4(in, n, out, cdf); else if (e5) l = e5(in, n, out, cdf, m + 1);
        else { memcpy(out, in, n); l = n; }
        double t1 = now();
        if (t1 - t0 < te) te = t1 - t0;
        if (!l && n) { printf("%2d: encode failed: %s\n", id, trc_last_error()); return 1; }
    }
    for (int r = 0; r < runs; r++) {
        double t0 = now();
        size_t k = n;
        if (l == n) memcpy(cpy, out, n);               /* stored raw: the caller copies (CCPY) */
        else if (d3) k = d3(out, n, cpy); else if (d4) k = d4(out, n, cpy, cdf); else if (d5) k = d5(out, n, cpy, cdf, m + 1);
        double t1 = now();
        if (t1 - t0 < td) td = t1 - t0;
        if (k != n) { printf("%2d: decode failed: %s\n", id, trc_last_error()); return 1; }
    }
    int bad = memcheck(in, cpy, n) != 0;
    printf("%12zu %6.2f%% %10.2f %10.2f   %2d:%s%s\n", l, 100.0 * l / (double)n, n / te / 1e6, n / td / 1e6, id, name, bad ? "  **MISMATCH**" : "");
    return bad;
}

int main(int argc, char **argv)
{
    const char *ids = "1,42,44,45,46,47,56,65,79", *file = 0;
    int runs = 3, kind = -1, pin = 0;
    size_t n = 0;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "-e") && i + 1 < argc) ids = argv[++i];
        else if (!strcmp(argv[i], "-I") && i + 1 < argc) runs = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-c") && i + 1 < argc) { if (trc_set_chunk((unsigned)atoi(argv[++i]))) return 2; }
        else if (!strcmp(argv[i], "--pin")) pin = 1;   /* page-lock in / out / cpy once (what a caller that reuses its buffers would do) */
        else if (!strcmp(argv[i], "--zipf") && i + 1 < argc) { kind = 0; n = strtoull(argv[++i], 0, 10); }
        else if (!strcmp(argv[i], "--text") && i + 1 < argc) { kind = 1; n = strtoull(argv[++i], 0, 10); }
        else if (!strcmp(argv[i], "--uniform") && i + 1 < argc) { kind = 2; n = strtoull(argv[++i], 0, 10); }
        else if (!strcmp(argv[i], "--nibble") && i + 1 < argc) { kind = 3; n = strtoull(argv[++i], 0, 10); }
        else if (!strcmp(argv[i], "--int16") && i + 1 < argc) { kind = 4; g_elem = 2; n = strtoull(argv[++i], 0, 10) & ~(size_t)1; }
        else if (!strcmp(argv[i], "--int32") && i + 1 < argc) { kind = 4; g_elem = 4; n = strtoull(argv[++i], 0, 10) & ~(size_t)3; }
        else file = argv[i];
    }
    if (file) {
        FILE *f = fopen(file, "rb");
        if (!f) { perror(file); return 2; }
        fseek(f, 0, SEEK_END); n = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
        unsigned char *tmp = malloc(n + 1);
        if (fread(tmp, 1, n, f) != n) { perror("read"); return 2; }
        fclose(f);
        unsigned char *in = malloc(n * 4 / 3 + 1024); memcpy(in, tmp, n); free(tmp);
        unsigned char *out = malloc(n * 4 / 3 + 1024), *cpy = malloc(n * 4 / 3 + 1024);
        if (pin && (trc_host_pin(in, n * 4 / 3 + 1024) || trc_host_pin(out, n * 4 / 3 + 1024) || trc_host_pin(cpy, n * 4 / 3 + 1024))) { fprintf(stderr, "--pin: %s\n", trc_last_error()); return 2; }
        printf("file %s: %zu bytes\n      C Size  ratio%%    E MB/s     D MB/s   Name (host pointers: PCIe included)\n", file, n);
        int bad = 0; char *s = strdup(ids), *sv = 0;       /* strtok_r: the HIP runtime uses strtok itself while initialising */
        for (char *t = strtok_r(s, ",", &sv); t; t = strtok_r(0, ",", &sv)) bad |= bench(in, n, out, cpy, atoi(t), runs);
        return bad;
    }
    if (kind < 0 || !n) { fprintf(stderr, "usage: trcbench [-e ids] [-I runs] [-c chunk] [--pin] (file | --zipf N | --text N | --uniform N | --nibble N | --int16 N | --int32 N)\n"); return 2; }
    unsigned char *in = malloc(n * 4 / 3 + 1024), *out = malloc(n * 4 / 3 + 1024), *cpy = malloc(n * 4 / 3 + 1024);
    gen(in, n, kind);
    if (pin && (trc_host_pin(in, n * 4 / 3 + 1024) || trc_host_pin(out, n * 4 / 3 + 1024) || trc_host_pin(cpy, n * 4 / 3 + 1024))) { fprintf(stderr, "--pin: %s\n", trc_last_error()); return 2; }
    printf("synthetic kind %d: %zu bytes%s\n      C Size  ratio%%    E MB/s     D MB/s   Name (host pointers: PCIe included)\n", kind, n, pin ? " (buffers page-locked)" : "");
    int bad = 0; char *s = strdup(ids), *sv = 0;
    for (char *t = strtok_r(s, ",", &sv); t; t = strtok_r(0, ",", &sv)) bad |= bench(in, n, out, cpy, atoi(t), runs);
    return bad;
}
