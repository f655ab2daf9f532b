/* trcgather.c -- plain-C multi-GPU driver: one process per GPU, chunk ranges sharded over the ranks, static rANS, results
 * gathered over RCCL through the library's own entry points (trc_hist_allreduce_dev, trc_exchange_dev) -- what a
 * TurboRC-style C caller needs to run `--gpus N` without any Python.
 *
 *   trcgather --gpus N [--size BYTES] [--chunk BYTES] [--steps K] [--batches B] [--watchdog SECONDS] [--quiet]
 *
 * The parent forks N ranks; rank r takes HIP device r % (visible devices) -- several ranks on one device is how the
 * test suite runs the exchange arithmetic with world > 1 on a one-GPU box (over tests/fake_rccl.c, selected with
 * TRC_RCCL_LIB).  Rank 0 creates the RCCL unique id and hands it to the others through a file.  There are B batches
 * (default 1): batch j is its own synthetic input (seed j), every rank codes its contiguous range of whole chunks of
 * every batch with the CDF of the WHOLE batch (histogram all-reduce), and ONE trc_exchange_dev call gathers batch j
 * onto rank j % N (B = 1: the plain gather onto rank 0; B = N: every directed link carries one payload at once).
 * Each root decodes the container it assembled in one piece, compares it with the batch's input and prints an
 * FNV-1a-64 of directory + payload: the hashes do not depend on N (the container is the single-GPU container).
 * Prints encode+gather MB/s (MB = 10^6, input-referred over all batches, best of K).
 *
 * Never hangs silently: every phase is announced on stderr (unbuffered, with the time since start) unless --quiet,
 * and a per-phase watchdog (alarm) names the phase it fired in and exits with status 4.  RCCL is loaded with dlopen
 * inside a phase of its own (librccl.so is 570 MB: on a box whose image is still cold that load alone takes a while). */
#define _GNU_SOURCE 1
#define __HIP_PLATFORM_AMD__ 1
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>
#include "../include/trc_hip.h"

static int g_rank = -1, g_quiet = 0, g_watchdog = 60;
static const char *volatile g_phase = "start";
static double g_t0;
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static size_t up(size_t x) { return (x + 255) & ~(size_t)255; }

static void on_alarm(int sig)
{
    (void)sig;
    char msg[256];
    int k = snprintf(msg, sizeof msg, "trcgather: rank %d: WATCHDOG: no progress for %d s in phase '%s' -- giving up\n", g_rank, g_watchdog, g_phase);
    if (k > 0) { ssize_t w = write(2, msg, (size_t)k); (void)w; }
    _exit(4);
}
/* announce a phase and re-arm the watchdog: `budget` multiplies the per-phase limit (library loads on a cold box) */
static void phase(const char *name, int budget)
{
    g_phase = name;
    if (g_watchdog > 0) alarm((unsigned)(g_watchdog * budget));
    if (!g_quiet) { fprintf(stderr, "trcgather: rank %d: +%.2fs %s\n", g_rank, now() - g_t0, name); fflush(stderr); }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "rank %d: [%s] %s -> %s\n", g_rank, g_phase, #x, hipGetErrorString(e_)); exit(3); } } while (0)
#define NK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "rank %d: [%s] %s -> %s\n", g_rank, g_phase, #x, R.GetErrorString(r_)); exit(3); } } while (0)
#define TK(x) do { if ((x) != 0) { fprintf(stderr, "rank %d: [%s] %s -> %s\n", g_rank, g_phase, #x, trc_last_error()); exit(3); } } while (0)

/* the four RCCL calls the driver itself makes (the transfers are the library's: trc_rccl.hip) */
static struct {
    ncclResult_t (*GetUniqueId)(ncclUniqueId *);
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    const char *(*GetErrorString)(ncclResult_t);
} R;
static void load_rccl(void)
{
    const char *path = getenv("TRC_RCCL_LIB");
    void *h = dlopen(path && *path ? path : "librccl.so.1", RTLD_NOW | RTLD_GLOBAL);   /* GLOBAL: the library finds the same copy */
    if (!h) { fprintf(stderr, "rank %d: cannot load RCCL: %s\n", g_rank, dlerror()); exit(3); }
    *(void **)&R.GetUniqueId = dlsym(h, "ncclGetUniqueId"); *(void **)&R.CommInitRank = dlsym(h, "ncclCommInitRank");
    *(void **)&R.CommDestroy = dlsym(h, "ncclCommDestroy"); *(void **)&R.GetErrorString = dlsym(h, "ncclGetErrorString");
    if (!R.GetUniqueId || !R.CommInitRank || !R.CommDestroy || !R.GetErrorString) { fprintf(stderr, "rank %d: RCCL symbols missing\n", g_rank); exit(3); }
}

/* batch j's input: skewed bytes (4th power of a uniform value) from a splitmix64 stream seeded with j */
static void gen_input(unsigned char *h, size_t n, int j)
{
    uint64_t z = 0x9E3779B97F4A7C15ull * (uint64_t)(j + 1);
    for (size_t i = 0; i < n; i++) {
        z += 0x9E3779B97F4A7C15ull; uint64_t x = z; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; x ^= x >> 31;
        const unsigned u = (unsigned)(x >> 40) & 0xffff, v = (unsigned)(((uint64_t)u * u) >> 16);
        h[i] = (unsigned char)((((uint64_t)v * v) >> 24) + (unsigned)j);
    }
}
static uint64_t fnv1a(uint64_t hsh, const unsigned char *p, size_t n) { for (size_t i = 0; i < n; i++) { hsh ^= p[i]; hsh *= 0x100000001B3ull; } return hsh; }

typedef struct { unsigned char *h, *d_in, *d_payload, *d_all_payload, *d_out; uint32_t *d_clen, *d_all_clen; uint64_t *d_total, *d_hist; uint16_t *d_cdf; } batch_bufs;

static int run_rank(int rank, int world, size_t n, uint32_t chunk, int steps, int nb, const char *idfile)
{
    g_rank = rank;
    signal(SIGALRM, on_alarm);
    phase("hip-init", 2);
    int ndev = 0; CK(hipGetDeviceCount(&ndev));
    if (ndev < 1) { fprintf(stderr, "rank %d: no HIP device\n", rank); return 3; }
    CK(hipSetDevice(rank % ndev));
    CK(hipFree(0));
    phase("load-rccl", 4);
    load_rccl();
    phase("unique-id", 1);
    ncclUniqueId id;
    if (rank == 0) {
        NK(R.GetUniqueId(&id));
        char tmp[512]; snprintf(tmp, sizeof tmp, "%s.tmp", idfile);
        FILE *f = fopen(tmp, "wb");
        if (!f || fwrite(&id, sizeof id, 1, f) != 1) { fprintf(stderr, "rank 0: cannot write %s\n", tmp); return 3; }
        fclose(f); rename(tmp, idfile);
    } else {
        FILE *f = 0;
        for (int i = 0; i < 100 * g_watchdog * 4 && !(f = fopen(idfile, "rb")); i++) { usleep(10000); if (i % 100 == 99 && g_watchdog > 0) alarm((unsigned)g_watchdog); }
        if (!f || fread(&id, sizeof id, 1, f) != 1) { fprintf(stderr, "rank %d: no unique id\n", rank); return 3; }
        fclose(f);
    }
    phase("comm-init", 4);
    ncclComm_t comm;
    NK(R.CommInitRank(&comm, world, id, rank));
    phase("buffers+input", 2);
    hipStream_t s; CK(hipStreamCreate(&s));

    const size_t nch = (n + chunk - 1) / chunk, per = nch / world, rem = nch % world;
    const size_t c0 = rank * per + (rank < (int)rem ? rank : rem), mych = per + (rank < (int)rem ? 1 : 0);
    const size_t off = c0 * chunk, mylen = mych ? ((c0 + mych) * (size_t)chunk < n ? mych * (size_t)chunk : n - off) : 0;
    const size_t wb = trc_work_bytes(TRC_ANS4S, mylen ? mylen : chunk, chunk), wball = trc_work_bytes(TRC_ANS4S, n, chunk);
    unsigned char *d_work, *d_meta; int32_t *d_status;
    CK(hipMalloc((void **)&d_work, (wball > wb ? wball : wb) + 256));
    CK(hipMalloc((void **)&d_status, 256)); CK(hipMalloc((void **)&d_meta, 16 * (size_t)nb * (world + 1) + 256));
    batch_bufs *bb = (batch_bufs *)calloc(nb, sizeof *bb);
    trc_batch *b = (trc_batch *)calloc(nb, sizeof *b);
    for (int j = 0; j < nb; j++) {
        batch_bufs *q = &bb[j];
        const int root = j % world;
        q->h = (unsigned char *)malloc(n + 512);
        gen_input(q->h, n, j);                                     /* the whole input on every rank's host (for the root's check) */
        CK(hipMalloc((void **)&q->d_in, up(mylen + 512))); CK(hipMalloc((void **)&q->d_payload, up(mylen + 512)));
        CK(hipMalloc((void **)&q->d_clen, up(4 * mych + 256))); CK(hipMalloc((void **)&q->d_total, 256)); CK(hipMalloc((void **)&q->d_hist, 4096));
        CK(hipMalloc((void **)&q->d_cdf, 1024));
        if (rank == root) { CK(hipMalloc((void **)&q->d_all_payload, up(n + 512))); CK(hipMalloc((void **)&q->d_all_clen, up(4 * nch + 256))); CK(hipMalloc((void **)&q->d_out, up(n + 512))); }
        CK(hipMemsetAsync(q->d_in, 0, up(mylen + 512), s));
        CK(hipMemcpyAsync(q->d_in, q->h + off, mylen, hipMemcpyHostToDevice, s));
        CK(hipMemsetAsync(q->d_total, 0, 8, s));
        b[j].d_clen = q->d_clen; b[j].nchunks = mych; b[j].d_payload = q->d_payload; b[j].d_total = q->d_total;
        b[j].d_clen_all = q->d_all_clen; b[j].d_payload_all = q->d_all_payload;
    }
    CK(hipStreamSynchronize(s));

    phase("histogram+cdf", 2);                                      /* one CDF per batch for the whole job */
    for (int j = 0; j < nb; j++) {
        TK(trc_hist_dev(bb[j].d_in, mylen, bb[j].d_hist, s));
        TK(trc_hist_allreduce_dev(comm, bb[j].d_hist, s));
        TK(trc_cdf_from_hist_dev(bb[j].d_hist, n, bb[j].d_cdf, 256, d_status, s));
        int32_t st = 0; CK(hipMemcpyAsync(&st, d_status, 4, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
        if (st < 0) { fprintf(stderr, "rank %d: cdfini failed (batch %d)\n", rank, j); return 3; }
    }

    uint64_t *sizes = (uint64_t *)malloc(16 * (size_t)world * nb);
    double best = 1e30;
    /* steps + 1 passes: the first is untimed (connections, allocations, idle clocks); with one rank a few more untimed
     * passes bring the GPU up to its clocks, bounded by wall time only (ranks must issue the same number of exchanges,
     * so with peers the count is fixed) */
    phase("first-pass", 2);
    const double w0 = now();
    for (int k = 0, warm = 1; k < steps + 1; ) {
        CK(hipStreamSynchronize(s));
        const double t0 = now();
        for (int j = 0; j < nb; j++)
            if (mylen) TK(trc_encode_dev(TRC_ANS4S, bb[j].d_in, mylen, chunk, bb[j].d_cdf, 256, bb[j].d_clen, bb[j].d_payload, bb[j].d_total, d_work, wb, s));
        TK(trc_exchange_dev(comm, nb, b, sizes, d_meta, s));
        CK(hipStreamSynchronize(s));
        const double t1 = now();
        if (warm) {
            if (world == 1 && t1 - w0 < 0.3) continue;              /* still warming (at most 0.3 s of wall time) */
            warm = 0; k = 1; phase("timed-passes", 2); continue;
        }
        if (t1 - t0 < best) best = t1 - t0;
        k++;
    }
    phase("verify", 2);
    int rc = 0;
    for (int j = 0; j < nb; j++) {
        if (rank != j % world) continue;
        batch_bufs *q = &bb[j];
        uint64_t total = 0, chunks = 0;
        for (int r = 0; r < world; r++) { total += sizes[((size_t)r * nb + j) * 2]; chunks += sizes[((size_t)r * nb + j) * 2 + 1]; }
        if (chunks != nch) { printf("FAILED: batch %d: gathered %llu chunks, expected %zu\n", j, (unsigned long long)chunks, nch); rc = 1; continue; }
        if (total > n) { printf("FAILED: batch %d: gathered %llu payload bytes for %zu input bytes\n", j, (unsigned long long)total, n); rc = 1; continue; }
        TK(trc_decode_dev(TRC_ANS4S, q->d_all_clen, q->d_all_payload, n, chunk, q->d_cdf, 256, q->d_out, d_work, wball, s));
        unsigned char *back = (unsigned char *)malloc(n + 4 * nch + total);
        CK(hipMemcpyAsync(back, q->d_out, n, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
        if (memcmp(back, q->h, n)) { printf("FAILED: batch %d: the gathered container does not decode to the input\n", j); rc = 1; }
        CK(hipMemcpy(back, q->d_all_clen, 4 * nch, hipMemcpyDeviceToHost)); CK(hipMemcpy(back + 4 * nch, q->d_all_payload, total, hipMemcpyDeviceToHost));
        const uint64_t hsh = fnv1a(0xCBF29CE484222325ull, back, 4 * nch + total);
        free(back);
        printf("batch %d on rank %d: %zu bytes -> %llu (%.2f%%) container %016llx%s\n", j, rank, n, (unsigned long long)(32 + 4 * nch + total),
               100.0 * (32 + 4 * nch + total) / n, (unsigned long long)hsh, rc ? "" : "  [container verified]");
    }
    if (rank == 0) printf("%d rank(s), %d batch(es): encode + gather %.1f MB/s\n", world, nb, (double)n * nb / best / 1e6);
    fflush(stdout);
    phase("comm-destroy", 1);
    R.CommDestroy(comm);
    phase("done", 1);
    alarm(0);
    if (rank == 0) unlink(idfile);
    return rc;
}

int main(int argc, char **argv)
{
    int world = 1, steps = 3, nb = 1; size_t n = 100u * 1000 * 1000; uint32_t chunk = 512;
    g_t0 = now();
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--gpus") && i + 1 < argc) world = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--size") && i + 1 < argc) n = (size_t)strtoull(argv[++i], 0, 10);
        else if (!strcmp(argv[i], "--chunk") && i + 1 < argc) chunk = (uint32_t)atoi(argv[++i]);
        else if (!strcmp(argv[i], "--steps") && i + 1 < argc) steps = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--batches") && i + 1 < argc) nb = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--watchdog") && i + 1 < argc) g_watchdog = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--quiet")) g_quiet = 1;
        else { fprintf(stderr, "usage: trcgather --gpus N [--size BYTES] [--chunk BYTES] [--steps K] [--batches B] [--watchdog SECONDS] [--quiet]\n"); return 2; }
    }
    if (world < 1 || world > 64 || !n || nb < 1 || nb > TRC_EXCHANGE_MAX_BATCH || steps < 1) return 2;
    setvbuf(stderr, 0, _IONBF, 0);
    char idfile[256]; snprintf(idfile, sizeof idfile, "/tmp/trcgather_%d.id", (int)getpid());
    if (world == 1) return run_rank(0, 1, n, chunk, steps, nb, idfile);
    pid_t pids[64];
    for (int r = 0; r < world; r++) {
        pids[r] = fork();                                       /* fork BEFORE any HIP call: each rank initialises its own runtime */
        if (pids[r] == 0) { int rc = run_rank(r, world, n, chunk, steps, nb, idfile); fflush(stdout); _exit(rc); }
    }
    int bad = 0;
    for (int r = 0; r < world; r++) {
        int st = 0; waitpid(pids[r], &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st)) { bad = WIFEXITED(st) ? WEXITSTATUS(st) : 5; fprintf(stderr, "trcgather: rank %d ended with status %d\n", r, bad); }
    }
    unlink(idfile);
    return bad;
}
