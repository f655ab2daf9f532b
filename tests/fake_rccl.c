/* fake_rccl.c -- TEST INFRASTRUCTURE, not product: a stand-in for the dozen RCCL entry points libturborc_hip.so and
 * harness/trcgather.c resolve at run time, so that the multi-rank exchange (trc_exchange_dev, trc_hist_allreduce_dev:
 * offsets, rotating roots, grouped send/receive order) executes with world > 1 on a box with ONE GPU.  Selected with
 * TRC_RCCL_LIB=tests/libfake_rccl.so; the product never loads it otherwise.
 *
 * Ranks are processes (several may share a device).  They meet in a file under /tmp named inside the ncclUniqueId:
 * a control block plus one outbox slot per rank, mapped shared.  Data moves device -> outbox -> device with plain
 * hipMemcpy after a hipStreamSynchronize of the caller's stream, so stream order is kept trivially.
 *
 * Stricter than RCCL where that finds bugs: a receive must meet a send of exactly the same byte count from that peer
 * (in posting order per directed pair, as NCCL matches them), every send of a group must be consumed inside the same
 * group, and every wait has a time limit -- a schedule that would hang or corrupt on the real library FAILS here.
 * Restrictions: ncclGroupEnd is collective over the communicator (every rank calls it, also with nothing posted --
 * which is how trc_exchange_dev uses it); collectives are not allowed inside a group; all-reduce is u64 sum only. */
#define _GNU_SOURCE 1
#define __HIP_PLATFORM_AMD__ 1
#include <errno.h>
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#define FK_MAXW 16
#define FK_MAXOPS 1024
#define FK_MAGIC 0x46524343u

typedef struct { int32_t dst; uint32_t pad; uint64_t bytes, off; } fk_desc;
typedef struct {
    _Atomic uint32_t magic;
    uint32_t world;
    uint64_t slot_bytes, data_off;
    _Atomic uint32_t arrived, generation;
    _Atomic int32_t error;
    struct { uint32_t nsend; fk_desc send[FK_MAXOPS]; } r[FK_MAXW];
} fk_ctl;

struct ncclComm { fk_ctl *ctl; unsigned char *base; size_t map_bytes; int rank, world; char path[128]; };

typedef struct { int is_send; void *buf; size_t bytes; int peer; struct ncclComm *comm; hipStream_t stream; } fk_op;
static __thread int g_depth;
static __thread int g_nops;
static __thread fk_op g_ops[FK_MAXOPS];

static struct ncclComm *g_last_comm;      /* the communicator an empty group's barriers run on (one communicator per test process) */

static double fk_now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static double fk_limit(void) { const char *e = getenv("TRC_FAKE_RCCL_TIMEOUT"); return e ? atof(e) : 60.0; }
static int fk_fail(struct ncclComm *c, const char *what)
{
    fprintf(stderr, "fake_rccl: rank %d: %s\n", c ? c->rank : -1, what);
    if (c && c->ctl) atomic_store(&c->ctl->error, 1);
    return 1;
}
static size_t fk_size(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}
/* sense-reversing barrier over the communicator, with a time limit */
static ncclResult_t fk_barrier(struct ncclComm *c)
{
    fk_ctl *k = c->ctl;
    if (atomic_load(&k->error)) return ncclSystemError;
    const uint32_t gen = atomic_load(&k->generation);
    if (atomic_fetch_add(&k->arrived, 1) + 1 == (uint32_t)c->world) { atomic_store(&k->arrived, 0); atomic_fetch_add(&k->generation, 1); return ncclSuccess; }
    const double t0 = fk_now(), lim = fk_limit();
    for (unsigned spins = 0; atomic_load(&k->generation) == gen; spins++) {
        if (atomic_load(&k->error)) return ncclSystemError;
        if (spins > 200) usleep(50);
        if (fk_now() - t0 > lim) { fk_fail(c, "barrier timed out (a peer is missing, or the ranks disagree on the sequence of calls)"); return ncclSystemError; }
    }
    return atomic_load(&k->error) ? ncclSystemError : ncclSuccess;
}
static unsigned char *fk_slot(struct ncclComm *c, int r) { return c->base + c->ctl->data_off + (size_t)r * c->ctl->slot_bytes; }

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "/tmp/trc_fake_rccl_%d_%llx", (int)getpid(), (unsigned long long)(fk_now() * 1e6));
    const int fd = open(id->internal, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    if (ftruncate(fd, sizeof(fk_ctl)) != 0) { close(fd); return ncclSystemError; }
    close(fd);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int world, ncclUniqueId id, int rank)
{
    if (world < 1 || world > FK_MAXW || rank < 0 || rank >= world) return ncclInvalidArgument;
    struct ncclComm *c = (struct ncclComm *)calloc(1, sizeof *c);
    c->rank = rank; c->world = world;
    memcpy(c->path, id.internal, sizeof c->path); c->path[sizeof c->path - 1] = 0;
    const char *mb = getenv("TRC_FAKE_RCCL_SLOT_MB");
    const uint64_t slot = (uint64_t)(mb ? atoi(mb) : 256) << 20, data_off = (sizeof(fk_ctl) + 4095) & ~(size_t)4095;
    c->map_bytes = data_off + (size_t)world * slot;
    const double t0 = fk_now();
    int fd = -1;
    while ((fd = open(c->path, O_RDWR)) < 0) { if (fk_now() - t0 > fk_limit()) { free(c); return ncclSystemError; } usleep(1000); }
    if (rank == 0 && ftruncate(fd, (off_t)c->map_bytes) != 0) { close(fd); free(c); return ncclSystemError; }   /* sparse: only touched pages exist */
    if (rank != 0) {                                             /* wait until rank 0 has sized the file */
        struct stat st;
        while (fstat(fd, &st) == 0 && (size_t)st.st_size < c->map_bytes) { if (fk_now() - t0 > fk_limit()) { close(fd); free(c); return ncclSystemError; } usleep(1000); }
    }
    c->base = (unsigned char *)mmap(0, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->base == MAP_FAILED) { free(c); return ncclSystemError; }
    c->ctl = (fk_ctl *)c->base;
    if (rank == 0) { c->ctl->world = (uint32_t)world; c->ctl->slot_bytes = slot; c->ctl->data_off = data_off; atomic_store(&c->ctl->magic, FK_MAGIC); }
    else while (atomic_load(&c->ctl->magic) != FK_MAGIC) { if (fk_now() - t0 > fk_limit()) return ncclSystemError; usleep(1000); }
    if (c->ctl->world != (uint32_t)world) { fk_fail(c, "ranks disagree on the world size"); return ncclInvalidArgument; }
    *out = c; g_last_comm = c;
    return fk_barrier(c);
}

ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (!c) return ncclSuccess;
    fk_barrier(c);
    if (c->rank == 0) unlink(c->path);
    munmap(c->base, c->map_bytes);
    free(c);
    return ncclSuccess;
}
ncclResult_t ncclCommCount(const ncclComm_t c, int *n) { *n = c->world; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t c, int *r) { *r = c->rank; return ncclSuccess; }
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : r == ncclInvalidArgument ? "invalid argument (fake_rccl)" : "system error (fake_rccl: see stderr)"; }

ncclResult_t ncclAllGather(const void *sendbuf, void *recvbuf, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t s)
{
    const size_t bytes = count * fk_size(t);
    if (g_depth) { fk_fail(c, "collective inside a group"); return ncclInvalidUsage; }
    if (!fk_size(t) || bytes > c->ctl->slot_bytes) return ncclInvalidArgument;
    if (hipStreamSynchronize(s) != hipSuccess || hipMemcpy(fk_slot(c, c->rank), sendbuf, bytes, hipMemcpyDeviceToHost) != hipSuccess) { fk_fail(c, "all-gather: copy out failed"); return ncclUnhandledCudaError; }
    ncclResult_t r = fk_barrier(c); if (r != ncclSuccess) return r;
    for (int p = 0; p < c->world; p++)
        if (hipMemcpy((unsigned char *)recvbuf + (size_t)p * bytes, fk_slot(c, p), bytes, hipMemcpyHostToDevice) != hipSuccess) { fk_fail(c, "all-gather: copy in failed"); return ncclUnhandledCudaError; }
    return fk_barrier(c);
}

ncclResult_t ncclAllReduce(const void *sendbuf, void *recvbuf, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t s)
{
    const size_t bytes = count * 8;
    if (g_depth) { fk_fail(c, "collective inside a group"); return ncclInvalidUsage; }
    if (t != ncclUint64 || op != ncclSum || bytes > c->ctl->slot_bytes) { fk_fail(c, "all-reduce: only u64 sum"); return ncclInvalidArgument; }
    if (hipStreamSynchronize(s) != hipSuccess || hipMemcpy(fk_slot(c, c->rank), sendbuf, bytes, hipMemcpyDeviceToHost) != hipSuccess) { fk_fail(c, "all-reduce: copy out failed"); return ncclUnhandledCudaError; }
    ncclResult_t r = fk_barrier(c); if (r != ncclSuccess) return r;
    uint64_t *acc = (uint64_t *)calloc(count, 8);
    for (int p = 0; p < c->world; p++) { const uint64_t *v = (const uint64_t *)fk_slot(c, p); for (size_t i = 0; i < count; i++) acc[i] += v[i]; }
    const hipError_t e = hipMemcpy(recvbuf, acc, bytes, hipMemcpyHostToDevice);
    free(acc);
    if (e != hipSuccess) { fk_fail(c, "all-reduce: copy in failed"); return ncclUnhandledCudaError; }
    return fk_barrier(c);
}

/* one round of point-to-point transfers: everything posted since the outermost ncclGroupStart */
static ncclResult_t fk_round(void)
{
    struct ncclComm *c = 0;
    for (int i = 0; i < g_nops; i++) { if (c && g_ops[i].comm != c) { fk_fail(c, "one communicator per group"); return ncclInvalidUsage; } c = g_ops[i].comm; }
    if (!c) c = g_last_comm;                                     /* nothing posted here: still take part (a peer may have sent to us) */
    if (!c) return ncclSuccess;
    fk_ctl *k = c->ctl;
    uint64_t off = 0; uint32_t ns = 0;
    for (int i = 0; i < g_nops; i++) {
        fk_op *o = &g_ops[i];
        if (hipStreamSynchronize(o->stream) != hipSuccess) { fk_fail(c, "stream sync failed"); return ncclUnhandledCudaError; }
        if (!o->is_send) continue;
        if (off + o->bytes > k->slot_bytes || ns >= FK_MAXOPS) { fk_fail(c, "outbox full (TRC_FAKE_RCCL_SLOT_MB)"); return ncclSystemError; }
        if (hipMemcpy(fk_slot(c, c->rank) + off, o->buf, o->bytes, hipMemcpyDeviceToHost) != hipSuccess) { fk_fail(c, "send: copy out failed"); return ncclUnhandledCudaError; }
        k->r[c->rank].send[ns].dst = o->peer; k->r[c->rank].send[ns].bytes = o->bytes; k->r[c->rank].send[ns].off = off;
        ns++; off += (o->bytes + 63) & ~(uint64_t)63;
    }
    k->r[c->rank].nsend = ns;
    ncclResult_t r = fk_barrier(c); if (r != ncclSuccess) return r;
    uint32_t cursor[FK_MAXW] = {0}, taken[FK_MAXW] = {0};
    for (int i = 0; i < g_nops; i++) {
        fk_op *o = &g_ops[i];
        if (o->is_send) continue;
        const int p = o->peer;
        uint32_t q = cursor[p];
        while (q < k->r[p].nsend && k->r[p].send[q].dst != c->rank) q++;
        if (q == k->r[p].nsend) { fk_fail(c, "receive without a matching send in this group (RCCL would hang)"); return ncclSystemError; }
        if (k->r[p].send[q].bytes != o->bytes) { char m[160]; snprintf(m, sizeof m, "receive of %zu bytes from rank %d meets a send of %llu bytes", o->bytes, p, (unsigned long long)k->r[p].send[q].bytes); fk_fail(c, m); return ncclSystemError; }
        if (hipMemcpy(o->buf, fk_slot(c, p) + k->r[p].send[q].off, o->bytes, hipMemcpyHostToDevice) != hipSuccess) { fk_fail(c, "receive: copy in failed"); return ncclUnhandledCudaError; }
        cursor[p] = q + 1; taken[p]++;
    }
    for (int p = 0; p < c->world; p++) {
        uint32_t addressed = 0;
        for (uint32_t q = 0; q < k->r[p].nsend; q++) addressed += k->r[p].send[q].dst == c->rank;
        if (addressed != taken[p]) { char m[160]; snprintf(m, sizeof m, "rank %d sent %u message(s) here, %u were received (RCCL would hang)", p, addressed, taken[p]); fk_fail(c, m); return ncclSystemError; }
    }
    return fk_barrier(c);
}

ncclResult_t ncclGroupStart(void) { g_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd(void)
{
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth) return ncclSuccess;
    const ncclResult_t r = fk_round();
    g_nops = 0;
    return r;
}
static ncclResult_t fk_post(int is_send, void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s)
{
    if (!fk_size(t) || peer < 0 || peer >= c->world || peer == c->rank) return ncclInvalidArgument;
    if (g_nops >= FK_MAXOPS) return ncclSystemError;
    const int single = !g_depth;
    if (single) g_depth = 1;
    g_ops[g_nops].is_send = is_send; g_ops[g_nops].buf = buf; g_ops[g_nops].bytes = count * fk_size(t); g_ops[g_nops].peer = peer;
    g_ops[g_nops].comm = c; g_ops[g_nops].stream = s; g_nops++;
    return single ? ncclGroupEnd() : ncclSuccess;
}
ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s) { g_last_comm = c; return fk_post(1, (void *)buf, count, t, peer, c, s); }
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s) { g_last_comm = c; return fk_post(0, buf, count, t, peer, c, s); }
