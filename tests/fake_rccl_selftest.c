/* fake_rccl_selftest.c -- TEST INFRASTRUCTURE: checks the checker.  Two ranks (processes on device 0) over tests/fake_rccl.c:
 *   case 0  matched grouped send/receive both ways + all-gather + all-reduce: data arrives, everything returns success
 *   case 1  a receive that meets a send of a different size            -> the group must end with an error
 *   case 2  a send nobody receives                                     -> the group must end with an error
 *   case 3  a receive nobody sends to                                  -> the group must end with an error
 * Exit status 0 = the case behaved as stated.  usage: fake_rccl_selftest <path to libfake_rccl.so> <case> */
#define _GNU_SOURCE 1
#define __HIP_PLATFORM_AMD__ 1
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <unistd.h>

#define SYM(n) __typeof__(&n) p_##n = (__typeof__(&n))dlsym(h, #n); if (!p_##n) { fprintf(stderr, "missing %s\n", #n); return 9; }

static int rank_main(void *h, int rank, int tcase, const char *idfile)
{
    SYM(ncclGetUniqueId) SYM(ncclCommInitRank) SYM(ncclCommDestroy) SYM(ncclSend) SYM(ncclRecv) SYM(ncclGroupStart) SYM(ncclGroupEnd) SYM(ncclAllGather) SYM(ncclAllReduce)
    ncclUniqueId id;
    if (rank == 0) {
        if (p_ncclGetUniqueId(&id) != ncclSuccess) return 8;
        FILE *f = fopen(idfile, "wb"); fwrite(&id, sizeof id, 1, f); fclose(f);
        char done[300]; snprintf(done, sizeof done, "%s.ok", idfile); f = fopen(done, "wb"); fclose(f);
    } else {
        char done[300]; snprintf(done, sizeof done, "%s.ok", idfile);
        for (int i = 0; i < 3000 && access(done, F_OK) != 0; i++) usleep(10000);
        FILE *f = fopen(idfile, "rb"); if (!f || fread(&id, sizeof id, 1, f) != 1) return 8; fclose(f);
    }
    if (hipSetDevice(0) != hipSuccess) return 8;
    ncclComm_t c;
    if (p_ncclCommInitRank(&c, 2, id, rank) != ncclSuccess) return 8;
    const size_t n = 1000;
    uint32_t *d_a, *d_b, hbuf[2000];
    uint64_t *d_r, hr[4];
    hipMalloc((void **)&d_a, 8000); hipMalloc((void **)&d_b, 8000); hipMalloc((void **)&d_r, 64);
    for (size_t i = 0; i < n; i++) hbuf[i] = (uint32_t)(rank * 100000 + i);
    hipMemcpy(d_a, hbuf, 4 * n, hipMemcpyHostToDevice); hipMemset(d_b, 0, 8000);
    const int peer = 1 - rank;
    ncclResult_t r = ncclSuccess, r2;
    int rc = 0;
    if (tcase == 0) {
        p_ncclGroupStart();
        if (rank == 0) { p_ncclSend(d_a, n, ncclUint32, peer, c, 0); p_ncclRecv(d_b, n, ncclUint32, peer, c, 0); }
        else { p_ncclRecv(d_b, n, ncclUint32, peer, c, 0); p_ncclSend(d_a, n, ncclUint32, peer, c, 0); }
        r = p_ncclGroupEnd();
        hipMemcpy(hbuf, d_b, 4 * n, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < n; i++) if (hbuf[i] != (uint32_t)(peer * 100000 + i)) rc = 1;
        r2 = p_ncclAllGather(d_a, d_b, 10, ncclUint32, c, 0);
        hipMemcpy(hbuf, d_b, 80, hipMemcpyDeviceToHost);
        for (int p = 0; p < 2; p++) for (int i = 0; i < 10; i++) if (hbuf[p * 10 + i] != (uint32_t)(p * 100000 + i)) rc = 1;
        hr[0] = 5 + rank; hr[1] = 1ull << 40;
        hipMemcpy(d_r, hr, 16, hipMemcpyHostToDevice);
        if (p_ncclAllReduce(d_r, d_r, 2, ncclUint64, ncclSum, c, 0) != ncclSuccess) rc = 1;
        hipMemcpy(hr, d_r, 16, hipMemcpyDeviceToHost);
        if (hr[0] != 11 || hr[1] != (2ull << 40)) rc = 1;
        if (r != ncclSuccess || r2 != ncclSuccess) rc = 1;
        p_ncclCommDestroy(c);
        return rc;
    }
    p_ncclGroupStart();
    if (tcase == 1) { if (rank == 0) p_ncclSend(d_a, n, ncclUint32, peer, c, 0); else p_ncclRecv(d_b, n - 1, ncclUint32, peer, c, 0); }
    if (tcase == 2) { if (rank == 0) p_ncclSend(d_a, n, ncclUint32, peer, c, 0); }
    if (tcase == 3) { if (rank == 1) p_ncclRecv(d_b, n, ncclUint32, peer, c, 0); }
    r = p_ncclGroupEnd();
    /* the rank that can see the defect must get an error; the other may get it at this or at its next call */
    const int sees = (tcase == 1 && rank == 1) || (tcase == 2 && rank == 1) || (tcase == 3 && rank == 1);
    if (sees && r == ncclSuccess) return 1;
    if (!sees && r == ncclSuccess && p_ncclAllGather(d_a, d_b, 10, ncclUint32, c, 0) == ncclSuccess) return 1;
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    const int tcase = atoi(argv[2]);
    char idfile[256]; snprintf(idfile, sizeof idfile, "/tmp/fake_rccl_selftest_%d.id", (int)getpid());
    setenv("TRC_FAKE_RCCL_TIMEOUT", "10", 0);
    setenv("TRC_FAKE_RCCL_SLOT_MB", "4", 0);
    pid_t pid[2];
    for (int r = 0; r < 2; r++) {
        pid[r] = fork();
        if (pid[r] == 0) { void *h = dlopen(argv[1], RTLD_NOW); if (!h) { fprintf(stderr, "%s\n", dlerror()); _exit(9); } _exit(rank_main(h, r, tcase, idfile)); }
    }
    int bad = 0;
    for (int r = 0; r < 2; r++) { int st = 0; waitpid(pid[r], &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st)) { fprintf(stderr, "case %d: rank %d status %d\n", tcase, r, WIFEXITED(st) ? WEXITSTATUS(st) : -1); bad = 1; } }
    char done[300]; snprintf(done, sizeof done, "%s.ok", idfile); unlink(done); unlink(idfile);
    printf("case %d: %s\n", tcase, bad ? "FAILED" : "ok");
    return bad;
}
