// trc_rccl.hip -- the path's one exchange step behind the C-ABI: the gather of per-rank results over RCCL (xGMI).
//
// Chunks shard with no data-path collective; what moves between GPUs is the RESULT: every rank's directory slice
// and payload go to one GPU.  trc_exchange_dev gathers `nbatch` consecutive results at once, batch j onto rank
// j % world, with ONE all-gather of the sizes and ONE group of point-to-point transfers (ncclGroupStart ... ncclSend /
// ncclRecv ... ncclGroupEnd): xGMI is point-to-point, so the transfers into a root each ride their own link, and with
// the roots rotating every directed link carries one payload per `world` batches, all at the same time.
// nbatch = 1 is the plain gather onto rank 0.  Same schedule as turbo-range-coder_amd/shard.py (exchange_group), which
// tests/test_shard_gloo.py runs on CPU tensors over gloo.
//
// RCCL is resolved at run time: the file named by TRC_RCCL_LIB if that is set (a site's own build; the test suite's
// tests/fake_rccl.c, which lets world > 1 run on one GPU), else what the process has already loaded (dlsym -- e.g. the
// librccl.so inside a PyTorch wheel), else dlopen of librccl.so.1.  The library itself does not depend on RCCL, a
// single-GPU user never loads it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <mutex>
#include <stdio.h>
#include <string.h>
#include "../../include/trc_hip.h"

namespace {
struct Rccl {
    bool ok = false;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::once_flag g_rccl_once;
void rccl_load()
{
    void *h = nullptr;
    const char *forced = getenv("TRC_RCCL_LIB");
    if (forced && *forced) {
        h = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);            // the same handle a caller's own dlopen of that file returns
        if (!h) { fprintf(stderr, "libturborc_hip: TRC_RCCL_LIB=%s: %s\n", forced, dlerror()); return; }
    } else {
        h = RTLD_DEFAULT;
        if (!dlsym(h, "ncclSend")) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
    }
    Rccl &r = g_rccl;
#define TRC_SYM(field, name) *(void **)(&r.field) = dlsym(h, name)
    TRC_SYM(AllGather, "ncclAllGather"); TRC_SYM(AllReduce, "ncclAllReduce"); TRC_SYM(Send, "ncclSend"); TRC_SYM(Recv, "ncclRecv");
    TRC_SYM(GroupStart, "ncclGroupStart"); TRC_SYM(GroupEnd, "ncclGroupEnd");
    TRC_SYM(CommCount, "ncclCommCount"); TRC_SYM(CommUserRank, "ncclCommUserRank"); TRC_SYM(GetErrorString, "ncclGetErrorString");
#undef TRC_SYM
    r.ok = r.AllGather && r.AllReduce && r.Send && r.Recv && r.GroupStart && r.GroupEnd && r.CommCount && r.CommUserRank && r.GetErrorString;
}
}  // namespace

// {payload bytes, chunks} of every batch into the all-gather's send buffer.  The chunk counts travel as kernel arguments:
// nothing on the stream ever reads host memory of this call (round 2 copied them from a stack variable with hipMemcpyAsync).
struct MetaArgs { const uint64_t *total[TRC_EXCHANGE_MAX_BATCH]; uint64_t nchunks[TRC_EXCHANGE_MAX_BATCH]; };
__global__ void trc_exchange_meta_kernel(uint64_t *meta, MetaArgs a, int nbatch)
{
    const int j = threadIdx.x;
    if (j < nbatch) { meta[2 * j] = *a.total[j]; meta[2 * j + 1] = a.nchunks[j]; }
}

int trc_fail(int code, const char *fmt, ...);      // trc_api.hip: sets trc_last_error(), prints, returns code

#define RCHK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) return trc_fail(TRC_E_HIP, "%s -> %s", #x, g_rccl.GetErrorString(r_)); } while (0)
// inside ncclGroupStart ... ncclGroupEnd: a failing call must not leave the group open (every later RCCL call of the
// thread would be queued into it)
#define GCHK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { g_rccl.GroupEnd(); return trc_fail(TRC_E_HIP, "%s -> %s", #x, g_rccl.GetErrorString(r_)); } } while (0)
#define HCHK2(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return trc_fail(TRC_E_HIP, "%s -> %s", #x, hipGetErrorString(e_)); } while (0)

extern "C" int trc_exchange_dev(void *nccl_comm, int nbatch, const trc_batch *b, uint64_t *h_sizes, void *d_meta, void *stream)
{
    std::call_once(g_rccl_once, rccl_load);
    if (!g_rccl.ok) return trc_fail(TRC_E_NODEV, "RCCL not available (no ncclSend in the process and librccl.so.1 cannot be loaded)");
    if (!nccl_comm || nbatch < 1 || nbatch > TRC_EXCHANGE_MAX_BATCH || !b || !h_sizes || !d_meta) return trc_fail(TRC_E_ARG, "exchange: bad arguments");
    ncclComm_t comm = (ncclComm_t)nccl_comm;
    hipStream_t s = (hipStream_t)stream;
    int world = 0, rank = 0;
    RCHK(g_rccl.CommCount(comm, &world));
    RCHK(g_rccl.CommUserRank(comm, &rank));
    // 1. sizes: every rank contributes {payload bytes, chunks} per batch
    uint64_t *meta_mine = (uint64_t *)d_meta, *meta_all = meta_mine + 2 * (size_t)nbatch;
    MetaArgs ma;
    for (int j = 0; j < nbatch; j++) {
        if (!b[j].d_total) return trc_fail(TRC_E_ARG, "exchange: batch %d has no d_total", j);
        ma.total[j] = b[j].d_total; ma.nchunks[j] = b[j].nchunks;
    }
    hipLaunchKernelGGL(trc_exchange_meta_kernel, dim3(1), dim3(TRC_EXCHANGE_MAX_BATCH), 0, s, meta_mine, ma, nbatch);
    HCHK2(hipGetLastError());
    RCHK(g_rccl.AllGather(meta_mine, meta_all, 2 * (size_t)nbatch, ncclUint64, comm, s));
    HCHK2(hipMemcpyAsync(h_sizes, meta_all, 16 * (size_t)nbatch * world, hipMemcpyDeviceToHost, s));
    HCHK2(hipStreamSynchronize(s));                    // the transfer sizes are needed on the host (as in shard.exchange_group)
    // h_sizes[(r * nbatch + j) * 2 + {0,1}] = rank r's {bytes, chunks} of batch j
    // 2. all transfers of the group in one grouped call; pairs of ranks see theirs in the same order (batch order,
    //    directory before payload)
    for (int j = 0; j < nbatch; j++)
        if (rank == j % world && (!b[j].d_clen_all || !b[j].d_payload_all)) return trc_fail(TRC_E_ARG, "exchange: rank %d is the root of batch %d and has no receive buffers", rank, j);
    RCHK(g_rccl.GroupStart());
    for (int j = 0; j < nbatch; j++) {
        const int root = j % world;
        if (rank == root) {
            uint64_t coff = 0, poff = 0;
            for (int r = 0; r < world; r++) {
                const uint64_t bytes = h_sizes[((size_t)r * nbatch + j) * 2], nc = h_sizes[((size_t)r * nbatch + j) * 2 + 1];
                if (r != rank) {
                    if (nc) GCHK(g_rccl.Recv(b[j].d_clen_all + coff, nc, ncclUint32, r, comm, s));
                    if (bytes) GCHK(g_rccl.Recv((uint8_t *)b[j].d_payload_all + poff, bytes, ncclUint8, r, comm, s));
                }
                coff += nc; poff += bytes;
            }
        } else {
            const uint64_t bytes = h_sizes[((size_t)rank * nbatch + j) * 2], nc = h_sizes[((size_t)rank * nbatch + j) * 2 + 1];
            if (nc) GCHK(g_rccl.Send(b[j].d_clen, nc, ncclUint32, root, comm, s));
            if (bytes) GCHK(g_rccl.Send(b[j].d_payload, bytes, ncclUint8, root, comm, s));
        }
    }
    RCHK(g_rccl.GroupEnd());
    // 3. the root's own piece
    for (int j = 0; j < nbatch; j++) {
        if (rank != j % world) continue;
        uint64_t coff = 0, poff = 0;
        for (int r = 0; r < rank; r++) { poff += h_sizes[((size_t)r * nbatch + j) * 2]; coff += h_sizes[((size_t)r * nbatch + j) * 2 + 1]; }
        const uint64_t bytes = h_sizes[((size_t)rank * nbatch + j) * 2], nc = h_sizes[((size_t)rank * nbatch + j) * 2 + 1];
        if (nc && b[j].d_clen_all + coff != b[j].d_clen) HCHK2(hipMemcpyAsync(b[j].d_clen_all + coff, b[j].d_clen, 4 * nc, hipMemcpyDeviceToDevice, s));
        if (bytes && (uint8_t *)b[j].d_payload_all + poff != (const uint8_t *)b[j].d_payload)
            HCHK2(hipMemcpyAsync((uint8_t *)b[j].d_payload_all + poff, b[j].d_payload, bytes, hipMemcpyDeviceToDevice, s));
    }
    return TRC_OK;
}

// Histogram all-reduce for the static coders (one CDF for the whole job): 256 x u64 summed over the ranks, in place.
extern "C" int trc_hist_allreduce_dev(void *nccl_comm, uint64_t *d_hist, void *stream)
{
    std::call_once(g_rccl_once, rccl_load);
    if (!g_rccl.ok) return trc_fail(TRC_E_NODEV, "RCCL not available");
    if (!nccl_comm || !d_hist) return trc_fail(TRC_E_ARG, "hist all-reduce: bad arguments");
    RCHK(g_rccl.AllReduce(d_hist, d_hist, 256, ncclUint64, ncclSum, (ncclComm_t)nccl_comm, (hipStream_t)stream));
    return TRC_OK;
}
