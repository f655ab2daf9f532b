"""Multi-GPU sharding of the chunked coders: one process per GPU, torch.distributed.

Chunks are independent, so the path shards by contiguous chunk ranges with NO data-path collective
inside the coders.  The only exchanges are
  * (static coders) one all-reduce of the 256-bin byte histogram, so every rank builds the same CDF
    and the assembled container equals the single-GPU container of the whole input bit for bit;
  * the gather of the per-rank results to rank 0: all_gather of the payload sizes, then one
    point-to-point transfer per peer (over xGMI each rides its own direct link; a ring would
    serialise on single links) for the directory slice and the payload.
  * the same gather for a GROUP of batches at once (`exchange_group`): batch j of the group goes to root j mod world,
    all transfers of the group in ONE grouped send/receive call.  With a fixed root only the root's links carry
    payload and a step can never be shorter than C / link; with the roots rotating every directed link carries one
    payload per `world` batches and all of them move at the same time.
Everything here works on CPU tensors with the gloo backend too (tests/test_shard_gloo.py).
"""
import struct

import numpy as np

HDR = 32
MAGIC = 0x31435254


def shard_bounds(n, world, chunk):
    """-> list of (byte_start, byte_len) per rank: contiguous ranges of whole chunks, last rank ragged."""
    nch = (n + chunk - 1) // chunk
    per, rem = divmod(nch, world)
    out, c = [], 0
    for r in range(world):
        k = per + (1 if r < rem else 0)
        start = c * chunk
        end = min(n, (c + k) * chunk)
        out.append((start, max(0, end - start)))
        c += k
    return out


def allreduce_hist(dist, hist):
    """sum the per-rank 256-bin histograms in place (int64 tensor on the backend's device)"""
    dist.all_reduce(hist, op=dist.ReduceOp.SUM)
    return hist


def gather_to_root(dist, rank, world, total, clen, payload, recv_clen=None, recv_payload=None):
    """Gather every rank's (clen slice, payload) to rank 0.

    total   : int64[1] tensor   this rank's payload bytes
    clen    : int32 tensor      this rank's chunk lengths (already trimmed to its chunk count)
    payload : uint8 tensor      at least total bytes
    recv_*  : rank 0 only: lists (len world-1) of pre-allocated receive tensors, or None to allocate
    Returns on rank 0: (sizes list, [clen tensors per rank], [payload tensors per rank]); else (sizes, None, None).
    """
    import torch
    meta = torch.stack([total.reshape(()).to(torch.int64), torch.tensor(clen.numel(), dtype=torch.int64, device=total.device)])
    allmeta = torch.empty(2 * world, dtype=torch.int64, device=total.device)
    dist.all_gather_into_tensor(allmeta, meta)
    m = allmeta.tolist()
    sizes = [(m[2 * r], m[2 * r + 1]) for r in range(world)]
    ops = []
    if rank == 0:
        cl = [clen]
        pl = [payload[:sizes[0][0]]]
        for r in range(1, world):
            tb, nc = sizes[r]
            c_r = recv_clen[r - 1][:nc] if recv_clen else torch.empty(nc, dtype=clen.dtype, device=clen.device)
            p_r = recv_payload[r - 1][:tb] if recv_payload else torch.empty(tb, dtype=torch.uint8, device=payload.device)
            cl.append(c_r); pl.append(p_r)
            if nc:
                ops.append(dist.P2POp(dist.irecv, c_r, r))
            if tb:
                ops.append(dist.P2POp(dist.irecv, p_r, r))
    else:
        cl = pl = None
        tb, nc = sizes[rank]
        if nc:
            ops.append(dist.P2POp(dist.isend, clen, 0))
        if tb:
            ops.append(dist.P2POp(dist.isend, payload[:tb], 0))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return sizes, cl, pl


def group_plan(k, group, last):
    """Where step k of a run goes when results are exchanged `group` steps at a time out of two alternating banks of
    result buffers: -> (slot in the group, bank, ns) with ns = number of steps to exchange once this step is coded
    (0: the group is still filling; the group's size when it is complete, or its fill when `last` flushes it)."""
    j, bank = k % group, (k // group) % 2
    return j, bank, (j + 1 if (j == group - 1 or last) else 0)


def exchange_group(dist, rank, world, totals, clens, payloads, recv_clen=None, recv_payload=None):
    """Gather the results of len(totals) consecutive batches, batch j onto rank j % world, in one grouped P2P call.

    totals[j]   : int64[1] tensor   this rank's payload bytes of batch j
    clens[j]    : int32 tensor      this rank's chunk lengths of batch j (trimmed to its chunk count)
    payloads[j] : uint8 tensor      at least totals[j] bytes
    recv_*      : {j: list (len world-1, peers in rank order without the root) of receive tensors} for the batches
                  this rank is the root of, or None to allocate
    Returns (sizes, got): sizes[j][r] = (payload bytes, chunks) of rank r in batch j; got[j] = (clen list, payload
    list) in rank order for the batches rooted here (this rank's own pieces included, not copied).
    Pairs of ranks see their transfers in the same order on both sides (batch order, directory before payload), which
    is all a grouped send/receive needs; a group of one batch rooted at rank 0 is `gather_to_root`.
    """
    import torch
    ns = len(totals)
    dev = totals[0].device
    meta = torch.stack([t.reshape(()).to(torch.int64) for t in totals] +
                       [torch.tensor(c.numel(), dtype=torch.int64, device=dev) for c in clens])
    allmeta = torch.empty(2 * ns * world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(allmeta, meta)
    m = allmeta.tolist()
    sizes = [[(m[2 * ns * r + j], m[2 * ns * r + ns + j]) for r in range(world)] for j in range(ns)]
    ops, got = [], {}
    for j in range(ns):
        root = j % world
        if rank == root:
            cl, pl, k = [], [], 0
            for r in range(world):
                tb, nc = sizes[j][r]
                if r == rank:
                    cl.append(clens[j]); pl.append(payloads[j][:tb])
                    continue
                c_r = recv_clen[j][k][:nc] if recv_clen else torch.empty(nc, dtype=clens[j].dtype, device=dev)
                p_r = recv_payload[j][k][:tb] if recv_payload else torch.empty(tb, dtype=torch.uint8, device=dev)
                k += 1
                cl.append(c_r); pl.append(p_r)
                if nc:
                    ops.append(dist.P2POp(dist.irecv, c_r, r))
                if tb:
                    ops.append(dist.P2POp(dist.irecv, p_r, r))
            got[j] = (cl, pl)
        else:
            tb, nc = sizes[j][rank]
            if nc:
                ops.append(dist.P2POp(dist.isend, clens[j], root))
            if tb:
                ops.append(dist.P2POp(dist.isend, payloads[j][:tb], root))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return sizes, got


def assemble_container(codec, n, chunk, cdfnum, clens, payloads):
    """rank 0: TRC1 container bytes (include/trc_hip.h) from the gathered per-rank pieces (numpy arrays)."""
    clen = np.concatenate([np.asarray(c, dtype=np.uint32) for c in clens]) if clens else np.zeros(0, np.uint32)
    pay = np.concatenate([np.asarray(p, dtype=np.uint8) for p in payloads]) if payloads else np.zeros(0, np.uint8)
    hdr = struct.pack("<IBBHIIQQ", MAGIC, codec, 1, cdfnum, chunk, clen.size, n, pay.size)
    return np.concatenate([np.frombuffer(hdr, dtype=np.uint8), clen.view(np.uint8), pay])
