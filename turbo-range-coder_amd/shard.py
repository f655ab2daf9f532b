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
_COUNTS = {}                                                    # exchange_group: device tensors of the chunk counts of a group


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


def exchange_group(dist, rank, world, totals, clens, payloads, recv_clen=None, recv_payload=None, meta=None):
    """Gather the results of len(totals) consecutive batches, batch j onto rank j % world, in one grouped P2P call.

    totals[j]   : int64[1] tensor   this rank's payload bytes of batch j
    clens[j]    : int32 tensor      this rank's chunk lengths of batch j (trimmed to its chunk count)
    payloads[j] : uint8 tensor      at least totals[j] bytes
    recv_*      : {j: list (len world-1, peers in rank order without the root) of receive tensors} for the batches
                  this rank is the root of, or None to allocate
    Returns (sizes, got): sizes[j][r] = (payload bytes, chunks) of rank r in batch j; got[j] = (clen list, payload
    list) in rank order for the batches rooted here (this rank's own pieces included, not copied).
    meta        : optional int64 tensor of 2 * len(totals) elements, (payload bytes, chunks) of batch j at [2j], [2j + 1], that the
                  caller keeps up to date by itself -- the totals are VIEWS into it that the encoder writes, the counts are written
                  once -- so that a group's exchange starts with no kernel of its own (round 6: building it from eight one-element
                  tensors was nine small launches per group on the side stream)
    Pairs of ranks see their transfers in the same order on both sides (batch order, directory before payload), which
    is all a grouped send/receive needs; a group of one batch rooted at rank 0 is `gather_to_root`.
    """
    import torch
    ns = len(totals)
    dev = totals[0].device
    # the chunk counts are host numbers: one cached device tensor per shape of group (round 4 -- eight torch.tensor(n, device=...)
    # per group were eight blocking host-to-device copies on the side stream, each waiting for the group's last encode with the
    # HOST: 36 us per step of idle GPU at groups of 8 on one rank, profiles/r04_notes.md)
    allmeta = torch.empty(2 * ns * world, dtype=torch.int64, device=dev)
    if meta is not None:
        dist.all_gather_into_tensor(allmeta, meta.reshape(-1)[:2 * ns])
        m = allmeta.tolist()
        sizes = [[(m[2 * ns * r + 2 * j], m[2 * ns * r + 2 * j + 1]) for r in range(world)] for j in range(ns)]
    else:
        key = (str(dev), tuple(int(c.numel()) for c in clens))
        counts = _COUNTS.get(key)
        if counts is None:
            counts = _COUNTS[key] = torch.tensor(key[1], dtype=torch.int64).to(dev)
        meta = torch.cat([torch.cat([t.reshape(1).to(torch.int64) for t in totals]), counts])
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


class StagedDist:
    """torch.distributed over a backend that moves HOST tensors (gloo), for DEVICE tensors: every call stages through host
    memory -- the current stream is synchronised before a tensor is read, results are copied back on the current stream.
    Not a transport of the product: it lets the device-side schedule of `bench.py --gpus N` (StepPipeline on CudaRuntime: side
    stream, events, both banks, rotating roots, preallocated receive buffers, views of CUDA tensors) run against REAL peer
    processes on a box where RCCL has nobody to talk to (`bench.py --backend gloo --same-device`: all ranks on cuda:0).
    Pairs of ranks post their transfers in the same order on both sides, which is what gloo matches by."""
    class P2POp:
        def __init__(self, op, tensor, peer):
            self.op, self.tensor, self.peer = op, tensor, peer

    class _Work:
        def __init__(self, work, host, dst):
            self.work, self.host, self.dst = work, host, dst

        def wait(self):
            self.work.wait()
            if self.dst is not None:
                self.dst.copy_(self.host)                       # on the current stream (the pipeline's side stream)

    def __init__(self, dist, torch):
        self.dist, self.torch = dist, torch
        self.ReduceOp = dist.ReduceOp
        self.isend, self.irecv = "isend", "irecv"

    def _sync(self, t):
        if t.is_cuda:
            self.torch.cuda.current_stream(t.device).synchronize()

    def barrier(self):
        self.dist.barrier()

    def all_reduce(self, t, op=None):
        op = self.dist.ReduceOp.SUM if op is None else op
        self._sync(t)
        h = t.cpu()
        self.dist.all_reduce(h, op=op)
        t.copy_(h)

    def all_gather_into_tensor(self, out, inp):
        self._sync(inp)
        h = inp.cpu().contiguous()
        parts = [self.torch.empty_like(h) for _ in range(self.dist.get_world_size())]
        self.dist.all_gather(parts, h)
        out.copy_(self.torch.cat(parts))

    def all_gather_object(self, lst, obj):
        self.dist.all_gather_object(lst, obj)

    def batch_isend_irecv(self, ops):
        works = []
        if ops:
            self._sync(ops[0].tensor)
        for o in ops:
            if o.op == "isend":
                h = o.tensor.cpu().contiguous()
                works.append(self._Work(self.dist.isend(h, o.peer), h, None))
            else:
                h = self.torch.empty(o.tensor.shape, dtype=o.tensor.dtype)
                works.append(self._Work(self.dist.irecv(h, o.peer), h, o.tensor))
        return works

    def destroy_process_group(self):
        self.dist.destroy_process_group()


class HostRuntime:
    """Stream/event plumbing of StepPipeline for CPU tensors (gloo): everything is synchronous, events are no-ops."""
    def record_main(self):
        return None

    def main_wait(self, ev):
        pass

    def side_wait(self, ev):
        pass

    def on_side(self):
        import contextlib
        return contextlib.nullcontext()

    def record_side(self):
        return None


class CudaRuntime:
    """... for one GPU: `main` = torch's current stream (where the coder kernels are enqueued), one side stream for the
    exchanges."""
    def __init__(self, torch, dev):
        self.torch, self.dev = torch, dev
        self.side = torch.cuda.Stream(device=dev)

    def record_main(self):
        ev = self.torch.cuda.Event()
        ev.record(self.torch.cuda.current_stream(self.dev))
        return ev

    def main_wait(self, ev):
        if ev is not None:
            self.torch.cuda.current_stream(self.dev).wait_event(ev)

    def side_wait(self, ev):
        if ev is not None:
            self.side.wait_event(ev)

    def on_side(self):
        return self.torch.cuda.stream(self.side)

    def record_side(self):
        ev = self.torch.cuda.Event()
        ev.record(self.side)
        return ev


class StepPipeline:
    """The schedule `bench.py --gpus N` runs: code a step, exchange finished GROUPS of steps behind the coder's back.

    Steps are exchanged in groups of `group` (= world with rotating roots: step j of a group is gathered onto rank j;
    = 1 for the plain per-step gather to rank 0).  A group's exchange (exchange_group: one all_gather of sizes + ONE
    grouped send/receive call) runs on the side stream while the next group is coded into the OTHER bank of result
    buffers; a bank is coded into again only after its own exchange has finished (event).  The exchange reads the sizes
    on the host, so it is issued one step late -- after the next step's kernels are in the queue -- and the last,
    possibly partial, group of a run is flushed by the step flagged `last`.

    banks      [2][group] result buffers: tuples (clen int32 tensor, payload uint8 tensor, total int64[>=1] tensor)
    recv       [2] of (list of world-1 clen receive tensors, list of world-1 payload receive tensors) or None
               (ranks that are never a root, or tests that let exchange_group allocate)
    nch        chunks per step on this rank (clen tensors are trimmed to it)
    on_gathered(first_step_of_group, sizes, got)  optional hook, called on every rank after its part of a group's
               exchange has been issued (tests assert every step's container there)
    """
    def __init__(self, dist, rank, world, group, banks, recv, nch, rt, rotate=True, on_gathered=None, lag=1, metas=None):
        self.dist, self.rank, self.world, self.group = dist, rank, world, group
        self.banks, self.recv, self.nch, self.rt, self.rotate = banks, recv, nch, rt, rotate
        self.on_gathered = on_gathered
        self.metas = metas                                     # [2] int64 tensors [group, 2]: (total, chunks) per step of the bank; the totals of `banks` are views into them
        self.done = [None, None]
        self.pending = []
        # `lag`: how many steps after a group's last step its exchange is issued (the host reads the sizes then and waits for
        # that group's last encode: the more steps are queued behind it, the less the GPU can run dry meanwhile; the later the
        # transfers start).  1 = rounds 2-3.  At most `group`: the bank is coded into again `group` steps later.
        self.lag = max(1, min(int(lag), max(1, group)))

    def reset(self):
        """between independent runs (set-up / warmup / timed): forget finished exchanges"""
        assert not self.pending
        self.done = [None, None]

    def _exchange(self, bank, ns, first):
        res = self.banks[bank][:ns]
        mine = self.rank if self.rotate else 0                 # the step of the group this rank is the root of
        rc = rp = None
        if self.recv[bank] is not None and mine < ns and (self.rotate or self.rank == 0):
            rc, rp = {mine: self.recv[bank][0]}, {mine: self.recv[bank][1]}
        sizes, got = exchange_group(self.dist, self.rank, self.world, [r[2][:1] for r in res], [r[0][:self.nch] for r in res],
                                    [r[1] for r in res], rc, rp, meta=None if self.metas is None else self.metas[bank])
        if self.on_gathered:
            self.on_gathered(first, sizes, got)

    def _run_pending(self, upto=None):
        """issue the exchanges that are due (all of them with upto=None)"""
        while self.pending and (upto is None or self.pending[0][4] <= upto):
            bank, ns, coded, first, _ = self.pending.pop(0)
            self.rt.side_wait(coded)                           # the exchange may start once the group's last encode is done
            with self.rt.on_side():
                self._exchange(bank, ns, first)
                self.done[bank] = self.rt.record_side()

    def step(self, k, last, encode, decode):
        """step k of a run: encode(result) codes this rank's shard into the result buffers `result`, decode(result)
        decodes them again (both only enqueue work on the main stream)"""
        j, bank, ns = group_plan(k, self.group, last)
        if j == 0:
            if any(p[0] == bank for p in self.pending):        # (lag == group: the exchange out of this bank must be issued before it is waited for)
                self._run_pending()
            self.rt.main_wait(self.done[bank])                 # the previous exchange out of this bank is done
        result = self.banks[bank][j]
        encode(result)
        coded = self.rt.record_main() if ns else None
        decode(result)
        self._run_pending(k)                                   # `lag` steps late: the GPU has those steps' kernels queued meanwhile
        if ns:
            self.pending.append((bank, ns, coded, k - j, k + self.lag))
        if last:
            self._run_pending()


def assemble_container(codec, n, chunk, cdfnum, clens, payloads):
    """rank 0: TRC1 container bytes (include/trc_hip.h) from the gathered per-rank pieces (numpy arrays)."""
    clen = np.concatenate([np.asarray(c, dtype=np.uint32) for c in clens]) if clens else np.zeros(0, np.uint32)
    pay = np.concatenate([np.asarray(p, dtype=np.uint8) for p in payloads]) if payloads else np.zeros(0, np.uint8)
    hdr = struct.pack("<IBBHIIQQ", MAGIC, codec, 1, cdfnum, chunk, clen.size, n, pay.size)
    return np.concatenate([np.frombuffer(hdr, dtype=np.uint8), clen.view(np.uint8), pay])
