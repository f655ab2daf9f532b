"""CPU, world_size 2, gloo: the N>1 path (shard -> histogram all-reduce -> per-rank coding ->
gather to rank 0 -> container) with the CPU oracle standing in for the HIP coders.  The assembled
container must equal the single-process container of the whole input, and decode back."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import trc_testlib as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "turbo-range-coder_amd"))
import shard  # noqa: E402


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, codec, n, chunk, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        data = T.zipf_bytes(n, 1.1, 256, 99)
        start, ln = shard.shard_bounds(n, world, chunk)[rank]
        mine = data[start:start + ln]
        cdf, cdfnum = None, 0
        if codec in T.STATIC_CODECS:
            hist = torch.from_numpy(np.bincount(mine, minlength=256).astype(np.int64))
            shard.allreduce_hist(dist, hist)
            # CDF from the GLOBAL histogram with the reference's rule == cdfini(whole input)
            _, cdf, cdfnum = T.orc_cdfini(data)
            assert np.array_equal(hist.numpy(), np.bincount(data, minlength=256))
        payload, clen, _ = T.orc_chunked_enc(codec, mine, chunk, cdf, cdfnum or 256) if ln else (np.zeros(0, np.uint8), np.zeros(0, np.uint32), None)
        total = torch.tensor([payload.size], dtype=torch.int64)
        sizes, cl, pl = shard.gather_to_root(dist, rank, world, total, torch.from_numpy(clen.view(np.int32).copy()),
                                             torch.from_numpy(np.concatenate([payload, np.zeros(8, np.uint8)])))
        if rank == 0:
            cont = shard.assemble_container(codec, n, chunk, cdfnum, [c.numpy().view(np.uint32) for c in cl], [p.numpy() for p in pl])
            full_payload, full_clen, _ = T.orc_chunked_enc(codec, data, chunk, cdf, cdfnum or 256)
            ref = shard.assemble_container(codec, n, chunk, cdfnum, [full_clen], [full_payload])
            ok = np.array_equal(cont, ref)
            nch = full_clen.size
            dec = T.orc_chunked_dec(codec, cont[32 + 4 * nch:], cont[32:32 + 4 * nch].view(np.uint32), n, chunk, cdf, cdfnum or 256)
            q.put(bool(ok and np.array_equal(dec, data)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("codec,n,chunk", [(T.ANS4S, 300001, 4096), (T.RCS2, 70000, 1024), (T.RCB, 20000, 4096), (T.ANS4S, 3000, 4096)])
def test_two_rank_shard_gather_matches_single(codec, n, chunk):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, codec, n, chunk, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _group_worker(rank, world, port, codec, n, chunk, nbatch, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        datas, totals, clens, payloads, cdfs = [], [], [], [], []
        for j in range(nbatch):                                # batch j: its own data, its own CDF
            data = T.zipf_bytes(n + 17 * j, 1.1, 256, 500 + j)
            start, ln = shard.shard_bounds(data.size, world, chunk)[rank]
            mine = data[start:start + ln]
            _, cdf, cdfnum = T.orc_cdfini(data)
            payload, clen, _ = T.orc_chunked_enc(codec, mine, chunk, cdf, cdfnum) if ln else (np.zeros(0, np.uint8), np.zeros(0, np.uint32), None)
            datas.append(data); cdfs.append((cdf, cdfnum))
            totals.append(torch.tensor([payload.size], dtype=torch.int64))
            clens.append(torch.from_numpy(clen.view(np.int32).copy()))
            payloads.append(torch.from_numpy(np.concatenate([payload, np.zeros(8, np.uint8)])))
        # pre-allocated receive buffers (what bench.py passes) for the batches rooted here, on odd worlds; allocation inside otherwise
        rc = rp = None
        if world % 2:
            mine = [j for j in range(nbatch) if j % world == rank]
            rc = {j: [torch.empty(4096, dtype=torch.int32) for _ in range(world - 1)] for j in mine}
            rp = {j: [torch.empty(n + 4096, dtype=torch.uint8) for _ in range(world - 1)] for j in mine}
        sizes, got = shard.exchange_group(dist, rank, world, totals, clens, payloads, rc, rp)
        assert sorted(got) == [j for j in range(nbatch) if j % world == rank]
        for j, (cl, pl) in got.items():
            data, (cdf, cdfnum) = datas[j], cdfs[j]
            cont = shard.assemble_container(codec, data.size, chunk, cdfnum, [c.numpy().view(np.uint32) for c in cl], [p.numpy() for p in pl])
            full_payload, full_clen, _ = T.orc_chunked_enc(codec, data, chunk, cdf, cdfnum)
            ref = shard.assemble_container(codec, data.size, chunk, cdfnum, [full_clen], [full_payload])
            assert [tuple(x) for x in sizes[j]] == [(int(p.numel()), int(c.numel())) for c, p in zip(cl, pl)]
            q.put((j, bool(np.array_equal(cont, ref))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,nbatch", [(2, 2), (3, 3), (3, 2), (3, 5), (2, 1), (8, 8), (8, 9)])
def test_group_exchange_with_rotating_roots(world, nbatch):
    """`exchange_group`: batch j of a group is gathered onto rank j % world, all transfers in one grouped call; every
    assembled container must equal the single-process container of that batch (full groups, a partial group -- the
    flush at the end of a run -- and a group longer than the world)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_group_worker, args=(r, world, port, T.ANS4S, 60001, 4096, nbatch, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(nbatch))
    assert res == [(j, True) for j in range(nbatch)]


def _pipeline_worker(rank, world, port, group, steps, prealloc, q, lag=1, staged=False):
    """bench.py's N>1 schedule (shard.StepPipeline) with CPU tensors: the oracle stands in for the HIP coder.
    staged: through shard.StagedDist (the adapter `bench.py --backend gloo` puts between the pipeline and gloo) and with the
    (total, chunks) rows of a bank in one tensor (`metas`), as bench.py keeps them since round 6"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    xdist = shard.StagedDist(dist, torch) if staged else dist
    try:
        codec, n, chunk = T.ANS4S, 40001, 1024
        start, ln = shard.shard_bounds(n, world, chunk)[rank]
        nch = (ln + chunk - 1) // chunk
        cap = ln + 64

        def step_data(k):
            return T.zipf_bytes(n, 1.1, 256, 700 + k)

        def new_result():
            return (torch.zeros(nch + 8, dtype=torch.int32), torch.zeros(cap, dtype=torch.uint8), torch.zeros(2, dtype=torch.int64))
        rotate = group > 1
        banks = [[new_result() for _ in range(group)] for _ in range(2)]
        metas = None
        if staged:
            metas = [torch.zeros(group, 2, dtype=torch.int64) for _ in range(2)]
            for b in range(2):
                metas[b][:, 1] = nch
                banks[b] = [(r[0], r[1], metas[b][j, 0:1]) for j, r in enumerate(banks[b])]
        recv = [None, None]
        if prealloc and (rotate or rank == 0):
            recv = [([torch.empty(64, dtype=torch.int32) for _ in range(world - 1)], [torch.empty(n + 64, dtype=torch.uint8) for _ in range(world - 1)])
                    for _ in range(2)]
        seen = []

        def on_gathered(first, sizes, got):
            for j, (cl, pl) in got.items():
                data = step_data(first + j)
                _, cdf, cdfnum = T.orc_cdfini(data)
                cont = shard.assemble_container(codec, n, chunk, cdfnum, [c.numpy().view(np.uint32) for c in cl], [p.numpy() for p in pl])
                fp, fc, _ = T.orc_chunked_enc(codec, data, chunk, cdf, cdfnum)
                ref = shard.assemble_container(codec, n, chunk, cdfnum, [fc], [fp])
                seen.append((first + j, bool(np.array_equal(cont, ref))))

        pipe = shard.StepPipeline(xdist, rank, world, group, banks, recv, nch, shard.HostRuntime(), rotate=rotate, on_gathered=on_gathered, lag=lag, metas=metas)
        cur = {}

        def encode(result):
            data = step_data(cur["k"])
            _, cdf, cdfnum = T.orc_cdfini(data)                 # (bench.py: histogram all-reduce; here every rank sees the whole input)
            payload, clen, _ = T.orc_chunked_enc(codec, data[start:start + ln], chunk, cdf, cdfnum)
            result[0][:nch] = torch.from_numpy(clen.view(np.int32).copy())
            result[1][:payload.size] = torch.from_numpy(payload)
            result[2][0] = payload.size

        def decode(result):
            pass

        for run in range(2):                                    # two runs back to back, as warmup + timed region
            for k in range(steps):
                cur["k"] = k
                pipe.step(k, k == steps - 1, encode, decode)
            pipe.reset()
        q.put((rank, seen))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,group,steps,prealloc,lag,staged", [(2, 2, 5, True, 1, False), (4, 4, 9, True, 1, False), (4, 4, 4, False, 4, False), (2, 1, 3, True, 1, False), (3, 3, 7, False, 2, False),
                                                                   (8, 8, 17, True, 1, False), (8, 8, 9, False, 4, False), (8, 8, 24, True, 8, False),
                                                                   (2, 2, 5, True, 1, True), (4, 4, 9, False, 2, True), (2, 1, 3, True, 1, True), (8, 8, 17, True, 1, True)])
def test_step_pipeline_schedule(world, group, steps, prealloc, lag, staged):
    """shard.StepPipeline -- the class bench.py --gpus N runs its steps through -- with CPU tensors over gloo: every step
    of two consecutive runs must arrive whole on its root (step j of a group on rank j; group 1: rank 0) and equal the
    single-process container of that step's data, incl. the partial last group and bank reuse.  World 8 (round 4) is the
    shape of the first real 8-GPU run: full groups, a group of one (9 steps), three full groups, exchanges issued 1, 4 and
    8 steps late."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, world, port, group, steps, prealloc, q, lag, staged)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    got = dict(q.get(timeout=5) for _ in range(world))
    for r in range(world):
        mine = [k for k in range(steps) if (k % group if group > 1 else 0) == r] if group > 1 or r == 0 else []
        assert sorted(got[r]) == sorted([(k, True) for k in mine] * 2), (r, got[r])


def test_group_plan_exchanges_every_step_once():
    """the schedule bench.py follows at N > 1: every step lands in exactly one exchange, groups fill slots 0..ns-1 of one
    bank, consecutive groups alternate banks, and only the last group of a run may be partial"""
    for group in range(1, 9):
        for steps in range(1, 41):
            seen, open_slots, groups = [], [], []
            for k in range(steps):
                j, bank, ns = shard.group_plan(k, group, k == steps - 1)
                assert j == len(open_slots) and (not open_slots or open_slots[-1][1] == bank)
                open_slots.append((k, bank))
                if ns:
                    assert ns == len(open_slots)
                    groups.append((bank, ns)); seen += [x[0] for x in open_slots]; open_slots = []
            assert not open_slots and seen == list(range(steps))
            assert all(a[0] != b[0] for a, b in zip(groups, groups[1:]))
            assert all(ns == group for _, ns in groups[:-1])


def test_shard_bounds_cover_everything():
    for n, world, chunk in [(100, 2, 64), (10**6, 8, 4096), (4096, 8, 4096), (1, 4, 256)]:
        b = shard.shard_bounds(n, world, chunk)
        assert sum(l for _, l in b) == n
        pos = 0
        for s, l in b:
            assert s == pos or l == 0
            assert s % chunk == 0 or l == 0
            pos += l
