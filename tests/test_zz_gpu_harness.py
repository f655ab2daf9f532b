"""GPU (-m gpu), LAST in collection order: every test here starts a separate program -- the plain-C harnesses (trcbench,
trcfile, trcgather), the reference's own harness linked against the library, the unmodified reference tool.  They live in
their own file, named to sort behind every oracle / golden parity test, so that under `pytest -x` a harness problem can
never stop a parity test from running (round 2: a 300 s hang of the RCCL driver masked the file-format interop test).
Order inside the file: file-format interop (parity with the unmodified reference binary) first, RCCL drivers last."""
import os
import subprocess

import numpy as np
import pytest

import trc
import trc_testlib as T
from golden.make_golden import gen

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU (and must not silently fall back)"
    return torch


def host_chunk(codec, chunk):
    """chunk size the host-pointer layer really uses for the configured one (bitwise rANS: at most one reference block;
    order-1 coder: never below 4096)"""
    return min(chunk, 8192) if codec == trc.ANSB else max(chunk, 4096) if codec == trc.ANSO1 else chunk


def harness(name):
    exe = os.path.join(ROOT, "harness", name)
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "harness")])
    return exe


def test_reference_file_format_interop(torch_cuda, tmp_path):
    """the reference's own file container (hd_t / hdb_t, turborc.c:666-733, block loop :1044-1167) for file
    codec 1 (rcsenc per block).  With blocks that are legal chunk sizes a block IS a chunk, so
      * a file written by `trcfile C` (GPU, one launch for all blocks) is decompressed by the UNMODIFIED reference tool
        (oracle/_ref/turborc_ref, CPU), and is byte-identical to what the reference writes itself;
      * a file written by the reference (`turborc -01 -b65536B`) is decompressed by `trcfile D` on the GPU."""
    root = ROOT
    ref = os.path.join(root, "oracle", "_ref", "turborc_ref")
    exe = os.path.join(root, "harness", "trcfile")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/turborc_ref not built (scripts/link_reference_harness.sh --install, build container only)")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "harness")])
    for kind, n, bs in (("text", 3000001, 65536), ("zipf", 65536 * 3, 65536), ("uniform", 200000, 65536), ("runs", 100000, 4096), ("text", 70, 65536)):
        src, ours, theirs, back = tmp_path / "in.bin", tmp_path / "ours.rc", tmp_path / "theirs.rc", tmp_path / "back.bin"
        gen(kind, n, 77).tofile(src)
        r = subprocess.run([exe, "C", str(src), str(ours), str(bs)], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        r = subprocess.run([ref, "-01", "-b%dB" % bs, str(src), str(theirs)], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        assert open(ours, "rb").read() == open(theirs, "rb").read(), (kind, n, "files differ")
        r = subprocess.run([ref, "-d", str(ours), str(back)], capture_output=True, text=True, timeout=120)     # reference reads ours
        assert r.returncode == 0 and open(back, "rb").read() == open(src, "rb").read(), (kind, n, r.stdout + r.stderr)
        os.remove(back)
        r = subprocess.run([exe, "D", str(theirs), str(back)], capture_output=True, text=True, timeout=120)   # we read the reference's
        assert r.returncode == 0 and open(back, "rb").read() == open(src, "rb").read(), (kind, n, r.stdout + r.stderr)


def test_reference_harness_runs_on_the_gpu_library(torch_cuda, tmp_path):
    """oracle/_ref/turborc_hip is the REFERENCE's own harness (turborc.c bench(), turborc.c:420-579) and its non-hot
    objects linked, unchanged, against libturborc_hip.so by scripts/link_reference_harness.sh (build container only; the
    binary travels like the reference oracle build).  Its hot ids call cdfini / rccdfs2enc / anscdfenc / ... by the
    reference's names; its own memcheck (turborc.c:287-295) verifies every round trip, and the compressed size it prints
    must be the TRC1 container of per-chunk reference payloads."""
    import re
    root = ROOT
    exe = os.path.join(root, "oracle", "_ref", "turborc_hip")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/turborc_hip not built (scripts/link_reference_harness.sh --install, build container only)")
    n, chunk = 3000001, 1024
    env = dict(os.environ, TRC_CHUNK=str(chunk))
    ansi = re.compile(r"[\b]+")

    def rows(out):
        got = {}
        for line in ansi.sub(" ", out).splitlines():
            m = re.match(r"\s*(\d+)\s+([\d.]+)%.*?\s(\d+):\S", line)
            if m:
                got[int(m.group(3))] = int(m.group(1))
        return got

    d = gen("text", n, 33)
    src = tmp_path / "text.bin"
    d.tofile(src)
    _, cdf, _ = T.orc_cdfini(d, 256)                           # the harness: cdfini(in, n, cdf, 0x100), then cdfnum = m + 1 (turborc.c:429-433)
    m1 = int(d.max()) + 1
    r = subprocess.run([exe, "-I1", "-J1", "-e1,42,43,44,45,46,47,48,49,56,57,58,64,66", str(src)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "ERROR" not in r.stdout and "ERROR" not in r.stderr, r.stdout[-3000:] + r.stderr[-2000:]
    got = rows(r.stdout)
    ids = {1: trc.RCB, 42: trc.RCS1, 43: trc.RCS1, 44: trc.RCSM, 45: trc.RCS2, 46: trc.RCA, 47: trc.RCAI, 48: trc.RCV8, 49: trc.RCVI8,
           56: trc.ANSA, 57: trc.ANSA,
           58: trc.ANSA, 64: trc.ANSO1, 66: trc.ANSB}
    nch = trc.nchunks(n, chunk)
    for i, codec in ids.items():
        assert i in got, (i, r.stdout[-3000:])
        hc = host_chunk(codec, chunk)
        _, exp_clen, _ = T.orc_chunked_enc(codec, d, hc, cdf, m1)          # the harness passes cdfnum = max symbol + 1
        assert got[i] == 32 + 4 * trc.nchunks(n, hc) + int(exp_clen.sum()), (i, got[i])
    # `turborc -n`: values 0..15 -> the one-table coders and the static rANS id 65 (harness gate m<16)
    r = subprocess.run([exe, "-n", "-I1", "-J1", "-e42,45,46,47,56,65", str(src)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "ERROR" not in r.stdout and "ERROR" not in r.stderr, r.stdout[-3000:] + r.stderr[-2000:]
    got = rows(r.stdout)
    dn = (d & 15).astype(np.uint8)
    _, cdfn, _ = T.orc_cdfini(dn, 256)
    mn = int(dn.max()) + 1
    for i, codec in {42: trc.RCS1, 45: trc.RCS2, 46: trc.RCA4, 47: trc.RCAI4, 56: trc.ANSA4, 65: trc.ANS4S}.items():
        assert i in got, (i, r.stdout[-3000:])
        _, exp_clen, _ = T.orc_chunked_enc(codec, dn, chunk, cdfn, mn)
        assert got[i] == 32 + 4 * nch + int(exp_clen.sum()), (i, got[i])


def test_file_tool_roundtrips(torch_cuda, tmp_path):
    """harness/trcfile.c: compress / decompress files through the reference-named functions"""
    exe = harness("trcfile")
    src = tmp_path / "in.bin"
    for kind, n in (("text", 3000001), ("uniform", 200000), ("zipf", 1)):
        gen(kind, n, 21).tofile(src)
        for cid in (1, 42, 44, 45, 46, 47, 56, 64, 65, 66):
            packed, back = tmp_path / ("p%d" % cid), tmp_path / ("b%d" % cid)
            r = subprocess.run([exe, "c", str(cid), str(src), str(packed)], capture_output=True, text=True, timeout=120)
            assert r.returncode == 0, r.stdout + r.stderr
            r = subprocess.run([exe, "d", str(packed), str(back)], capture_output=True, text=True, timeout=120)
            assert r.returncode == 0, r.stdout + r.stderr
            assert np.array_equal(np.fromfile(back, dtype=np.uint8), np.fromfile(src, dtype=np.uint8)), (kind, cid)
            if kind == "text":
                assert os.path.getsize(packed) < 0.9 * n


def test_host_layer_slice_plan(torch_cuda, tmp_path):
    """the host-pointer layer cuts a call into slices (ramping up from 1/8 of the slice target and down again) and pipelines
    them over three streams: with a tiny slice target (many slices, ramps included) the files the file tool writes are
    byte-identical to the ones of the default plan (a few MB = one slice), and they round-trip"""
    exe = harness("trcfile")
    src = tmp_path / "in.bin"
    gen("text", 5000003, 33).tofile(src)
    for cid in (65, 45, 46, 1):
        outs = []
        for tag, env in (("one", {}), ("many", {"TRC_HOST_SLICE": "131072"}), ("flat", {"TRC_HOST_SLICE": "131072", "TRC_HOST_NO_RAMP": "1"})):
            packed, back = tmp_path / ("p%d%s" % (cid, tag)), tmp_path / ("b%d%s" % (cid, tag))
            e = dict(os.environ, **env)
            r = subprocess.run([exe, "c", str(cid), str(src), str(packed)], capture_output=True, text=True, timeout=120, env=e)
            assert r.returncode == 0, r.stdout + r.stderr
            r = subprocess.run([exe, "d", str(packed), str(back)], capture_output=True, text=True, timeout=120, env=e)
            assert r.returncode == 0, r.stdout + r.stderr
            assert np.array_equal(np.fromfile(back, dtype=np.uint8), np.fromfile(src, dtype=np.uint8)), (cid, tag)
            outs.append(np.fromfile(packed, dtype=np.uint8))
        assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2]), cid
    # slice targets below two groups of 64 chunks (TRC_HOST_SLICE < 2 * chunk * 64): the ramps are then as long as whole slices
    # and the plan must not take them unless they fit (round 2: 257..383 chunks made the plan's remainder wrap around)
    for nchunks in (257, 300, 383, 384, 450):
        src2 = tmp_path / ("in%d.bin" % nchunks)
        gen("text", nchunks * 1024 - 17, 35).tofile(src2)
        outs = []
        for tag, env in (("one", {}), ("tiny", {"TRC_HOST_SLICE": "65536"}), ("tinyflat", {"TRC_HOST_SLICE": "65536", "TRC_HOST_NO_RAMP": "1"})):
            packed, back = tmp_path / ("q%d%s" % (nchunks, tag)), tmp_path / ("r%d%s" % (nchunks, tag))
            e = dict(os.environ, TRC_CHUNK="1024", **env)
            r = subprocess.run([exe, "c", "65", str(src2), str(packed)], capture_output=True, text=True, timeout=60, env=e)
            assert r.returncode == 0, (nchunks, tag, r.stdout + r.stderr)
            r = subprocess.run([exe, "d", str(packed), str(back)], capture_output=True, text=True, timeout=60, env=e)
            assert r.returncode == 0, (nchunks, tag, r.stdout + r.stderr)
            assert np.array_equal(np.fromfile(back, dtype=np.uint8), np.fromfile(src2, dtype=np.uint8)), (nchunks, tag)
            outs.append(np.fromfile(packed, dtype=np.uint8))
        assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2]), nchunks


def test_c_harness_links_and_roundtrips(torch_cuda):
    """the plain-C TurboRC-style harness (harness/trcbench.c: only include/turborc.h + anscdf.h) round-trips
    every hot-path id through the reference-named functions with host pointers"""
    exe = harness("trcbench")
    for args in (["--zipf", "3000001"], ["--text", "1000000", "-c", "1024"], ["--uniform", "500000"], ["--nibble", "2000003"]):
        r = subprocess.run([exe, "-I", "1", "-e", "1,42,43,44,45,46,47,48,49,56,57,58,64,65,66,79"] + args, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "MISMATCH" not in r.stdout and "failed" not in r.stdout, r.stdout
        assert r.stdout.count(":") >= 17, r.stdout            # every requested id printed its row
        assert ("nibble" in r.stdout) == (args[0] == "--nibble")   # values 0..15 route ids 46/47/56-58 to the one-table coders
    # page-locked caller buffers (--pin: trc_host_pin): the host-pointer calls DMA straight from / to them, no staging copies
    r = subprocess.run([exe, "-I", "2", "--pin", "-e", "1,42,45,46,56,65,79", "--text", "20000003"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "MISMATCH" not in r.stdout and "failed" not in r.stdout and "page-locked" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([exe, "-I", "1", "--pin", "-e", "42,65", "--uniform", "3000001"], capture_output=True, text=True, timeout=300)   # raw return through pinned `out`
    assert r.returncode == 0 and "MISMATCH" not in r.stdout and "failed" not in r.stdout, r.stdout + r.stderr
    # adaptive / bitwise coders are sliced by one residency round (65 536 chunks = 32 MB at the automatic chunk 512): 150 MB is five
    # slices with the ramps at both ends, pageable (staged) and page-locked (direct)
    for pin in ([], ["--pin"]):
        r = subprocess.run([exe, "-I", "1", "-e", "1,46,66"] + pin + ["--text", "150000001"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "MISMATCH" not in r.stdout and "failed" not in r.stdout and r.stdout.count(":") >= 4, r.stdout + r.stderr
    for args in (["--int16", "2000000"], ["--int32", "4000000"]):        # integer series: the Turbo-VLC coders
        r = subprocess.run([exe, "-I", "1", "-e", "50,52,53,60,61,62,63"] + args, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "MISMATCH" not in r.stdout and "failed" not in r.stdout, r.stdout + r.stderr
        assert r.stdout.count("Turbo vlc") == (7 if args[0] == "--int16" else 5), r.stdout


# ---- the RCCL gather behind the C-ABI (trc_exchange_dev / trc_hist_allreduce_dev, harness/trcgather.c) ------------------

FAKE = os.path.join(ROOT, "tests", "libfake_rccl.so")


def run_gather(args, env=None, timeout=400):
    """harness/trcgather announces its phases on stderr and carries a per-phase watchdog (exit status 4 names the phase):
    a problem shows up as a failure with a trace within minutes, never as a silent hang"""
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    e.update(env or {})
    r = subprocess.run([harness("trcgather")] + args, capture_output=True, text=True, timeout=timeout, env=e)
    hashes = {}
    for line in r.stdout.splitlines():
        if line.startswith("batch "):
            w = line.split()
            hashes[int(w[1])] = (w[w.index("container") + 1], "[container verified]" in line)
    return r, hashes


@pytest.mark.parametrize("size,chunk", [(30000001, 512), (5000, 4096), (700, 256), (8000003, 2560)], ids=["30MB", "5000B", "700B", "8MB-world8"])
def test_c_gather_exchange_with_peers_on_one_gpu(torch_cuda, size, chunk):
    """trc_exchange_dev and trc_hist_allreduce_dev with world = 2, 3, 4 (and 5 for the ragged sizes: more ranks than chunks,
    empty shards) on ONE GPU: the ranks are processes sharing device 0 and the dozen RCCL calls are served by
    tests/fake_rccl.c (TRC_RCCL_LIB), which moves the bytes through host memory and FAILS on anything the real library
    would hang or corrupt on (unmatched or mis-sized send/receive pairs, sends left over in a group).  nbatch = 1 is the
    plain gather onto rank 0; nbatch = world the rotating roots (batch j onto rank j, every directed pair of ranks busy
    in one grouped call); nbatch = world + 1 wraps around.  Every root decodes the container it assembled and compares
    it with its batch's input; the container hashes must equal the single-process ones, batch by batch."""
    if not os.path.exists(FAKE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "harness"), "fake_rccl"])
    env = {"TRC_RCCL_LIB": FAKE, "TRC_FAKE_RCCL_TIMEOUT": "60"}
    base = ["--steps", "2", "--size", str(size), "--chunk", str(chunk), "--watchdog", "90"]
    # round 4: the shape of the first real 8-GPU run -- world 8, 8 batches (every rank a root once) and 9 (wraps around), the chunk
    # the library picks for BASELINE config 5's 1 GB shards (2560), shards of 1 MB
    eight = size == 8000003
    nref = 9 if eight else 5
    r, ref = run_gather(["--gpus", "1", "--batches", str(nref)] + base, env)
    assert r.returncode == 0 and len(ref) == nref and all(v for _, v in ref.values()), r.stdout + r.stderr
    for world in ((8,) if eight else (2, 3, 4) + ((5,) if size < 100000 else ())):
        for nb in ((8, 9) if eight else sorted({1, world, min(world + 1, 5)})):
            r, got = run_gather(["--gpus", str(world), "--batches", str(nb)] + base, env)
            assert r.returncode == 0 and "FAILED" not in r.stdout, (world, nb, r.stdout + r.stderr)
            assert sorted(got) == list(range(nb)), (world, nb, r.stdout + r.stderr)
            for j in range(nb):
                assert got[j][1] and got[j][0] == ref[j][0], (world, nb, j, got[j], ref[j], r.stderr[-2000:])


def test_fake_rccl_checks_what_it_claims(torch_cuda):
    """the checker itself (tests/fake_rccl_selftest.c, two ranks): matched transfers and collectives deliver the data; a
    mis-sized receive, a send nobody receives and a receive nobody sends to each end the group with an error inside the
    time limit -- schedules the real RCCL would hang or corrupt on cannot pass the tests above"""
    exe = os.path.join(ROOT, "tests", "fake_rccl_selftest")
    if not os.path.exists(exe) or not os.path.exists(FAKE):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "harness"), "fake_rccl"])
    for case in range(4):
        r = subprocess.run([exe, FAKE, str(case)], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and "ok" in r.stdout, (case, r.stdout + r.stderr)


def test_c_gather_driver_single_gpu(torch_cuda):
    """harness/trcgather.c over the REAL RCCL with one rank (what a one-GPU box can run of it): library load, unique id,
    communicator, histogram all-reduce, size all-gather, the root's own piece, whole-container decode.  The last test of
    the suite on purpose."""
    for args in (["--size", "30000001", "--chunk", "512"], ["--size", "5000", "--chunk", "4096"]):
        r, got = run_gather(["--gpus", "1", "--steps", "2", "--watchdog", "60"] + args, {"NCCL_DEBUG": "WARN"})
        assert r.returncode == 0 and got and got[0][1] and "FAILED" not in r.stdout, r.stdout + r.stderr
