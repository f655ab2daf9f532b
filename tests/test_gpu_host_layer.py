"""GPU (-m gpu): the host-pointer layer of round 6 -- the reference-named calls as a drop-in caller sees them.

  * the automatic chunk is chosen for the stored size (trc_auto_chunk_codec): at 100 MB the containers of `rccdfenc`,
    `anscdfenc` and `anscdf4senc` carry the per-chunk payloads of chunk 4096 -- hashed against the reference's
    (tests/golden/bench_configs.json) -- and stay within the stated bound of ONE whole-buffer call of the reference;
  * slices of a call are coded concurrently on several streams: every stream count gives the same bytes;
  * a call spread over several pipelines (trc_set_devices; the same device listed two and three times on a one-GPU box)
    returns the container of the one-device call, byte for byte: ragged shards, empty shards, raw chunks, raw calls.
"""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import trc
import trc_testlib as T
from golden.make_golden import gen

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
with open(os.path.join(GOLD, "bench_configs.json")) as _f:
    BENCH_GOLD = {e["name"]: e for e in json.load(_f)}


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU (and must not silently fall back)"
    return torch


# (golden entry, bound on container / input in percent -- VERDICT r5 "next" 1: rccdfenc on drift100m <= 28.5 % where one
# whole-buffer call of the reference stores 26.66 % and round 5's chunk 512 stored 38.7 %; anscdf4senc on text100m <= 63.7 %)
RATIO_CASES = [("rccdf-drift100m-4096", 28.5), ("anscdf-drift100m-4096", 28.5), ("anscdf4s-text100m-4096", 63.7)]


@pytest.mark.parametrize("name,bound", RATIO_CASES, ids=[c[0] for c in RATIO_CASES])
def test_host_layer_ratio_bound_and_total_parity(torch_cuda, name, bound):
    cfg = [c for c in T.BENCH_CONFIGS if c["name"] == name][0]
    g = BENCH_GOLD[name]
    n, codec = cfg["n"], cfg["codec"]
    d = T.bench_input(cfg["kind"], n, cfg["seed"])
    assert hashlib.sha256(d.tobytes()).hexdigest() == g["in_sha256"], "workload generator drifted"
    cdf, cdfnum = None, 0
    if codec in trc.STATIC:
        r, cdf, cdfnum = trc.host_cdfini(d, 256)
        assert r == n and hashlib.sha256(cdf[:257].tobytes()).hexdigest() == g["cdf_sha256"]
    assert trc.lib().trc_get_chunk() == 0, "the chunk must be automatic for this test"
    comp = trc.host_encode(codec, d, cdf, cdfnum)
    hdr, clen, payload = trc.parse_container(comp)
    assert hdr["chunk"] == 4096 == trc.lib().trc_auto_chunk_codec(codec, n)
    assert hashlib.sha256(clen.astype("<u4").tobytes()).hexdigest() == g["clen_sha256"], "length directory differs from the reference"
    assert hashlib.sha256(payload.tobytes()).hexdigest() == g["payload_sha256"], "payload differs from the reference"
    assert 100.0 * comp.size / n <= bound, (comp.size, bound)
    assert comp.size <= 1.06 * g["whole_buffer_bytes"], (comp.size, g["whole_buffer_bytes"])      # within 6 % of ONE reference call over the whole input
    assert np.array_equal(trc.host_decode(codec, comp, n, cdf, cdfnum), d)


def _child(code, env):
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env), cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return r.stdout


CHILD = r"""
import sys, hashlib
sys.path.insert(0, "tests"); sys.path.insert(0, "turbo-range-coder_amd")
import numpy as np, trc, trc_testlib as T
from golden.make_golden import gen
out = []
for codec, kind, n in ((trc.ANS4S, "text", 40000003), (trc.RCA, "text", 30000001), (trc.RCB, "zipf", 20000001), (trc.RCS2, "text", 50000017), (trc.ANSA, "zipf", 9000001)):
    d = gen(kind, n, 41)
    _, cdf, cdfnum = T.orc_cdfini(d)
    comp = trc.host_encode(codec, d, cdf, cdfnum)
    assert np.array_equal(trc.host_decode(codec, comp, n, cdf, cdfnum), d)
    out.append(hashlib.sha256(comp.tobytes()).hexdigest())
print(" ".join(out))
"""


def test_host_layer_stream_counts_agree(torch_cuda):
    """1, 2, 3 and 8 coder streams (TRC_HOST_STREAMS is read once per process: children) and a forced small slice: same containers"""
    ref = _child(CHILD, {"TRC_HOST_STREAMS": "1"}).split()
    assert len(ref) == 5
    for env in ({"TRC_HOST_STREAMS": "2"}, {"TRC_HOST_STREAMS": "3"}, {"TRC_HOST_STREAMS": "8"},
                {"TRC_HOST_STREAMS": "4", "TRC_HOST_SLICE": "1048576"}, {"TRC_HOST_STREAMS": "8", "TRC_HOST_SLICE": "524288", "TRC_HOST_NO_RAMP": "1"}):
        assert _child(CHILD, env).split() == ref, env


def mixed(n, seed):
    """compressible text with incompressible stretches: raw chunks inside a coded container"""
    d = gen("text", n, seed)
    u = gen("uniform", n, seed + 1)
    for lo in range(0, n, 700001):
        d[lo:lo + 90000] = u[lo:lo + 90000]
    return d


@pytest.mark.parametrize("codec", [trc.ANS4S, trc.RCS2, trc.RCA, trc.ANSA, trc.RCB], ids=lambda c: trc.CODEC_NAMES[c])
def test_host_layer_multi_device(torch_cuda, codec):
    chunk = 1024
    assert trc.lib().trc_set_chunk(chunk) == 0
    try:
        cases = [("text", 3000001), ("mixed", 5000003), ("text", 70001), ("text", 64 * 1024 * 5), ("uniform", 300000), ("zipf", 40)]
        single = []
        for kind, n in cases:
            d = mixed(n, 51) if kind == "mixed" else gen(kind, n, 51)
            _, cdf, cdfnum = T.orc_cdfini(d)
            comp = trc.host_encode(codec, d, cdf, cdfnum)
            single.append((d, cdf, cdfnum, comp))
        assert any(comp.size == d.size and d.size > 100 for d, _, _, comp in single), "the uniform case must come back raw"
        _, clen, _ = trc.parse_container(single[1][3])
        lens = np.minimum(np.full(clen.size, chunk), 5000003 - np.arange(clen.size) * chunk)
        assert 0 < int((clen == lens).sum()) < clen.size, "the mixed case must hold raw and coded chunks"
        for devs in ([0, 0], [0, 0, 0], [0, 0, 0, 0, 0]):
            trc.set_devices(devs)
            for (d, cdf, cdfnum, comp) in single:
                if codec in trc.STATIC:                    # cdfini over the list: per-pipeline histograms summed on the host -> the same CDF
                    r, cdf2, _ = trc.host_cdfini(d, cdfnum)
                    assert r == d.size and np.array_equal(cdf2[:cdfnum + 1], cdf[:cdfnum + 1]), (devs, d.size, "cdfini over the device list")
                multi = trc.host_encode(codec, d, cdf, cdfnum)
                assert multi.size == comp.size and np.array_equal(multi, comp), (devs, d.size, "container differs from the one-device call")
                if comp.size != d.size:
                    assert np.array_equal(trc.host_decode(codec, comp, d.size, cdf, cdfnum), d), (devs, d.size, "decode over the device list")
            trc.set_devices([])
    finally:
        trc.set_devices([])
        trc.lib().trc_set_chunk(0)


def test_device_list_from_the_environment(torch_cuda):
    """TRC_DEVICES=0,0,0 / all: read at the first host-pointer call; a bad list is ignored with a message"""
    code = CHILD + "\nimport ctypes\nprint(trc.lib().trc_get_devices(None, 0))"
    one = _child(code, {}).split()
    three = _child(code, {"TRC_DEVICES": "0,0,0"}).split()
    every = _child(code, {"TRC_DEVICES": "all"}).split()
    assert one[:-1] == three[:-1] == every[:-1]
    assert one[-1] == "0" and three[-1] == "3" and int(every[-1]) >= 1
    bad = _child(code, {"TRC_DEVICES": "0,x"}).split()
    assert bad[:-1] == one[:-1] and bad[-1] == "0"


STRIPED = r"""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "turbo-range-coder_amd")
import numpy as np, trc, trc_testlib as T
from golden.make_golden import gen
import ctypes
lib = trc.lib()
lib.trc_host_pin.restype = ctypes.c_int; lib.trc_host_pin.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
lib.trc_host_unpin.restype = ctypes.c_int; lib.trc_host_unpin.argtypes = [ctypes.c_void_p]
rng = np.random.default_rng(5)
n_calls = 0
for codec in (trc.RCA, trc.RCAI, trc.ANSA, trc.RCB):
    for it in range(7):
        # the same staging / device buffers call after call, different bytes every time: anything read before its pass has landed, or
        # left in a cache by the call before, shows as a payload that differs from the oracle's
        n = int(rng.integers(1, 6 * 1000 * 1000)) if it else 4096 * 517
        if it == 1: n = 4096 * 64 * 3 + 1
        if it == 2: n = 4095
        d = gen(("text", "zipf", "runs")[it % 3], n, 900 + 31 * it + codec)
        if it == 4:
            u = gen("uniform", n, 77); d[n // 3:n // 3 + 50000] = u[n // 3:n // 3 + 50000]      # raw chunks among coded ones
        pin = it != 5                                      # striped input needs a page-locked source (every copy queued before the waiting kernel)
        if pin: assert lib.trc_host_pin(d.ctypes.data, d.nbytes) == 0
        comp = trc.host_encode(codec, d)
        if pin: lib.trc_host_unpin(d.ctypes.data)
        if comp.size == n:
            assert np.array_equal(comp, d); continue
        hdr, clen, payload = trc.parse_container(comp)
        chunk = hdr["chunk"]
        assert chunk == trc.lib().trc_auto_chunk_codec(codec, n) and chunk >= 2048
        exp_payload, exp_clen = T.orc_chunked_enc_mt(codec, d, chunk, None, 0)
        assert np.array_equal(clen, exp_clen) and np.array_equal(payload, exp_payload), (codec, it, n)
        assert np.array_equal(trc.host_decode(codec, comp, n), d)
        n_calls += 1
print("ok", n_calls)
"""


def test_striped_encode_matches_the_oracle_call_after_call(torch_cuda):
    """round 6: the encoders of rccdf / rccdfi / anscdf / rcs are launched BEFORE their input has arrived and wait at an arrival gate
    for each of its K passes (trc_io.h, trc_host.inc).  28 calls on the same buffers, sizes from one chunk to 6 MB, ragged ends, raw
    chunks: every payload equals the oracle's per-chunk output; the same through the slice pipeline (TRC_HOST_NO_STRIPE) and over
    a device list.  The input is page-locked (striping needs every copy queued before the waiting kernel); one call in seven and every
    output buffer are pageable (streamed decodes scatter through the staging slots)."""
    for env in ({}, {"TRC_HOST_NO_STRIPE": "1"}, {"TRC_DEVICES": "0,0"}, {"TRC_HOST_PIECE": "262144"}):
        out = _child(STRIPED, env).split()
        assert out[0] == "ok" and int(out[1]) >= 20, (env, out)


def test_a_gate_that_never_opens_fails_over_to_the_slice_pipeline(torch_cuda):
    """TRC_HOST_GATE_SABOTAGE (test hook): the last pass of every striped slice never opens its gate.  The waiting waves give up after
    about a second and say so (word 63 of the gate area), the host repeats the call through the slice pipeline and stops using gates in
    this process: the caller gets the right container, late, and one line on stderr -- no trap, no hang"""
    code = r'''
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, "turbo-range-coder_amd")
import numpy as np, trc, trc_testlib as T
from golden.make_golden import gen
import ctypes
d = gen("text", 3000001, 5)
trc.lib().trc_host_pin.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert trc.lib().trc_host_pin(d.ctypes.data, d.nbytes) == 0
t0 = time.time(); a = trc.host_encode(trc.RCA, d); t1 = time.time(); b = trc.host_encode(trc.RCA, d); t2 = time.time()
hdr, clen, payload = trc.parse_container(a)
ep, ec = T.orc_chunked_enc_mt(trc.RCA, d, hdr["chunk"], None, 0)
assert np.array_equal(clen, ec) and np.array_equal(payload, ep) and np.array_equal(a, b)
assert np.array_equal(trc.host_decode(trc.RCA, a, d.size), d)
print("ok %.2f %.2f" % (t1 - t0, t2 - t1))
'''
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, TRC_HOST_GATE_SABOTAGE="1"), cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.split()[0] == "ok" and "an arrival gate timed out" in r.stderr, r.stdout + r.stderr[-2000:]
    first, second = float(r.stdout.split()[1]), float(r.stdout.split()[2])
    assert first > 0.3 and second < 0.3, (first, second)               # the first call waited for the timeout, the second one no longer uses gates


def test_soak_of_the_host_layer(torch_cuda):
    """scripts/soak_host_layer.py for half a minute: random calls (nine coders, 1 B ... 40 MB, five kinds of data, pageable and page-locked
    buffers, one pipeline or a device list, one caller or three at once), every container against the oracle, every decode against the
    input -- and no arrival gate may time out on the way (the first soak of round 6 found a copy stuck behind a waiting kernel)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "soak_host_layer.py"), "30", "3"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "all containers equal the oracle's" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    assert "timed out" not in r.stderr, r.stderr[-2000:]
