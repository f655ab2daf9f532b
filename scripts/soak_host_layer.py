"""Soak of the host-pointer layer (GPU box): random calls through the reference-named functions -- coders, sizes from one byte to tens
of MB, text / skewed / run-heavy / incompressible / mixed data, pageable and page-locked buffers, one pipeline or a device list (the
same device several times), one caller or three at once -- every container compared with the oracle's per-chunk output, every decode
with the input.  usage: soak_host_layer.py [seconds] [seed]"""
import concurrent.futures as cf
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "turbo-range-coder_amd")]
import trc  # noqa: E402
import trc_testlib as T  # noqa: E402
from golden.make_golden import gen  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
lib = trc.lib()
lib.trc_host_pin.restype = C.c_int; lib.trc_host_pin.argtypes = [C.c_void_p, C.c_size_t]
lib.trc_host_unpin.restype = C.c_int; lib.trc_host_unpin.argtypes = [C.c_void_p]
CODECS = (trc.ANS4S, trc.RCS1, trc.RCS2, trc.RCA, trc.RCAI, trc.ANSA, trc.RCB, trc.ANSB, trc.RCV8, trc.ANSO1)
u8p = C.POINTER(C.c_uint8)


def make(n, kind, s):
    if kind == "mixed":
        d = gen("text", n, s)
        u = gen("uniform", n, s + 1)
        for lo in range(0, n, 300007):
            d[lo:lo + 40000] = u[lo:lo + 40000]
        return d
    return gen(kind, n, s)


def one_call(job):
    codec, n, kind, s, pin = job
    d = make(n, kind, s)
    cdf, cdfnum = None, 0
    if codec in trc.STATIC:
        if int(d.max()) == int(d.min()):
            return "skip"
        r, cdf, cdfnum = T.orc_cdfini(d)
        if r < 0:
            return "skip"
    out = np.zeros(n + n // 3 + 1024, dtype=np.uint8)
    back = np.full(n + 64, 0xA5, dtype=np.uint8)
    bufs = (d, out, back)
    if pin:
        for a in bufs:
            assert lib.trc_host_pin(a.ctypes.data, a.nbytes) == 0, lib.trc_last_error()
    try:
        enc = trc._host_fn(trc._HOST_ENC[codec], codec)
        dec = trc._host_fn(trc._HOST_DEC[codec], codec)
        ae = [d.ctypes.data_as(u8p), n, out.ctypes.data_as(u8p)]
        ad = [out.ctypes.data_as(u8p), n, back.ctypes.data_as(u8p)]
        if codec == trc.ANS4S:
            ae.append(cdf.ctypes.data_as(C.POINTER(C.c_uint16))); ad.append(ae[-1])
        elif codec in trc.STATIC:
            ae += [cdf.ctypes.data_as(C.POINTER(C.c_uint16)), cdfnum]; ad += ae[-2:]
        l = enc(*ae)
        assert l > 0 or n == 0, lib.trc_last_error()
        if l == n:
            assert np.array_equal(out[:n], d), "raw copy differs"
            return "raw"
        hdr, clen, payload = trc.parse_container(out[:l])
        assert hdr["n"] == n and hdr["codec"] == codec
        ep, ec = T.orc_chunked_enc_mt(codec, d, hdr["chunk"], cdf, cdfnum, threads=8)
        assert np.array_equal(clen, ec) and np.array_equal(payload, ep), ("container differs from the oracle", codec, n, kind, pin)
        k = dec(*ad)
        assert k == n and np.array_equal(back[:n], d), ("decode differs", codec, n, kind, pin)
        assert back[n] == 0xA5, "decoder wrote past its output"
        return "ok"
    finally:
        if pin:
            for a in bufs:
                lib.trc_host_unpin(a.ctypes.data)


t_end = time.time() + secs
counts = {}
rounds = 0
while time.time() < t_end:
    rounds += 1
    devs = [[], [], [0, 0], [0, 0, 0]][int(rng.integers(0, 4))]
    trc.set_devices(devs)
    nthreads = 1 if rng.random() < 0.5 else 3
    jobs = []
    for _ in range(6):
        codec = CODECS[int(rng.integers(0, len(CODECS)))]
        n = int(np.exp(rng.uniform(0, np.log(40e6))))
        kind = ("text", "zipf", "runs", "uniform", "mixed")[int(rng.integers(0, 5))]
        if codec == trc.RCV8 and kind == "uniform":
            kind = "text"
        jobs.append((codec, max(1, n), kind, int(rng.integers(0, 1 << 30)), bool(rng.random() < 0.4)))
    if nthreads == 1:
        res = [one_call(j) for j in jobs]
    else:
        with cf.ThreadPoolExecutor(nthreads) as ex:
            res = list(ex.map(one_call, jobs))
    for r in res:
        counts[r] = counts.get(r, 0) + 1
trc.set_devices([])
print("soak: %d rounds, %s in %.0f s (seed %d): all containers equal the oracle's, all decodes the input" % (rounds, counts, secs, seed))
