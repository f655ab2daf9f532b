#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: encode+decode MB/s, order-0 static-CDF rANS, 100 MB bytes.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without WORLD_SIZE: starts its own N ranks, see launch_command)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A "step" = one encode pass + one decode pass of the hot path over this rank's 100 MB shard, inputs
already resident in HBM (weak scaling: every rank codes its own 100 MB of independent chunks; for
N > 1 the per-rank compressed payloads of every step are gathered onto one GPU with RCCL inside the
timed region -- the path's only exchange step -- with the root rotating over the ranks and N steps per
grouped exchange, on a side stream, double-buffered, so that the transfers of one group overlap the
coding of the next; all K steps are gathered inside the timed region).
value = bytes all ranks processed / max-over-ranks wall time, with
MB = 10^6 (reference include_/time_.h:113,233), i.e. N / (t_enc + t_dec) per SURVEY 8d.

Workload (configs[1]): "text100m" -- 100 000 000 i.i.d. bytes from an English-like order-0 table,
the enwik8 stand-in of SURVEY 8d (no corpus, no network); set ENWIK8=/path/to/enwik8 to use the real
file (ENWIK8BWT=/path for the adaptive coders' run-heavy workload, BASELINE config 3).  Coder: static-CDF rANS, payload(chunk) bit-identical to anscdf4senc(chunk); CDF from cdfini
on device (untimed, as in the reference harness turborc.c:429-433).

Other workloads (each prints its own JSON line, same contract):
  --workload zipf1g   BASELINE config 5's per-GPU shard: 10^9 Zipf(1.1) bytes generated ON THE DEVICE (seed 1000+rank),
                      static rANS, sampled chunks checked against the oracle after the timed region
  --workload mix100m  heterogeneous stretches (text / incompressible / runs / constant, 1-64 KiB each): lanes of a wave
                      differ in rate, raw chunks sit among coded ones

Extra objects on the JSON line:
  roofline      dominant coder kernel: algorithmic bytes (N + C) per launch / mean launch duration
                measured with HIP events recorded around that kernel on its own stream, vs 8 TB/s HBM; the event pairs sit on
                every 4th timed step (--time-every: on every step they cost 4-5 % of the step; launches_timed says how many)
  flags         what the timed region leans on: TABLES_READY (coder tables derived once per CDF, untimed, as the
                reference harness builds its CDF untimed) and DIR_READY (the decode of a step reuses the directory sums
                its encode left in the workspace); value_cold = the same steps with both off
  clock_warmup  the untimed preamble (the same step for ~0.4 s) that precedes the W warm-up steps: a fresh process finds the GPU
                at idle clocks and 25 steps are over before it has ramped up; value_cold_clocks / ms_per_step_cold_clocks =
                the same W + K protocol measured BEFORE the preamble (--clock-warmup-ms 0: no preamble, value is that)
  payload_sha256  SHA-256 of the payload area the last timed step produced on rank 0 (the committed reference hash for
                the default configurations is in tests/golden/bench_configs.json)
  cpu_baseline  the reference (oracle/_ref) or, where absent, the oracle port, 1 thread, on a bounded
                sample of the same workload (rank 0, N = 1 only)
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "turbo-range-coder_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


# what each coder is, and the SURVEY 8d workload it is measured on (text100m: order-0 English-like bytes, cfg 2/4;
# drift100m: piecewise-stationary "BWT-like" bytes, cfg 3 -- round 4: whole-buffer rccdfenc 26.7 % as SURVEY 8d asks for
# (README.md:88,92), static 60 %; rounds 1-3 ran cfg 3 on bwt100m = stationary runs, where an adaptive model gains nothing:
# `--input bwt` still selects it; nib100m: run-heavy values 0..15 for the `turborc -n` coders)
CODEC_INFO = {
    "anscdf4s": ("static-CDF rANS, 2 states (anscdf4senc/anscdf4sdec per chunk), tables in LDS", "text"),
    "rccdfs":   ("static-CDF range coder (rccdfsenc/rccdfs*dec per chunk), tables in LDS", "text"),
    "rccdfsm":  ("static-CDF range coder, 32-bit range / 16-bit I/O (rccdfsmenc/rccdfsm*dec per chunk, `-e44`), tables in LDS", "text"),
    "rccdfs2":  ("static-CDF range coder, 2 interleaved streams (rccdfs2enc/rccdfs*2dec per chunk, `-e45`), tables in LDS", "text"),
    "rcs":      ("bitwise order-0 range coder (rcsenc/rcsdec per chunk), 512 B bit model per lane in LDS", "text"),
    "rccdf":    ("adaptive-CDF byte range coder (rccdfenc/rccdfdec per chunk), 544 B CDF16 model per lane in LDS", "drift"),
    "rccdfi":   ("adaptive-CDF byte range coder, 2 streams (rccdfienc/rccdfidec per chunk), 544 B CDF16 model per lane in LDS", "drift"),
    "anscdf":   ("adaptive-CDF byte rANS, 4 states (anscdfenc/anscdfdec per chunk), 544 B CDF16 model per lane in LDS", "drift"),
    "anscdf1":  ("order-1 adaptive-CDF byte rANS, 4 states (anscdf1enc/anscdf1dec per chunk), 136 KiB model per chunk in HBM", "drift"),
    "ansb":     ("bitwise order-0 rANS, 4 states (ansbc/ansbd per chunk), 512 B bit model per lane in LDS", "text"),
    "rccdfu16": ("Turbo-VLC (6-bit exponent) over the adaptive CDF range coder, 16-bit elements (rccdfuenc16/rccdfudec16 per chunk)", "i16"),
    "rccdfu32": ("Turbo-VLC (6-bit exponent) over the adaptive CDF range coder, 32-bit elements (rccdfuenc32/rccdfudec32 per chunk)", "i32"),
    "rccdfv16": ("Turbo-VLC (7-bit exponent) over the adaptive CDF range coder, 16-bit elements (rccdfvenc16/rccdfvdec16 per chunk)", "i16"),
    "rccdfv32": ("Turbo-VLC (7-bit exponent) over the adaptive CDF range coder, 32-bit elements (rccdfvenc32/rccdfvdec32 per chunk)", "i32"),
    "rccdfvz16": ("Turbo-VLC on zigzag deltas over the adaptive CDF range coder, 16-bit elements (rccdfvzenc16/rccdfvzdec16 per chunk)", "i16"),
    "rccdfvz32": ("Turbo-VLC on zigzag deltas over the adaptive CDF range coder, 32-bit elements (rccdfvzenc32/rccdfvzdec32 per chunk)", "i32"),
    "anscdfu16": ("Turbo-VLC (6-bit exponent) over the adaptive CDF rANS, 16-bit elements (anscdfuenc16/anscdfudec16 per chunk)", "i16"),
    "anscdfuz16": ("Turbo-VLC (6-bit exponent) on zigzag deltas over the adaptive CDF rANS, 16-bit elements (anscdfuzenc16/anscdfuzdec16 per chunk)", "i16"),
    "anscdfv16": ("Turbo-VLC (7-bit exponent) over the adaptive CDF rANS, 16-bit elements (anscdfvenc16/anscdfvdec16 per chunk)", "i16"),
    "anscdfvz16": ("Turbo-VLC (7-bit exponent) on zigzag deltas over the adaptive CDF rANS, 16-bit elements (anscdfvzenc16/anscdfvzdec16 per chunk)", "i16"),
    "anscdfv32": ("Turbo-VLC (7-bit exponent) over the adaptive CDF rANS, 32-bit elements (anscdfvenc32/anscdfvdec32 per chunk)", "i32"),
    "anscdfvz32": ("Turbo-VLC (7-bit exponent) on zigzag deltas over the adaptive CDF rANS, 32-bit elements (anscdfvzenc32/anscdfvzdec32 per chunk)", "i32"),
    "rccdf8":   ("variable-length ('vnibble') adaptive-CDF range coder, 1-3 CDF16 symbols per byte (rccdfenc8/rccdfdec8 per chunk, `-e48`)", "small"),
    "rccdfi8":  ("variable-length ('vnibble') adaptive-CDF range coder, 2 streams (rccdfienc8/rccdfidec8 per chunk, `-e49`)", "small"),
    "rccdf4":   ("adaptive-CDF nibble range coder (rccdf4enc/rccdf4dec per chunk), one CDF16 table per lane in LDS", "nib"),
    "rccdf4i":  ("adaptive-CDF nibble range coder, 2 streams (rccdf4ienc/rccdf4idec per chunk), one CDF16 table per lane in LDS", "nib"),
    "anscdf4":  ("adaptive-CDF nibble rANS, 2 states (anscdf4enc/anscdf4dec per chunk), one CDF16 table per lane in LDS", "nib"),
}


# chunk size with the best throughput at 100 MB per GPU where it is not 512: ONE residency round of the coder's waves.
# A launch's time is (rounds of resident waves) x (one wave's time, ~ chunk bytes), so the chunk that makes the input exactly
# one round costs the fewest per-round starts and codes best.  Static coders hold 12 waves per CU: 512 (3052 waves).
# rccdfs2 runs one lane per stream: chunk 1024 gives it the lanes chunk 512 gives the one-stream coder.  The coders with a
# model per lane in LDS (byte-adaptive, bitwise) hold 4 waves per CU = 1024 waves: chunk 1536 (1018 waves) -- rccdf 51.7 -> 66.6,
# anscdf 45.6 -> 56.2, ansb 26.6 -> 33.5, rcs 35.7 (at 768) -> 38.9 GB/s; 1280 (1221 waves, two rounds) is the worst point of
# the sweep at 35 GB/s (profiles/r03_notes.md section 7).  The order-1 coder needs room for its context statistics.
BEST_CHUNK = {"rccdfs2": 1024, "anscdf1": 4096, "rcs": 1536, "rccdf": 1536, "rccdfi": 1536, "anscdf": 1536, "ansb": 1536}
# Round 4: the table above is what the LIBRARY's rule gives at 100 MB -- trc_round_chunk(codec, n): the largest chunk
# (multiple of 64, <= 4096) that makes n a whole number of residency rounds of the coder's lanes -- and bench.py asks the
# library, for any --size (tests/test_gpu_chunk_policy.py sweeps 70 ... 333 MB); the table stays as the record of what was measured.


def make_input(n, rank, kind="text"):
    import trc_testlib as T
    path = os.environ.get("ENWIK8")
    if kind == "text" and path and os.path.exists(path):
        d = np.fromfile(path, dtype=np.uint8)
        reps = (n + d.size - 1) // d.size
        return np.tile(d, reps)[:n].copy(), "enwik8"
    if kind == "bwt":
        path = os.environ.get("ENWIK8BWT")                     # BASELINE config 3's corpus, where a box has it
        if path and os.path.exists(path):
            d = np.fromfile(path, dtype=np.uint8)
            return np.tile(d, (n + d.size - 1) // d.size)[:n].copy(), "enwik8bwt"
        return T.runs_bytes(n, 3 + rank), "bwt%dm" % (n // 1000000)
    if kind == "drift":
        path = os.environ.get("ENWIK8BWT")                     # BASELINE config 3's corpus, where a box has it
        if path and os.path.exists(path):
            d = np.fromfile(path, dtype=np.uint8)
            return np.tile(d, (n + d.size - 1) // d.size)[:n].copy(), "enwik8bwt"
        return T.drift_bytes(n, 3 + rank), "drift%dm" % (n // 1000000)
    if kind in ("i16", "i32"):                               # slow random walk: what the zigzag-delta coders are for
        return T.int_bytes(n, 2 if kind == "i16" else 4, "walk", 9 + rank), "walk%dm-%s" % (n // 1000000, kind)
    if kind == "small":
        return T.small_bytes(n, 15 + rank, "geo"), "small%dm" % (n // 1000000)
    if kind == "nib":
        return T.nibble_bytes(n, 5 + rank, "runs"), "nib%dm" % (n // 1000000)
    return T.text_bytes(n, 7 + rank), "text%dm" % (n // 1000000)


def cpu_baseline(codec, d, cdf, cdfnum, sample):
    """CPU timing of the same workload on this box's host cores.

    value      all host threads, shard-parallel: the buffer is cut into one shard per thread and every
               thread runs the whole-buffer reference call on its shard (ctypes releases the GIL)
    one_thread the reference's own regime (1 thread, whole-buffer call on the first `sample` bytes)
    encode / decode = the reference functions (oracle/_ref) when present, else the oracle port.  Two decoders are
    always the oracle port because the reference has none that round-trips: static rANS on a byte alphabet
     and the nibble rANS when the length is not a multiple of 4.  For the headline coder the literal
    `turborc -e45` pair rccdfs2enc/rccdfsb2dec (all reference) is listed for context."""
    import concurrent.futures as cf
    import trc_testlib as T
    use_ref = T.have_ref()
    if not use_ref:
        print("bench.py: oracle/_ref/libtrc_ref.so ABSENT (built from /root/reference in the build container only; it travels to the GPU box with "
              "the snapshot, a fresh clone does not have it): cpu_baseline is the oracle restatement, kind = \"port\"", file=sys.stderr)
    enc = (lambda x: T.ref_enc(codec, x, cdf, cdfnum)) if use_ref else (lambda x: T.orc_enc(codec, x, cdf, cdfnum))
    ref_dec_ok = use_ref and codec not in (T.ANS4S, T.ANSA4)
    dec_fn = (lambda c, n_: T.ref_dec(codec, c, n_, cdf, cdfnum)) if ref_dec_ok else (lambda c, n_: T.orc_dec(codec, c, n_, cdf, cdfnum))
    out = {"unit": "MB/s", "kind": "reference" if use_ref else "port"}

    s = np.ascontiguousarray(d[:sample])
    best_e = best_d = 1e9
    for _ in range(2):
        t0 = time.perf_counter(); comp = enc(s); t1 = time.perf_counter()
        dec = dec_fn(comp, s.size); t2 = time.perf_counter()
        best_e, best_d = min(best_e, t1 - t0), min(best_d, t2 - t1)
    assert np.array_equal(dec, s)
    one = {"cores": 1, "encdec_MBps": round(s.size / (best_e + best_d) / 1e6, 2), "enc_MBps": round(s.size / best_e / 1e6, 2),
           "dec_MBps": round(s.size / best_d / 1e6, 2), "sample_bytes": int(s.size)}
    if use_ref and codec == T.ANS4S:
        t0 = time.perf_counter(); c45 = T.ref_enc(T.RCS2, s, cdf, cdfnum); t1 = time.perf_counter()
        d45 = T.ref_dec(T.RCS2, c45, s.size, cdf, cdfnum); t2 = time.perf_counter()
        assert np.array_equal(d45, s)
        one["e45_reference_MBps"] = {"enc": round(s.size / (t1 - t0) / 1e6, 2), "dec": round(s.size / (t2 - t1) / 1e6, 2),
                                     "encdec": round(s.size / (t2 - t0) / 1e6, 2)}
    out["one_thread"] = one

    nthr = max(1, os.cpu_count() or 1)
    n = d.size
    per = (n + nthr - 1) // nthr
    shards = [np.ascontiguousarray(d[i:i + per]) for i in range(0, n, per)]
    with cf.ThreadPoolExecutor(nthr) as ex:
        list(ex.map(lambda x: x.sum(), shards))                        # spin the pool up
        t0 = time.perf_counter(); comps = list(ex.map(enc, shards)); t1 = time.perf_counter()
        decs = list(ex.map(lambda cx: dec_fn(cx[0], cx[1].size), zip(comps, shards))); t2 = time.perf_counter()
    assert all(np.array_equal(a, b) for a, b in zip(decs, shards))
    out["value"] = round(n / (t2 - t0) / 1e6, 1)
    out["enc_MBps"] = round(n / (t1 - t0) / 1e6, 1)
    out["dec_MBps"] = round(n / (t2 - t1) / 1e6, 1)
    out["cores"] = nthr
    name = T.CODEC_NAMES[codec]
    out["sample"] = ("whole workload (%d B) cut into %d shards, one whole-buffer call per thread on %d host threads; encode = %s "
                     "%senc, decode = %s %sdec; one_thread = first %d B, min of 2 runs"
                     % (n, len(shards), nthr, "reference" if use_ref else "oracle port", name,
                        "reference" if ref_dec_ok else "oracle port", name, s.size))
    return out


def beyond_cache_leg(torch, trc, T, codec, chunk, dev, steps=5, warmup=2, n=1000 * 1000 * 1000):
    """The default workload (100 MB in, 64.5 MB of payload) sits inside the 256 MiB Infinity Cache.  This leg repeats the
    step on BASELINE config 5's per-GPU shard -- 10^9 Zipf(1.1) bytes generated on the device, beyond any cache -- and
    reports it next to the headline: value, kernel times and the roofline fraction of its dominant kernel."""
    d_in = torch.zeros(n + 512, dtype=torch.uint8, device=dev)
    T.table_bytes_device(torch, dev, n, T.zipf_weights(1.1, 256), 1000, out=d_in)
    dc = trc.DeviceCoder(codec, n, chunk, dev)
    d_out = torch.zeros(n + 512, dtype=torch.uint8, device=dev)
    if codec in trc.STATIC:
        dc.cdfini(d_in, n, 256)
    for _ in range(warmup):
        dc.encode(d_in, n); dc.decode(d_out, n, dir_ready=True)
    torch.cuda.synchronize(dev)
    assert torch.equal(d_out[:n], d_in[:n]), "round trip failed (beyond-cache leg)"
    trc.timing_enable(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        dc.encode(d_in, n); dc.decode(d_out, n, dir_ready=True)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    enc_ms, enc_cnt = trc.timing_read(False)
    dec_ms, dec_cnt = trc.timing_read(True)
    trc.timing_enable(False)
    total_c = int(dc.total[0].item())
    e, d = enc_ms / max(enc_cnt, 1), dec_ms / max(dec_cnt, 1)
    dom = max(e, d)
    ach = (n + total_c) / (dom * 1e-3) / 1e9
    return {"workload": "zipf1000m: 10^9 Zipf(1.1) bytes generated on the device (seed 1000), chunk %d" % chunk,
            "value": round(n * steps / dt / 1e6, 1), "unit": "MB/s", "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps,
            "compressed_bytes": total_c, "enc_kernel_ms": round(e, 4), "dec_kernel_ms": round(d, 4),
            "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4)}


def host_pointer_leg(trc, T, rank, runs=7):
    """What a drop-in caller sees (SURVEY 8d / BASELINE.md 3: "end-to-end time including H2D/D2H through the host-pointer API"): the
    reference-named functions called with HOST pointers on 100 MB -- anscdf4senc/anscdf4sdec on text100m, rccdfenc/rccdfdec on
    drift100m -- chunk chosen by the library (trc_auto_chunk_codec), timed around the whole call like the reference's harness does
    (include_/time_.h:174-213, min over runs), with pageable buffers (what a malloc-ing caller has) and with page-locked ones
    (trc_host_pin: DMA straight from / to the caller's memory).  The caller is harness/trcbench -- plain C against include/*.h, the
    shape of the reference's bench() -- started on the workload written to a temporary file; without the binary, the same calls
    through ctypes from this process.  ratio_container = bytes the call returned / input bytes; ratio_reference_whole_buffer = ONE
    call of the reference function over the whole input (tests/golden/bench_configs.json)."""
    import ctypes as C
    import re
    import tempfile
    lib = trc.lib()
    n = 100 * 1000 * 1000
    whole = {}
    gold = os.path.join(ROOT, "tests", "golden", "bench_configs.json")
    if os.path.exists(gold):
        for e in json.load(open(gold)):
            if "whole_buffer_bytes" in e and e["n"] == n:
                whole[(e["codec"], e["kind"])] = e["whole_buffer_bytes"]
    exe = os.path.join(ROOT, "harness", "trcbench")
    use_c = os.path.exists(exe)
    out = {"bytes": n, "runs": runs, "caller": "harness/trcbench (plain C, include/turborc.h + include/anscdf.h)" if use_c else "ctypes from bench.py",
           "timed": "wall clock around the whole call, min over runs (PCIe both ways included); MB = 10^6"}
    u8p = C.POINTER(C.c_uint8)
    for name, codec, kind, hid in (("anscdf4senc", trc.ANS4S, "text", 65), ("rccdfenc", trc.RCA, "drift", 46)):
        d, wname = make_input(n, rank, kind)
        res = {"workload": wname, "chunk": int(lib.trc_auto_chunk_codec(codec, n))}
        if use_c:
            with tempfile.NamedTemporaryFile(suffix=".bin") as f:
                d.tofile(f); f.flush()
                for mode, flag in (("pageable", []), ("pinned", ["--pin"])):
                    r = subprocess.run([exe, "-I", str(runs), "-e", str(hid)] + flag + [f.name], capture_output=True, text=True, timeout=300)
                    m = re.search(r"^\s*(\d+)\s+[\d.]+%%\s+([\d.]+)\s+([\d.]+)\s+%d:" % hid, r.stdout, re.M)
                    if r.returncode != 0 or not m or "MISMATCH" in r.stdout:
                        raise RuntimeError("trcbench -e %d failed: %s" % (hid, (r.stdout + r.stderr)[-300:]))
                    l, e_, d_ = int(m.group(1)), float(m.group(2)), float(m.group(3))
                    res[mode] = {"enc_MBps": round(e_, 1), "dec_MBps": round(d_, 1), "encdec_MBps": round(1.0 / (1.0 / e_ + 1.0 / d_), 1)}
        else:
            lib.trc_host_pin.restype = C.c_int; lib.trc_host_pin.argtypes = [C.c_void_p, C.c_size_t]
            lib.trc_host_unpin.restype = C.c_int; lib.trc_host_unpin.argtypes = [C.c_void_p]
            cdf = None
            if codec in trc.STATIC:
                r, cdf, _ = trc.host_cdfini(d, 256)               # untimed, as in the reference harness (turborc.c:429-433)
                assert r == n
            enc = trc._host_fn(trc._HOST_ENC[codec], codec)
            dec = trc._host_fn(trc._HOST_DEC[codec], codec)
            comp = np.zeros(n + n // 3 + 1024, dtype=np.uint8)     # the harness's OSIZE (turborc.c:418)
            back = np.zeros(n + 1024, dtype=np.uint8)
            args_e = [d.ctypes.data_as(u8p), n, comp.ctypes.data_as(u8p)]
            args_d = [comp.ctypes.data_as(u8p), n, back.ctypes.data_as(u8p)]
            if codec == trc.ANS4S:
                args_e.append(cdf.ctypes.data_as(C.POINTER(C.c_uint16))); args_d.append(args_e[-1])
            for mode in ("pageable", "pinned"):
                if mode == "pinned":
                    for a in (d, comp, back):
                        assert lib.trc_host_pin(a.ctypes.data, a.nbytes) == 0, lib.trc_last_error()
                te = td = 1e9
                for _ in range(runs):
                    t0 = time.perf_counter(); l = enc(*args_e); t1 = time.perf_counter()
                    assert 0 < l < n, (name, l)
                    k = dec(*args_d); t2 = time.perf_counter()
                    assert k == n
                    te, td = min(te, t1 - t0), min(td, t2 - t1)
                assert np.array_equal(back[:n], d), name + ": round trip through the host-pointer calls failed"
                res[mode] = {"enc_MBps": round(n / te / 1e6, 1), "dec_MBps": round(n / td / 1e6, 1), "encdec_MBps": round(n / (te + td) / 1e6, 1)}
                if mode == "pinned":
                    for a in (d, comp, back):
                        lib.trc_host_unpin(a.ctypes.data)
        res["container_bytes"] = int(l)
        res["ratio_container"] = round(l / n, 5)
        w = whole.get((trc.CODEC_NAMES[codec], kind))
        res["ratio_reference_whole_buffer"] = round(w / n, 5) if w and "ENWIK8" not in os.environ and "ENWIK8BWT" not in os.environ else None
        out[name] = res
    return out


# BASELINE.json's other single-GPU configurations, so that the driver's BENCH record carries a driver-timed number for each of them
# (VERDICT r4 #7): config 3 = adaptive-CDF byte coders (rccdf = -e46 literal, anscdf = -e56), config 4 = rcs (-e1), and rccdfs2 =
# what configs 1 / 2 literally name (-e45).  Each is THIS script run once more in its own process on the coder's own workload and the
# library's chunk, short (10 timed steps after a 100 ms clock preamble), no CPU leg; the sub-object keeps the fields a reader needs.
# (anscdf1 -- SURVEY 8f rank 2, not a BASELINE configuration -- rides along since the end of round 5: the coder VERDICT r4 listed first under "weak".)
# Round 6: `anscdf4s_chunk4096` -- the headline coder at SURVEY 8d config 2's stated default chunk (24 414 chunks = 382 waves on 1 024
# SIMDs): the other end of the chunk trade-off, driver-timed next to the headline's chunk 512.
OTHER_CONFIGS = ("anscdf4s_chunk4096", "rccdfs2", "rccdf", "anscdf", "rcs", "anscdf1")


def other_configs():
    out = {}
    for name in OTHER_CONFIGS:
        cmd = [sys.executable, os.path.abspath(__file__), "--codec", name.split("_chunk")[0], "--steps", "10", "--warmup", "2", "--clock-warmup-ms", "100",
               "--no-cpu", "--no-beyond", "--no-cold", "--no-configs", "--no-host"] + (["--chunk", name.split("_chunk")[1]] if "_chunk" in name else [])
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=180)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
            j = json.loads(line)
            rf = j["roofline"]
            out[name] = {"metric": j["metric"], "value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"], "steps": j["steps"],
                         "workload": j["config"]["workload"], "chunk": j["config"]["chunk"],
                         "enc_kernel_ms": rf["enc_kernel_ms"], "dec_kernel_ms": rf["dec_kernel_ms"],
                         "roofline": {"kernel": rf["kernel"], "achieved": rf["achieved"], "peak": rf["peak"], "unit": rf["unit"], "frac": rf["frac"],
                                      "traffic": rf["traffic"], "traffic_source": rf["traffic_source"], "step_frac": rf.get("step_frac"), "enc_path": rf.get("enc_path")},
                         "ratio_container": j["config"]["ratio_container"], "ratio_reference_whole_buffer": j["config"]["ratio_reference_whole_buffer"],
                         "payload_matches_reference_sha256": j.get("payload_matches_reference_sha256"),
                         "wall_s": round(time.perf_counter() - t0, 1)}
        except Exception as e:                                 # a failed leg must not take the headline line with it
            out[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    return out


def launch_command(ngpus, argv, env=None):
    """`python bench.py --gpus N` started by hand (no WORLD_SIZE in the environment) starts its own N ranks: the command and
    the environment additions of that launch -- one process per GPU under torch.distributed.run, rendezvous on 127.0.0.1
    (the container hostname may not resolve), a free port unless MASTER_PORT is given."""
    import socket
    env = os.environ if env is None else env
    port = env.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus),
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + [a for a in argv if a != "--dry-launch"]
    add = {"HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),     # dmabuf IPC: RCCL needs it on this stack
           "OMP_NUM_THREADS": env.get("OMP_NUM_THREADS", "8")}
    return cmd, add


def memory_budget(codec_name, n, chunk, world, group):
    """HBM bytes one rank of `bench.py --gpus world` allocates (needs the library, not a GPU): what the first real 8-GPU
    run of BASELINE config 5 (1 GB shards) has to fit.  input + decoded output; the coder's workspace, directory and
    payload (trc.DeviceCoder); two banks of `group` result sets (clen, payload, total) for the exchange; on a root, two
    banks of world - 1 receive sets."""
    sys.path[:0] = [os.path.join(ROOT, "turbo-range-coder_amd")]
    import trc
    codec = {v: k for k, v in trc.CODEC_NAMES.items()}[codec_name]
    chunk = chunk or int(trc.lib().trc_round_chunk(codec, n))
    nch = (n + chunk - 1) // chunk
    result = 4 * (max(nch, 1) + 64) + (n + trc.PAD + 64) + 16
    b = {"input": n + 512, "output": n + 512, "workspace": int(trc.lib().trc_work_bytes(codec, n, chunk)) + trc.PAD,
         "result_banks": (2 * group if world > 1 else 1) * result,
         "receive_banks": 2 * (world - 1) * (4 * nch + n + 1024) if world > 1 else 0, "small": 4096}
    b["total"] = sum(b.values())
    b["chunk"] = chunk
    return b


def self_launch(args):
    import subprocess
    cmd, add = launch_command(args.gpus, sys.argv[1:])
    if args.dry_launch:
        n = args.size or (1000 * 1000 * 1000 if args.workload == "zipf1g" else 100 * 1000 * 1000)
        print(json.dumps({"launch": cmd, "env": add, "hbm_bytes_per_rank": memory_budget(args.codec, n, args.chunk, args.gpus, args.group or args.gpus)}))
        return 0
    env = dict(os.environ); env.update(add)
    # the ranks inherit stdout: rank 0 prints the one JSON line; torch.distributed.run ends non-zero if any rank dies
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=0, help="bytes per GPU (default 100 MB; zipf1g: 1 GB)")
    ap.add_argument("--chunk", type=int, default=int(os.environ.get("TRC_BENCH_CHUNK", "0")),
                    help="chunk bytes (parallel unit); default: the coder's throughput optimum at 100 MB per GPU -- 512 = 12 resident "
                         "waves per CU for the static rANS; payload ratio cost vs 4096: +1.6 %% (DESIGN.md)")
    ap.add_argument("--codec", default="anscdf4s")
    ap.add_argument("--workload", default="default", choices=["default", "zipf1g", "mix100m"])
    ap.add_argument("--input", default=None, choices=["text", "bwt", "drift", "i16", "i32", "small", "nib"],
                    help="default workload: the generator (default: the coder's own, CODEC_INFO)")
    ap.add_argument("--cpu-sample", type=int, default=32 * 1000 * 1000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-cold", action="store_true", help="skip the value_cold pass (flags off)")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and run the exchange code even with 1 rank (self-test)")
    ap.add_argument("--group", type=int, default=0, help="steps per exchange group (default: the world size with rotating roots; with --force-dist on one rank, "
                    "--group 8 runs the 8-GPU schedule -- banks of 8 steps, one size all-gather + one grouped send/receive per 8 steps -- with nothing to send: the schedule's own cost)")
    ap.add_argument("--lag", type=int, default=1, help="steps after a group's last step before its exchange is issued (shard.StepPipeline)")
    ap.add_argument("--clock-warmup-ms", type=float, default=400.0,
                    help="untimed preamble: the same step repeated for about this long before the W warm-up steps, so that the K "
                         "timed steps run at the GPU's steady-state clocks (0 = none; the JSON also carries the cold-clock line)")
    ap.add_argument("--time-every", type=int, default=4,
                    help="HIP event pairs on the coder kernels of every Nth timed step (the pairs cost ~1.8 %% of a step each way: "
                         "1 = every step, as before; 4 = steps 0, 4, 8, ... of the timed region)")
    ap.add_argument("--inflight", type=int, default=1,
                    help="N = 1 only: steps kept in flight on as many streams / contexts (default 1: kernels run alone, so their event "
                         "timings are their own; 2-3 hide the payload gather and the launch gaps behind the next step's coder: "
                         "+9 %% at chunk 512, +31 %% at chunk 1024, profiles/r02_notes.md -- with longer per-kernel times)")
    ap.add_argument("--no-beyond", action="store_true", help="default workload, N = 1: skip the beyond-cache leg (the same step on 1 GB generated on the device)")
    ap.add_argument("--dry-launch", action="store_true", help="with --gpus N > 1 and no WORLD_SIZE: print the launch (command, environment) as JSON and exit")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the exchange: nccl (= RCCL, the product path) or gloo with the device tensors staged through "
                         "host memory (shard.StagedDist): the device-side schedule against real peer processes where RCCL has no peer")
    ap.add_argument("--same-device", action="store_true", help="every rank on cuda:0 (with --backend gloo: N ranks rehearse the N-GPU schedule on one GPU)")
    ap.add_argument("--verify-gather", action="store_true",
                    help="N > 1: after the timed region, 2 more groups of steps with every root hashing what it received against the SHA-256 each owner computed of its own result")
    ap.add_argument("--no-host", action="store_true", help="default line, N = 1: skip the host_pointer leg (the reference-named calls on host buffers, PCIe included)")
    ap.add_argument("--no-configs", action="store_true", help="default line, N = 1: skip the `configs` sub-objects (BASELINE configs 3 / 4 and the -e45 literal, "
                    "each a short run of this script in its own process after the headline leg)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:       # started by hand: start the N ranks ourselves
        sys.exit(self_launch(args))
    if args.dry_launch:
        print(json.dumps({"launch": None, "env": {}}))
        return

    import hashlib
    import torch
    import shard
    import trc
    import trc_testlib as T

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and not (world == 1 and args.gpus <= 1):
        sys.exit("bench.py --gpus %d inside a job of WORLD_SIZE=%d: the two must agree" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs a GPU: libturborc_hip has no CPU path"
    if args.same_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as tdist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.backend == "gloo":
            tdist.init_process_group("gloo", rank=rank, world_size=world)
            dist = shard.StagedDist(tdist, torch)              # device tensors staged through host memory: a rehearsal transport
        else:
            if args.same_device and world > 1:
                sys.exit("--same-device with RCCL: one communicator cannot hold two ranks of one GPU; use --backend gloo")
            tdist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            dist = tdist

    codec = {v: k for k, v in trc.CODEC_NAMES.items()}[args.codec]
    n = args.size or (1000 * 1000 * 1000 if args.workload == "zipf1g" else 100 * 1000 * 1000)
    chunk = args.chunk or int(trc.lib().trc_round_chunk(codec, n))
    kind = args.input or CODEC_INFO[args.codec][1]
    d = None                                                   # host copy of the workload (None: it exists on the device only)
    if args.workload == "zipf1g":                              # BASELINE config 5: 1 GB of Zipf(1.1) per GPU, generated on the device
        seed = 1000 + rank
        d_in = torch.zeros(n + 512, dtype=torch.uint8, device=dev)
        T.table_bytes_device(torch, dev, n, T.zipf_weights(1.1, 256), seed, out=d_in)
        wname = "zipf%dm (Zipf(1.1) bytes generated on the device, seed %d)" % (n // 1000000, seed)
    else:
        if args.workload == "mix100m":
            d, wname = T.mix_bytes(n, 13 + rank), "mix%dm" % (n // 1000000)
        else:
            d, wname = make_input(n, rank, kind)
        d_in = torch.from_numpy(np.concatenate([d, np.zeros(512, np.uint8)])).to(dev)
    dc = trc.DeviceCoder(codec, n, chunk, dev)
    cdfnum = 256
    if codec in trc.STATIC:                                # untimed, like the reference harness (turborc.c:429-433)
        if use_dist:                                       # one CDF for the whole job: all-reduce the 256-bin histogram
            hist = torch.zeros(256, dtype=torch.int64, device=dev)
            dc.hist(d_in, n, hist)
            shard.allreduce_hist(dist, hist)
            dc.cdf_from_hist(hist, n * world, cdfnum)
        else:
            dc.cdfini(d_in, n, cdfnum)
        torch.cuda.synchronize(dev)
        assert int(dc.status[0].item()) > 0, "cdfini failed"
    d_out = torch.zeros(n + 512, dtype=torch.uint8, device=dev)

    # ---- multi-GPU exchange (the path's only exchange step: the compressed results are gathered over xGMI) ----------
    # shard.StepPipeline holds the schedule (groups of `world` steps, step j of a group gathered onto rank j, one grouped
    # send/receive call per group on a side stream, two banks of result buffers); tests/test_shard_gloo.py drives the
    # same class with CPU tensors.  TRC_BENCH_EXCHANGE=root0 selects the plain per-step gather to rank 0.
    nch = trc.nchunks(n, chunk)
    rotate = use_dist and os.environ.get("TRC_BENCH_EXCHANGE", "rotate") != "root0"
    G = (args.group if args.group > 0 and use_dist else world) if rotate else 1     # steps per exchange group
    DIRR = os.environ.get('TRC_NO_DIRR') is None               # ablation knob: TRC_NO_DIRR=1 re-derives the group sums in every decode

    def encode(result):
        dc.clen, dc.payload, dc.total = result
        dc.encode(d_in, n)

    def decode(result, dir_ready=None):
        dc.decode(d_out, n, dir_ready=DIRR if dir_ready is None else dir_ready)   # the encode just left this directory's group sums in the workspace

    # --inflight N (one GPU): N contexts (workspace, result buffers, output) on N streams, step k on context k mod N
    inflight = max(1, args.inflight) if not use_dist else 1
    extra = []
    if inflight > 1:
        for _ in range(inflight - 1):
            c2 = trc.DeviceCoder(codec, n, chunk, dev)
            if codec in trc.STATIC:
                c2.cdf.copy_(dc.cdf); c2.cdfnum = dc.cdfnum; c2._tables()
            extra.append((c2, torch.zeros(n + 512, dtype=torch.uint8, device=dev), torch.cuda.Stream(device=dev)))
        torch.cuda.synchronize(dev)

    own = (dc.clen, dc.payload, dc.total)
    pipe = None
    if use_dist:
        # (total, chunks) of every step of a bank sit in ONE tensor: the encoder writes a step's total into its row, the chunk count is
        # written once, and a group's exchange all-gathers the rows as they are (shard.exchange_group `meta`: no kernel of its own)
        metas = [torch.zeros(G, 2, dtype=torch.int64, device=dev) for _ in range(2)]
        for mt in metas:
            mt[:, 1] = nch

        def new_result(b, j):
            return (torch.zeros_like(dc.clen), torch.zeros_like(dc.payload), metas[b][j, 0:1])
        banks = [[new_result(b, j) for j in range(G)] for b in range(2)]
        own = banks[0][0]
        recv = [None, None]
        if world > 1 and (rotate or rank == 0):                # this rank is the root of one step per group
            recv = [([torch.empty(nch, dtype=torch.int32, device=dev) for _ in range(world - 1)],
                     [torch.empty(n + 1024, dtype=torch.uint8, device=dev) for _ in range(world - 1)]) for _ in range(2)]
        pipe = shard.StepPipeline(dist, rank, world, G, banks, recv, nch, shard.CudaRuntime(torch, dev), rotate=rotate, lag=args.lag, metas=metas)

    def step(k, last):
        if pipe is None:
            i = k % inflight
            if i == 0:
                encode(own)
                decode(own)
            else:
                c2, o2, s2 = extra[i - 1]
                with torch.cuda.stream(s2):
                    c2.encode(d_in, n)
                    c2.decode(o2, n, dir_ready=DIRR)
        else:
            pipe.step(k, last, encode, decode)

    if pipe is not None and G > 1:                             # untimed set-up: one full group, so that every pair of ranks has
        for k in range(G):                                     # its point-to-point connection before the warmup steps run
            step(k, k == G - 1)
        torch.cuda.synchronize(dev)
        pipe.reset()
    def run_untimed(nsteps):
        for k in range(nsteps):
            step(k, k == nsteps - 1)
        torch.cuda.synchronize(dev)
        if pipe is not None:
            pipe.reset()

    def wall_of(nsteps):
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        for k in range(nsteps):
            step(k, k == nsteps - 1)
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)
        d = time.perf_counter() - t
        if use_dist:
            tt = torch.tensor([d], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            d = float(tt.item())
        if pipe is not None:
            pipe.reset()
        return d

    # ---- the GPU's clocks.  A fresh process finds the GPU at idle clocks, and W + K = 25 steps (5 ms) are over before it has
    # ramped up: the same K steps take 0.19 ms each on a cold GPU and 0.17 ms after 0.2 s of work (profiles/r02_notes.md).  The
    # line's `value` is the steady state: W warm-up steps and K timed steps AFTER an untimed preamble of the same step;
    # `value_cold_clocks` is the same protocol without the preamble, measured first.
    run_untimed(args.warmup)
    if not args.no_verify:
        assert torch.equal(d_out[:n], d_in[:n]), "round trip failed"
    cold_clocks = None
    pre_steps = 0
    if args.clock_warmup_ms > 0:
        dtc = wall_of(args.steps)
        cold_clocks = (n * args.steps * world / dtc / 1e6, dtc / args.steps * 1e3)
        est = dtc / args.steps
        pre_steps = int(min(max(args.clock_warmup_ms / 1e3 / est, G), 20000))
        pre_steps = ((pre_steps + G - 1) // G) * G
        run_untimed(pre_steps)
        run_untimed(args.warmup)

    trc.timing_enable(not os.environ.get('TRC_BENCH_NO_KTIMING'))   # (probe: event pairs on the coder launches off)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    te = max(1, args.time_every)
    t0 = time.perf_counter()
    for k in range(args.steps):
        if te > 1:
            trc.timing_pause(k % te != 0)                      # event pairs on a sample of the timed steps
        step(k, k == args.steps - 1)
    torch.cuda.synchronize(dev)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    trc.timing_pause(False)
    enc_ms, enc_cnt = trc.timing_read(False)
    dec_ms, dec_cnt = trc.timing_read(True)
    gat_ms, gat_cnt = trc.timing_read(2)                       # the encode path's scan + gather kernels

    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- after the clock: the same steps with nothing carried over between calls (N = 1) ---------------------------
    cold = None
    if not args.no_verify:
        for c2, o2, s2 in extra:
            assert torch.equal(o2[:n], d_in[:n]), "round trip failed (second context)"
    if world == 1 and pipe is None and not args.no_cold:
        trc.timing_enable(True)                                # same launches as the timed region (event pairs on the coder kernels)
        ready, dc.tables_ready = dc.tables_ready, 0            # tables derived from the CDF inside every call
        ksteps = max(1, min(args.steps, 10))
        for _ in range(2):
            encode(own); decode(own, dir_ready=False)
        torch.cuda.synchronize(dev)
        c0 = time.perf_counter()
        for _ in range(ksteps):
            encode(own); decode(own, dir_ready=False)           # ... and the directory sums re-derived by the decode
        torch.cuda.synchronize(dev)
        cdt = time.perf_counter() - c0
        dc.tables_ready = ready
        cold = (n * ksteps / cdt / 1e6, cdt / ksteps * 1e3, ksteps)
    trc.timing_enable(False)

    # ---- after the clock (N > 1): every root hashes what it receives against the hash each owner computed of its own result ----
    gather_check = None
    if pipe is not None and world > 1 and args.verify_gather:
        torch.cuda.synchronize(dev)
        own_res = pipe.banks[0][0]
        tot0 = int(own_res[2][0].item())
        mine_sha = hashlib.sha256(own_res[0][:nch].cpu().numpy().tobytes() + own_res[1][:tot0].cpu().numpy().tobytes()).hexdigest()
        owners = [None] * world
        dist.all_gather_object(owners, mine_sha)
        seen = {"steps": 0, "pieces": 0, "bad": []}

        def on_gathered(first, sizes, got):
            torch.cuda.current_stream(dev).synchronize()       # (the hook runs on the side stream, behind the group's transfers)
            for j, (cl, pl) in got.items():
                seen["steps"] += 1
                for r in range(world):
                    h = hashlib.sha256(cl[r].cpu().numpy().tobytes() + pl[r].cpu().numpy().tobytes()).hexdigest()
                    seen["pieces"] += 1
                    if h != owners[r]:
                        seen["bad"].append((first + j, r))
        pipe.on_gathered = on_gathered
        vsteps = 2 * G + (1 if G > 1 else 0)                   # two full groups (both banks) and a ragged last one
        run_untimed(vsteps)
        pipe.on_gathered = None
        allseen = [None] * world
        dist.all_gather_object(allseen, seen)
        gather_check = {"steps_run": vsteps, "steps_checked_on_their_roots": sum(s["steps"] for s in allseen), "pieces_hashed": sum(s["pieces"] for s in allseen),
                        "mismatches": sum(len(s["bad"]) for s in allseen), "ok": all(not s["bad"] for s in allseen) and sum(s["steps"] for s in allseen) == vsteps}
        assert gather_check["ok"], "gathered pieces differ from their owners' results: %s" % allseen

    last_result = own if pipe is None else pipe.banks[shard.group_plan(args.steps - 1, G, True)[1]][shard.group_plan(args.steps - 1, G, True)[0]]
    total_c = int(last_result[2][0].item())
    sha = clen_sha = None
    checked = None
    if rank == 0:
        sha = hashlib.sha256(last_result[1][:total_c].cpu().numpy().tobytes()).hexdigest()
        clen_sha = hashlib.sha256(last_result[0][:nch].cpu().numpy().view(np.uint32).astype("<u4").tobytes()).hexdigest()
        if args.workload == "zipf1g" and not args.no_verify:
            # sampled chunks against the oracle: the slice is regenerated on the host from the same stream (which also
            # pins the device generator), coded by the oracle, compared with the bytes the last timed step produced
            clen64 = last_result[0][:nch].to(torch.int64) & 0xffffffff
            off = torch.cumsum(clen64, 0) - clen64
            cdf_h = np.zeros(257, dtype=np.uint16); cdf_h[:cdfnum + 1] = dc.cdf[:cdfnum + 1].cpu().numpy().view(np.uint16)
            picks = sorted(set([0, 1, 63, 64, nch // 2, nch - 65, nch - 1] + [int(x) for x in np.random.default_rng(9).integers(0, nch, 57)]))
            for c in picks:
                ln = min(chunk, n - c * chunk)
                sl = T.table_bytes_range(c * chunk, ln, T.zipf_weights(1.1, 256), 1000 + rank)
                assert np.array_equal(sl, d_in[c * chunk:c * chunk + ln].cpu().numpy()), "device generator differs from the host stream at chunk %d" % c
                exp = T.orc_enc(codec, sl, cdf_h, cdfnum)
                o, l = int(off[c].item()), int(clen64[c].item())
                assert l == exp.size and np.array_equal(last_result[1][o:o + l].cpu().numpy(), exp), "chunk %d differs from the oracle" % c
            checked = len(picks)

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = (n * world * args.steps) / dt / 1e6
        enc_avg = enc_ms / max(enc_cnt, 1)
        dec_avg = dec_ms / max(dec_cnt, 1)
        # dominant kernel = the slower of the two directions' coder kernels; algorithmic bytes = N + C per launch
        dom = "enc" if enc_avg >= dec_avg else "dec"
        dom_ms = max(enc_avg, dec_avg)
        gat_avg = gat_ms / max(gat_cnt, 1)
        alg_bytes = n + total_c
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        # HBM bytes per launch of the dominant kernel come from separate rocprofv3 --pmc passes of this same command
        # (scripts/gpu_profile_round.sh -> scripts/pmc_traffic.py -> profiles/pmc_traffic.json): counters cannot be read
        # from inside the timed process, so the line names where the figure was taken (`traffic_source`)
        traffic = traffic_source = gather_traffic = enc_traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                if n == 100 * 1000 * 1000 and args.workload == "default":   # the PMC passes were taken on the default workload
                    tj = json.load(open(tpath))
                    traffic = tj.get("%s_%s_chunk%d" % (args.codec, dom, chunk))
                    gather_traffic = tj.get("gather_chunk%d" % chunk) if args.codec == "anscdf4s" else None
                    enc_traffic = tj.get("%s_enc_chunk%d" % (args.codec, chunk))
                    if traffic is not None:
                        traffic_source = "profiles/pmc_traffic.json (%s)" % tj.get("source" if args.codec == "anscdf4s" else "source_cfg34",
                                                                                    "rocprofv3 --pmc passes of `python bench.py`, TCC_EA0_RDREQ/WRREQ by request size")
            except Exception:
                traffic = None
        default_metric = args.codec == "anscdf4s" and args.workload == "default" and n == 100 * 1000 * 1000
        res = {
            "metric": ("encode+decode MB/s, order-0 static-CDF rANS, 100 MB bytes" if default_metric
                       else "encode+decode MB/s, %s, %s, %d bytes per GPU" % (args.codec, wname.split(" ")[0], n)),
            "value": round(value, 1), "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": "%s: %d B/GPU, %s, chunk %d B, 1 lane = 1 chunk, 64 chunks/wave"
                                   % (wname, n, CODEC_INFO[args.codec][0], chunk),
                       "codec": args.codec, "chunk": chunk, "bytes_per_gpu": n, "compressed_bytes_per_gpu": total_c,
                       "ratio": round(total_c / n, 5),
                       # what the chunking costs (round 4): the container as stored (32 B header + 4 B of directory per chunk + payloads)
                       # next to ONE call of the reference function over the whole input (filled in below from the committed fixture)
                       "ratio_container": round((32 + 4 * ((n + chunk - 1) // chunk) + total_c) / n, 5), "ratio_reference_whole_buffer": None,
                       "steps_in_flight": inflight, "exchange_group_steps": G if use_dist else None, "exchange_lag_steps": (pipe.lag if pipe is not None else None),
                       "exchange_schedule": (None if not use_dist else "rotate" if rotate else "root0"), "exchange_backend": (None if not use_dist else "rccl" if args.backend == "nccl" else "gloo, device tensors staged through host memory (rehearsal of the schedule, not a transport)"),
                       "ranks_share_device": bool(args.same_device and world > 1),
                       "exchange": ("none" if world == 1 else "rccl gather of every step's payloads, root rotating over the ranks, %d steps per grouped exchange" % G if rotate else "rccl gather of payloads to rank 0")},
            "flags": (["TABLES_READY"] if codec in trc.STATIC else []) + (["DIR_READY"] if DIRR else []) +
                     (["CLOCK_WARMUP: `value` is measured after an untimed preamble of %d steps (%.0f ms) of the same step; value_cold_clocks is the W + K protocol without it" % (pre_steps, args.clock_warmup_ms)] if pre_steps else []),
            "value_cold": round(cold[0], 1) if cold else None,
            "ms_per_step_cold": round(cold[1], 4) if cold else None,
            "clock_warmup": {"ms": args.clock_warmup_ms, "steps": pre_steps,
                             "note": "untimed preamble of the same step before the W warm-up steps: the K timed steps run at steady-state clocks"},
            "value_cold_clocks": round(cold_clocks[0], 1) if cold_clocks else None,
            "ms_per_step_cold_clocks": round(cold_clocks[1], 4) if cold_clocks else None,
            "payload_sha256": sha, "clen_sha256": clen_sha,
            "enc_MBps": round(n / (enc_avg * 1e-3) / 1e6, 1) if enc_avg else None,
            "dec_MBps": round(n / (dec_avg * 1e-3) / 1e6, 1) if dec_avg else None,
            "roofline": {"bound": "hbm", "kernel": trc.lib().trc_kernel_name(codec, dom == "dec").decode(),
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "alg_bytes_per_launch": alg_bytes,
                         # round 6 (VERDICT r5 "next" 2b): the whole step and the encode DIRECTION, not only the slower coder kernel.  step_frac
                         # = 2 (N + C) / ms_per_step / peak (everything a step launches, launch gaps included); enc_path = (N + C) / (encoder
                         # kernels + scan + gather): the payload gather is part of producing the compressed bytes and is timed with its own
                         # event pairs (trc_timing_read class 2)
                         "step_frac": round(2 * alg_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "enc_path": {"achieved": round(alg_bytes / ((enc_avg + gat_avg) * 1e-3) / 1e9, 1) if enc_avg + gat_avg > 0 else None,
                                      "frac": round(alg_bytes / ((enc_avg + gat_avg) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if enc_avg + gat_avg > 0 else None,
                                      "gather_kernel_ms": round(gat_avg, 4), "gather_launches_timed": gat_cnt,
                                      "traffic": (enc_traffic + gather_traffic) if (enc_traffic and gather_traffic) else None,
                                      "gather_traffic": gather_traffic},
                         "enc_kernel_ms": round(enc_avg, 4), "dec_kernel_ms": round(dec_avg, 4), "launches_timed": enc_cnt,
                         "timed": "HIP event pairs on the coder-kernel launches of every %s timed step (two-pass encoders: both passes summed); directory and gather kernels are in ms_per_step only" % ("" if te == 1 else {2: "2nd", 3: "3rd"}.get(te, "%dth" % te))},
        }
        if checked is not None:
            res["oracle_checked_chunks"] = checked
        if gather_check is not None:
            res["gather_check"] = gather_check
        gold = os.path.join(ROOT, "tests", "golden", "bench_configs.json")
        if args.workload == "default" and os.path.exists(gold) and "ENWIK8" not in os.environ:
            for e in json.load(open(gold)):
                if e["codec"] == args.codec and e["kind"] == kind and e["n"] == n and e["seed"] == {"text": 7, "bwt": 3, "drift": 3}.get(kind, -1) + rank:
                    if "whole_buffer_bytes" in e:
                        res["config"]["ratio_reference_whole_buffer"] = round(e["whole_buffer_bytes"] / n, 5)
                    if e["chunk"] == chunk:
                        res["payload_matches_reference_sha256"] = bool(e["payload_sha256"] == sha and e["clen_sha256"] == clen_sha)
            if "payload_matches_reference_sha256" not in res:
                res["golden"] = "no committed reference hash for this coder / workload / chunk (tests/golden/bench_configs.json)"
        if kind == "drift":
            res["config"]["workload_since"] = "round 4 (rounds 1-3 quoted these coders on bwt100m at chunk 512: --input bwt --chunk 512); chunk from trc_round_chunk since round 3"
        if world == 1 and default_metric and not args.no_beyond and inflight == 1:
            del d_out
            torch.cuda.empty_cache()
            res["beyond_cache"] = beyond_cache_leg(torch, trc, T, codec, chunk, dev)
            res["roofline"]["frac_beyond_l3"] = res["beyond_cache"]["frac"]
            res["roofline"]["note_l3"] = "frac is measured on the 100 MB workload, which (with its 64.5 MB payload) fits the 256 MiB Infinity Cache; frac_beyond_l3 is the same kernel on 1 GB"
        if world == 1 and default_metric and not args.no_host and not args.no_cpu and not args.no_beyond and inflight == 1:
            try:                                               # (a failed leg must not take the headline line with it)
                res["host_pointer"] = host_pointer_leg(trc, T, rank)
            except Exception as e:
                res["host_pointer"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        if world == 1 and default_metric and not args.no_configs and not args.no_cpu and not args.no_beyond and inflight == 1:
            res["configs"] = other_configs()                   # (the full default line only: the measuring scripts pass --no-cpu / --no-beyond)
        if world == 1 and not args.no_cpu:
            if d is None:                                      # device-only workload: time the CPU on the first bytes of the same stream
                d = T.table_bytes_range(0, min(n, 100 * 1000 * 1000), T.zipf_weights(1.1, 256), 1000 + rank)
            cdf = dc.cdf[:cdfnum + 1].cpu().numpy().view(np.uint16).copy()
            cdf_full = np.zeros(257, dtype=np.uint16); cdf_full[:cdfnum + 1] = cdf
            res["cpu_baseline"] = cpu_baseline(codec, d, cdf_full, cdfnum, min(args.cpu_sample, d.size))
        print(json.dumps(res))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
