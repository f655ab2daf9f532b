"""GPU (-m gpu): the HIP path through the C-ABI against the CPU oracle, bit-exact.

  * device layer (trc_encode_dev / trc_decode_dev): per-chunk payloads and lengths == oracle,
    decode round trip, ragged / tiny / incompressible inputs, several chunk sizes;
  * golden vectors from the reference (single-chunk containers == reference whole-buffer output);
  * host-pointer layer (reference prototypes): container parse, round trip, raw convention;
  * BASELINE-size property tests (100 MB): round trip, directory consistency, sampled chunks
    bit-exact against the oracle.
"""
import json
import os

import numpy as np
import pytest

import trc
import trc_testlib as T
from golden.make_golden import gen

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU (and must not silently fall back)"
    return torch


def fit(codec, d):
    """the `turborc -n` coders take values 0..15 (harness gate m<16): fold any test input into that range; the Turbo-VLC
    coders take 16/32-bit elements: every input byte becomes one (small) element, leftover bytes stay as they are"""
    if codec in trc.NIBBLE_CODECS:
        return (d & 15).astype(np.uint8)
    if codec in trc.VLC_CODECS:
        es = trc.VLC_ELEM[codec]
        ne = d.size // es
        body = d[:ne].astype(np.uint16 if es == 2 else np.uint32).view(np.uint8)
        return np.concatenate([body, d[:d.size - ne * es]]).astype(np.uint8)
    return d


SMALL_ALPHABET = trc.NIBBLE_CODECS + trc.VLC_CODECS          # inputs that always compress: nothing is stored raw


def cap(codec, chunk):
    """the bitwise rANS takes chunks of at most one reference block (8192 bytes)"""
    return min(chunk, 8192) if codec == trc.ANSB else chunk


def host_chunk(codec, chunk):
    """chunk size the host-pointer layer really uses for the configured one: the bitwise rANS is capped at one reference
    block, the order-1 coder never goes below 4096 (its 256 x 17 tables need data to learn from)"""
    return min(chunk, 8192) if codec == trc.ANSB else max(chunk, 4096) if codec == trc.ANSO1 else chunk


def to_dev(torch, a):
    return torch.from_numpy(np.concatenate([a, np.zeros(512, np.uint8)])).to("cuda:0")


def device_roundtrip(torch, codec, d, chunk, cdf, cdfnum):
    n = d.size
    dc = trc.DeviceCoder(codec, n, chunk, "cuda:0")
    if codec in trc.STATIC:
        dc.set_cdf(cdf, cdfnum)
    d_in = to_dev(torch, d)
    dc.encode(d_in, n)
    clen, payload = dc.result(n)
    exp_payload, exp_clen, _ = T.orc_chunked_enc(codec, d, chunk, cdf, cdfnum)
    assert np.array_equal(clen, exp_clen), "clen mismatch"
    assert payload.size == exp_payload.size and np.array_equal(payload, exp_payload), "payload mismatch"
    d_out = torch.full((n + 512,), 0xA5, dtype=torch.uint8, device="cuda:0")
    dc.decode(d_out, n)
    torch.cuda.synchronize()
    out = d_out.cpu().numpy()
    assert np.array_equal(out[:n], d), "decode mismatch"
    assert (out[n:] == 0xA5).all(), "decoder wrote past the end"
    # TRC_DIR_READY: the group sums the decode above left in the workspace serve a second decode of the same directory,
    # and the ones the encoder leaves serve the decode that follows it
    d_out.fill_(0x5A)
    dc.decode(d_out, n, dir_ready=True)
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy()[:n], d), "decode with TRC_DIR_READY (after a decode) mismatch"
    dc.encode(d_in, n)
    d_out.fill_(0x5A)
    dc.decode(d_out, n, dir_ready=True)
    torch.cuda.synchronize()
    out = d_out.cpu().numpy()
    assert np.array_equal(out[:n], d) and (out[n:] == 0x5A).all(), "decode with TRC_DIR_READY (after an encode) mismatch"
    # the decode-only contract: a workspace that has NEVER encoded -- a receiver of gathered containers, the reference
    # harness's repeat-decode loop -- decodes a foreign directory, then decodes it again under TRC_DIR_READY.  The workspace is
    # poisoned first: nothing an earlier call left behind may make this pass.
    rx = trc.DeviceCoder(codec, n, chunk, "cuda:0")
    rx.work.fill_(0xEE)
    if codec in trc.STATIC:
        rx.set_cdf(cdf, cdfnum)
    for flag in (False, True, True):
        d_out.fill_(0x3C)
        rx.decode(d_out, n, clen=dc.clen, payload=dc.payload, dir_ready=flag)
        torch.cuda.synchronize()
        out = d_out.cpu().numpy()
        assert np.array_equal(out[:n], d) and (out[n:] == 0x3C).all(), "decode-only workspace, dir_ready=%s: mismatch" % flag
    return clen, payload


@pytest.mark.parametrize("codec", trc.AVAILABLE, ids=lambda c: trc.CODEC_NAMES[c])
@pytest.mark.parametrize("kind", ["zipf", "text", "runs", "uniform", "nibble", "binary", "const"])
def test_device_layer_matches_oracle(torch_cuda, codec, kind):
    for n, chunk in [(1, 256), (5, 256), (255, 256), (256, 256), (257, 256), (4096, 4096), (4097, 4096), (70001, 1024),
                     (300007, 4096), (1 << 20, 65536), (999999, 2048)]:
        d = fit(codec, gen(kind, n, 4000 + n))
        _, cdf, cdfnum = T.orc_cdfini(d)
        device_roundtrip(torch_cuda, codec, d, cap(codec, chunk), cdf, cdfnum)


@pytest.mark.parametrize("codec", trc.AVAILABLE, ids=lambda c: trc.CODEC_NAMES[c])
def test_mixed_raw_and_coded_chunks(torch_cuda, codec):
    """incompressible slices inside compressible data: per-chunk raw fallback + raw copy at decode"""
    parts = [gen("zipf", 8192, 1), gen("uniform", 4096, 2), gen("const", 4096, 3), gen("uniform", 8192, 4), gen("text", 5000, 5)]
    d = fit(codec, np.concatenate(parts))
    _, cdf, cdfnum = T.orc_cdfini(d)
    clen, _ = device_roundtrip(torch_cuda, codec, d, 4096, cdf, cdfnum)
    if codec in SMALL_ALPHABET:
        return                                                        # small values always compress: nothing is stored raw
    assert clen[2] == 4096 and clen[4] == 4096 and clen[5] == 4096      # the uniform slices are stored raw
    assert clen[3] < 4096                                             # the constant slice is coded


@pytest.mark.parametrize("codec", trc.AVAILABLE, ids=lambda c: trc.CODEC_NAMES[c])
def test_golden_vectors_single_chunk(torch_cuda, codec):
    """n <= 65536 with chunk >= n: the one payload must equal the reference's whole-buffer output"""
    nib = codec in SMALL_ALPHABET
    vnib = codec in (trc.RCV8, trc.RCVI8)
    z = np.load(os.path.join(GOLD, "vlc_vectors.npz" if codec in trc.VLC_CODECS else "vnib_vectors.npz" if vnib else
                             "nibble_vectors.npz" if nib else "vectors.npz"))
    index = json.loads(bytes(z["index"]).decode())
    name = trc.CODEC_NAMES[codec]
    done = 0
    for ent in index:
        if name not in ent["out"] or ent["n"] > cap(codec, 65536):
            continue
        d = z["in_%d" % ent["case"]]
        n = ent["n"]
        chunk = min(65536, max(256, (n + 63) // 64 * 64))
        cdf = np.zeros(257, dtype=np.uint16)
        if not nib and not vnib:
            cdf[:ent["cdfnum"] + 1] = z["cdf_%d" % ent["case"]]
        dc = trc.DeviceCoder(codec, n, chunk, "cuda:0")
        if codec in trc.STATIC:
            dc.set_cdf(cdf, ent["cdfnum"])
        dc.encode(to_dev(torch_cuda, d), n)
        clen, payload = dc.result(n)
        assert clen.size == 1 and int(clen[0]) == ent["out"][name], (ent["kind"], n)
        exp = d if ent["out"][name] == n else z["out_%d_%s" % (ent["case"], name)]
        assert np.array_equal(payload, exp), (ent["kind"], n)
        done += 1
    assert done > (40 if nib else 100)


@pytest.mark.parametrize("codec", trc.VLC_CODECS, ids=lambda c: trc.CODEC_NAMES[c])
def test_vlc_tiny_payloads(torch_cuda, codec):
    """a few small elements code to 8 bytes (header + one range-coder word, no mantissa bits): the smallest coded chunk"""
    es = trc.VLC_ELEM[codec]
    for nel_last in (1, 2, 5, 6, 9):
        n = 30 * 1984 + nel_last * es
        d = T.int_bytes(n, es, "small", 77 + nel_last)
        clen, _ = device_roundtrip(torch_cuda, codec, d, 1984, None, 0)
        assert clen[-1] <= nel_last * es


def test_bitwise_rans_rejects_multi_block_chunks(torch_cuda):
    dc = trc.DeviceCoder(trc.ANSB, 100000, 16384, "cuda:0")
    with pytest.raises(trc.TrcError):
        dc.encode(to_dev(torch_cuda, gen("zipf", 100000, 1)), 100000)


def test_cdfini_on_device(torch_cuda):
    for kind, n in [("zipf", 1), ("zipf", 1000), ("text", 123457), ("nibble", 4096), ("uniform", 1 << 20), ("const", 777), ("runs", 3000001)]:
        d = gen(kind, n, 77)
        r0, cdf0, cdfnum = T.orc_cdfini(d)
        r1, cdf1, _ = trc.host_cdfini(d, cdfnum)
        assert r0 == r1 and np.array_equal(cdf0, cdf1), (kind, n)
    # a distribution the reference cannot normalise (flat + sparse): reference die()s, we return -1
    d = np.concatenate([np.arange(256, dtype=np.uint8)] * 3 + [np.zeros(5, np.uint8)])
    r0, _, _ = T.orc_cdfini(d, 256)
    r1, _, _ = trc.host_cdfini(d, 256)
    assert r0 == r1


@pytest.mark.parametrize("codec", trc.AVAILABLE, ids=lambda c: trc.CODEC_NAMES[c])
def test_host_pointer_layer(torch_cuda, codec):
    """the reference-named functions: container layout, per-chunk parity, round trip, raw rule"""
    chunk = 1024
    assert trc.lib().trc_set_chunk(chunk) == 0
    try:
        for kind, n in [("zipf", 100000), ("text", 70001), ("runs", 4096)]:
            d = fit(codec, gen(kind, n, 31))
            _, cdf, cdfnum = T.orc_cdfini(d)
            comp = trc.host_encode(codec, d, cdf, cdfnum)
            assert comp.size < n
            hdr, clen, payload = trc.parse_container(comp)
            assert hdr["magic"] == 0x31435254 and hdr["codec"] == codec and hdr["chunk"] == host_chunk(codec, chunk) and hdr["n"] == n
            exp_payload, exp_clen, _ = T.orc_chunked_enc(codec, d, host_chunk(codec, chunk), cdf, cdfnum)
            assert np.array_equal(clen, exp_clen) and np.array_equal(payload, exp_payload)
            assert comp.size == 32 + 4 * clen.size + payload.size
            assert np.array_equal(trc.host_decode(codec, comp, n, cdf, cdfnum), d)
        if codec not in SMALL_ALPHABET:
            d = gen("uniform", 100000, 9)                  # incompressible: returns n, out == in
            _, cdf, cdfnum = T.orc_cdfini(d)
            comp = trc.host_encode(codec, d, cdf, cdfnum)
            assert comp.size == d.size and np.array_equal(comp, d)
        d = fit(codec, gen("zipf", 40, 9))                            # tiny: container overhead >= n -> raw
        _, cdf, cdfnum = T.orc_cdfini(d)
        assert trc.host_encode(codec, d, cdf, cdfnum).size == 40
    finally:
        trc.lib().trc_set_chunk(0)


def test_static_rans_buckets_with_many_symbols(torch_cuda):
    """the static rANS decoder reads ONE table entry per symbol, over buckets of 8 slots; buckets that hold three or more
    symbols -- runs of low-frequency symbols -- take a second route (flag -> 1 KiB LUT row -> symbol table).  Real data
    seldom lands there, so this input is built to: a dominant symbol plus PRESENT rare symbols that sit next to each other
    in the alphabet (frequencies 1-3 of 32768), in every lane of a wave and at every position of a chunk."""
    rng = np.random.default_rng(77)
    for n, chunk, nrare in ((200000, 512, 40), (70001, 1024, 200), (4099, 256, 255)):
        d = np.full(n, 65, dtype=np.uint8)
        rare = np.arange(256 - nrare, 256, dtype=np.uint8) if nrare < 255 else np.array([x for x in range(256) if x != 65], dtype=np.uint8)
        pos = rng.choice(n, size=min(n // 6, 12 * rare.size), replace=False)
        d[pos] = rng.choice(rare, size=pos.size)
        _, cdf, cdfnum = T.orc_cdfini(d)
        f = np.diff(cdf[:cdfnum + 1].astype(np.int64))
        lut = np.searchsorted(cdf[1:cdfnum + 1].astype(np.int64), np.arange(32768), side="right").reshape(4096, 8)
        many = (lut[:, 7] - lut[:, 0]) >= 2
        present = np.zeros(256, bool); present[np.unique(d)] = True
        assert many.any() and present[lut[many].ravel()].any(), "the input must put present symbols into buckets of 3+ symbols"
        assert f.min() >= 1
        device_roundtrip(torch_cuda, trc.ANS4S, d, chunk, cdf, cdfnum)


def test_static_rans_decoder_both_forms(torch_cuda):
    """the static rANS decoder exists in two forms -- one lane per chunk (calls that fill the chip), two lanes per chunk, one per
    rANS state (calls that do not: fewer than 2048 waves) -- chosen by the number of chunks.  Both must decode the same containers:
    each is forced in a process of its own (TRC_ANS_PAIR is read once) on inputs with ragged tails, raw chunks among coded ones
    and every chunk length class (n % 4 = 0..3 decides which state codes the tail)."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys, numpy as np, torch
        sys.path[:0] = [%r, %r]
        import trc, trc_testlib as T
        from golden.make_golden import gen
        for kind, n, chunk in (("text", 300001, 512), ("zipf", 70002, 1024), ("text", 1000003, 4096), ("runs", 5000, 256), ("text", 64 * 512 * 3 + 7, 512)):
            d = gen(kind, n, 5)
            if kind == "text":
                d = d.copy(); d[4096:4096 + 1500] = np.random.default_rng(3).integers(0, 256, 1500, dtype=np.uint8)   # raw chunks
            _, cdf, cdfnum = T.orc_cdfini(d)
            dc = trc.DeviceCoder(trc.ANS4S, n, chunk, "cuda:0"); dc.set_cdf(cdf, cdfnum)
            d_in = torch.from_numpy(np.concatenate([d, np.zeros(512, np.uint8)])).to("cuda:0")
            dc.encode(d_in, n)
            clen, payload = dc.result(n)
            ep, ec, _ = T.orc_chunked_enc(trc.ANS4S, d, chunk, cdf, cdfnum)
            assert np.array_equal(clen, ec) and np.array_equal(payload, ep)
            out = torch.full((n + 512,), 0xA5, dtype=torch.uint8, device="cuda:0")
            dc.decode(out, n); torch.cuda.synchronize()
            o = out.cpu().numpy()
            assert np.array_equal(o[:n], d), (kind, n, chunk)
            assert (o[n:] == 0xA5).all()
        print("ok")
    """) % (os.path.dirname(os.path.abspath(trc.__file__)), os.path.dirname(os.path.abspath(__file__)))
    for form in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, TRC_ANS_PAIR=form))
        assert r.returncode == 0 and "ok" in r.stdout, (form, r.stdout[-2000:] + r.stderr[-3000:])


def test_bitwise_rc_encoder_both_forms(torch_cuda):
    """the bitwise range coder's encoder exists with the deepest tree level in LDS and with it in global memory (a 256-byte row
    per lane in the workspace, filled by the wave itself); the launch picks by wave count.  Each form is forced in a process of
    its own (TRC_RCB_L7G is read once): per-chunk parity with the oracle and round trip, ragged tails and a short last wave
    (dead lanes have rows of their own) included."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys, numpy as np, torch
        sys.path[:0] = [%r, %r]
        import trc, trc_testlib as T
        from golden.make_golden import gen
        for kind, n, chunk in (("text", 300001, 512), ("zipf", 64 * 1024 * 2 + 5, 1024), ("uniform", 40000, 256), ("runs", 70001, 4096), ("text", 700, 256)):
            d = gen(kind, n, 6)
            dc = trc.DeviceCoder(trc.RCB, n, chunk, "cuda:0")
            d_in = torch.from_numpy(np.concatenate([d, np.zeros(512, np.uint8)])).to("cuda:0")
            for rep in range(2):                                   # twice: the second call finds the rows of the first in the workspace
                dc.encode(d_in, n)
                clen, payload = dc.result(n)
                ep, ec, _ = T.orc_chunked_enc(trc.RCB, d, chunk, None, 0)
                assert np.array_equal(clen, ec) and np.array_equal(payload, ep), (kind, n, chunk, rep)
            out = torch.full((n + 512,), 0xA5, dtype=torch.uint8, device="cuda:0")
            dc.decode(out, n); torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy()[:n], d), (kind, n, chunk)
        print("ok")
    """) % (os.path.dirname(os.path.abspath(trc.__file__)), os.path.dirname(os.path.abspath(__file__)))
    for form in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, TRC_RCB_L7G=form))
        assert r.returncode == 0 and "ok" in r.stdout, (form, r.stdout[-2000:] + r.stderr[-3000:])


def test_round4_kernel_forms(torch_cuda):
    """round 4 gave the model-bound coders second forms that the launch code picks between: encoders as a model wave + a coder wave
    (default) or one wave, `anscdf` pass 2 with four lanes per chunk (default) or one, the order-1 model pass as a hi wave + a lo
    wave, decoders as two waves (measured slower, off by default), two histogram kernels.  Every form is forced in a process of its
    own (the switches are read once): per-chunk parity with the oracle and round trip for the coders involved, ragged tails, a
    short last wave, raw chunks."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys, numpy as np, torch
        sys.path[:0] = [%r, %r]
        import trc, trc_testlib as T
        from golden.make_golden import gen
        for codec in (trc.RCA, trc.RCAI, trc.RCB, trc.ANSA, trc.ANSO1):
            for kind, n, chunk in (("text", 300001, 512), ("runs", 64 * 1024 * 2 + 5, 1024), ("uniform", 40000, 256), ("zipf", 270001, 4096), ("text", 701, 256)):
                if codec == trc.ANSO1 and chunk < 1024:
                    continue
                d = gen(kind, n, 8)
                dc = trc.DeviceCoder(codec, n, chunk, "cuda:0")
                d_in = torch.from_numpy(np.concatenate([d, np.zeros(512, np.uint8)])).to("cuda:0")
                dc.encode(d_in, n)
                clen, payload = dc.result(n)
                ep, ec, _ = T.orc_chunked_enc(codec, d, chunk, None, 0)
                assert np.array_equal(clen, ec) and np.array_equal(payload, ep), (trc.CODEC_NAMES[codec], kind, n, chunk)
                out = torch.full((n + 512,), 0xA5, dtype=torch.uint8, device="cuda:0")
                dc.decode(out, n, dir_ready=True); torch.cuda.synchronize()
                o = out.cpu().numpy()
                assert np.array_equal(o[:n], d) and (o[n:] == 0xA5).all(), (trc.CODEC_NAMES[codec], kind, n, chunk)
        d = gen("text", 40000003, 9)                               # (several rounds of the histogram when TRC_HIST_ROUND_VECS=3: 12.6 MB per round)
        dc = trc.DeviceCoder(trc.ANS4S, d.size, 512, "cuda:0")
        d_in = torch.from_numpy(np.concatenate([d, np.zeros(512, np.uint8)])).to("cuda:0")
        hist = torch.zeros(256, dtype=torch.int64, device="cuda:0")
        dc.hist(d_in, d.size, hist); torch.cuda.synchronize()
        assert np.array_equal(hist.cpu().numpy(), np.bincount(d, minlength=256))
        print("ok")
    """) % (os.path.dirname(os.path.abspath(trc.__file__)), os.path.dirname(os.path.abspath(__file__)))
    forms = (dict(TRC_RCA_MC="0", TRC_RCB_MC="0", TRC_ANSA_MC="0", TRC_O1_MC="0", TRC_HIST_FORM="1"),      # everything as in round 3
             dict(TRC_HIST_ROUND_VECS="3"),                                                                  # the histogram's many-round path (real inputs: beyond 4.29 GB)
             dict(TRC_ANSA_CODEQ="0"),                                                                       # planar records, one lane per chunk in pass 2
             dict(TRC_RCA_DMC="1", TRC_ANSA_DMC="1"),                                                        # the two-wave decoders
             dict())                                                                                         # the defaults, same inputs
    for env in forms:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert r.returncode == 0 and "ok" in r.stdout, (env, r.stdout[-2000:] + r.stderr[-3000:])


def test_decoders_take_payloads_at_any_legal_alignment(torch_cuda):
    """round 5: the static decoders fetch their streams in segments aligned to 64 bytes of the PAYLOAD ADDRESS (StreamInT::align_start:
    the ring's stream starts below the first word, the cursor skips the difference).  The device layer only asks for a 2-byte-aligned
    payload pointer: the same container is decoded from payload copies at byte offsets 0 .. 126 of a 256-byte-aligned buffer (every
    residue of the first stream's start modulo 64, the first chunk's `soff < 62` clamp included), raw chunks and a ragged tail among them."""
    torch = torch_cuda
    for codec in (trc.ANS4S, trc.RCS1, trc.RCSM, trc.RCS2):
        n, chunk = 64 * 512 * 3 + 333, 512
        d = gen("text", n, 77)
        d[5 * chunk:6 * chunk] = gen("uniform", chunk, 3)                 # a raw chunk in the first group
        _, cdf, cdfnum = T.orc_cdfini(d)
        dc = trc.DeviceCoder(codec, n, chunk, "cuda:0")
        dc.set_cdf(cdf, cdfnum)
        d_in = to_dev(torch, d)
        dc.encode(d_in, n)
        clen, payload = dc.result(n)
        for shift in (0, 2, 6, 8, 30, 54, 56, 62, 64, 66, 126):
            buf = torch.zeros(payload.size + 1024, dtype=torch.uint8, device="cuda:0")
            buf[shift:shift + payload.size] = torch.from_numpy(payload).to("cuda:0")
            d_out = torch.full((n + 512,), 0xA5, dtype=torch.uint8, device="cuda:0")
            dc.decode(d_out, n, clen=dc.clen, payload=buf[shift:])
            torch.cuda.synchronize()
            out = d_out.cpu().numpy()
            assert np.array_equal(out[:n], d) and (out[n:] == 0xA5).all(), (trc.CODEC_NAMES[codec], shift)


def test_round5_workgroup_shapes(torch_cuda):
    """round 5: the launch code picks between two workgroup shapes by the size of the launch -- small workgroups (rounds 1-4), or
    one large workgroup per CU whose waves keep each other's pace (TrcPace, csrc/trc_dev.h) when the launch is one residency round:
    static rANS / range-coder encoders (1 / 4 vs 12 waves), the two-stream pair encoder (2 vs 12), the four-lanes-per-chunk rANS
    coding passes (4 vs 16), the order-1 decoder (eight lanes per chunk by default; one lane per chunk with 64 / 16 / 8 chunks per wave).  The pace-keeping itself only moves wave priorities.
    Every shape is FORCED in a process of its own (the switches are read once) on inputs of a few groups -- which the automatic
    rule would never give the large shape: ragged tails, a short last workgroup, raw chunks -- per-chunk parity with the oracle
    and round trip."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys, numpy as np, torch
        sys.path[:0] = [%r, %r]
        import trc, trc_testlib as T
        from golden.make_golden import gen
        for codec in (trc.ANS4S, trc.RCS1, trc.RCS2, trc.RCSM, trc.ANSA, trc.ANSB, trc.ANSO1, trc.RCA4, trc.RCAI4, trc.ANSA4):
            for kind, n, chunk in (("text", 300001, 512), ("runs", 64 * 1024 * 13 + 5, 1024), ("uniform", 40000, 256), ("zipf", 270001, 4096), ("text", 701, 256),
                                   ("text", 64 * 256 * 25 + 77, 256)):
                if codec == trc.ANSO1 and chunk < 1024:
                    continue
                d = gen(kind, n, 8)
                if codec in trc.NIBBLE_CODECS:
                    d = d & 15
                _, cdf, cdfnum = T.orc_cdfini(d)
                dc = trc.DeviceCoder(codec, n, chunk, "cuda:0")
                if codec in trc.STATIC:
                    dc.set_cdf(cdf, cdfnum)
                d_in = torch.from_numpy(np.concatenate([d, np.zeros(512, np.uint8)])).to("cuda:0")
                dc.encode(d_in, n)
                clen, payload = dc.result(n)
                ep, ec, _ = T.orc_chunked_enc(codec, d, chunk, cdf, cdfnum)
                assert np.array_equal(clen, ec) and np.array_equal(payload, ep), (trc.CODEC_NAMES[codec], kind, n, chunk)
                out = torch.full((n + 512,), 0xA5, dtype=torch.uint8, device="cuda:0")
                dc.decode(out, n, dir_ready=True); torch.cuda.synchronize()
                o = out.cpu().numpy()
                assert np.array_equal(o[:n], d) and (o[n:] == 0xA5).all(), (trc.CODEC_NAMES[codec], kind, n, chunk)
        print("ok")
    """) % (os.path.dirname(os.path.abspath(trc.__file__)), os.path.dirname(os.path.abspath(__file__)))
    forms = (dict(TRC_ENC_WPB="12", TRC_RCS_ENC_WPB="12", TRC_CODEQ_GPW="4", TRC_O1_ROWS="16", TRC_NIB_BIG="1"),    # the large shapes, on small inputs
             dict(TRC_ENC_WPB="4", TRC_RCS_ENC_WPB="1", TRC_CODEQ_GPW="1", TRC_O1_ROWS="64", TRC_NIB_BIG="0"),      # the small shapes
             dict(TRC_O1_ROWS="8"), dict(TRC_O1_ROWS="4"), dict(TRC_O1_ROWS="2"), dict(TRC_O1_ROWS="1"))     # (order-1 decoder: 4 / 2 / 1 = four / two / eight lanes per chunk)
    for env in forms:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))
        assert r.returncode == 0 and "ok" in r.stdout, (env, r.stdout[-2000:] + r.stderr[-3000:])


def test_order1_model_pass_by_chains_and_by_position(torch_cuda):
    """End of round 5: `anscdf1` has two model passes -- by CHAINS (csrc/trc_ans_o1.hip: the chunk's positions sorted by table, every
    table's chain walked by one lane, the records placed back) for chunks up to 4096 bytes, and the position-order passes of rounds
    1-4 for longer chunks or TRC_O1_CHAINS=0.  Both are forced, each in a process of its own (the switch is read once), on what
    the chain pass has special cases for: incompressible bytes (thousands of one-entry chains: the stream outgrows the placement
    kernel's LDS stage and is gathered from memory), one byte value (two chains as long as the chunk), two alternating values, text,
    runs; odd lengths (the coded dummy), a lone byte, chunk sizes below 4096, several groups of 64 chunks with a short last one,
    inputs of hundreds and thousands of chunks (the walk kernel's workgroups take ceil(chunks / 256) chunks each, 96 at most).
    Per-chunk payloads equal the oracle's; the round trip returns the input."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys, numpy as np, torch
        sys.path[:0] = [%r, %r]
        import trc, trc_testlib as T
        from golden.make_golden import gen
        rng = np.random.default_rng(11)
        def data(kind, n):
            if kind == "random": return rng.integers(0, 256, n, dtype=np.uint8)
            if kind == "const": return np.full(n, 0x41, np.uint8)
            if kind == "alt": return np.where(np.arange(n) & 1, 0x10, 0xfe).astype(np.uint8)
            if kind == "hi16": return (rng.integers(0, 16, n, dtype=np.uint8) << 4).astype(np.uint8)
            return gen(kind, n, 9)
        cases = [("random", 64 * 4096 * 2 + 4097, 4096), ("const", 4096 * 70, 4096), ("const", 4095, 4096), ("alt", 4096 * 3 + 1, 4096),
                 ("hi16", 4096 * 5, 4096), ("text", 64 * 4096 + 1234, 4096), ("runs", 4096 * 130 + 7, 4096), ("zipf", 100001, 2048),
                 ("random", 30001, 1024), ("text", 1, 4096), ("text", 2, 4096), ("random", 3, 1024), ("text", 1 << 20, 65536),
                 ("text", 4096 * 700 + 123, 4096), ("runs", 4096 * 5000 + 1, 4096), ("zipf", 2048 * 9000, 2048)]   # (round 6: the walk kernel takes 3 / 20 / 36 chunks per workgroup here)
        for kind, n, chunk in cases:
            d = data(kind, n)
            dc = trc.DeviceCoder(trc.ANSO1, n, chunk, "cuda:0")
            d_in = torch.from_numpy(np.concatenate([d, np.zeros(512, np.uint8)])).to("cuda:0")
            for rep in range(2):                                  # twice: the second call finds the workspace as the first left it
                dc.encode(d_in, n)
                clen, payload = dc.result(n)
                ep, ec, _ = T.orc_chunked_enc(trc.ANSO1, d, chunk, None, 256)
                assert np.array_equal(clen, ec) and np.array_equal(payload, ep), (kind, n, chunk, rep)
                out = torch.full((n + 512,), 0xA5, dtype=torch.uint8, device="cuda:0")
                dc.decode(out, n, dir_ready=True); torch.cuda.synchronize()
                o = out.cpu().numpy()
                assert np.array_equal(o[:n], d) and (o[n:] == 0xA5).all(), (kind, n, chunk, rep)
        print("ok")
    """) % (os.path.dirname(os.path.abspath(trc.__file__)), os.path.dirname(os.path.abspath(__file__)))
    for env in (dict(TRC_O1_CHAINS="1"), dict(TRC_O1_CHAINS="0")):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))
        assert r.returncode == 0 and "ok" in r.stdout, (env, r.stdout[-2000:] + r.stderr[-3000:])


def test_bounded_host_decoder(torch_cuda):
    """trc_decode_host: the decoder that is told how long its input really is.  A valid container round-trips; a truncated
    buffer, a header that claims more payload than the buffer holds and a directory that does not add up are REJECTED
    before anything is decoded (the reference-named decoders cannot know: their prototypes carry no input length)."""
    import ctypes as C
    L = trc.lib()
    L.trc_decode_host.restype = C.c_size_t
    L.trc_decode_host.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint]
    for codec in (trc.ANS4S, trc.RCS2, trc.RCA, trc.RCB):
        d = gen("text", 300001, 91)
        _, cdf, cdfnum = T.orc_cdfini(d)
        comp = trc.host_encode(codec, d, cdf, cdfnum)
        assert comp.size < d.size
        st = codec in trc.STATIC
        cp = cdf.ctypes.data_as(C.c_void_p) if st else None

        def dec(buf, inlen):
            out = np.full(d.size + 64, 0xA5, dtype=np.uint8)
            src = np.zeros(buf.size + 4096, dtype=np.uint8); src[:buf.size] = buf
            r = L.trc_decode_host(codec, src.ctypes.data_as(C.c_void_p), inlen, out.ctypes.data_as(C.c_void_p), d.size, cp, cdfnum if st else 0)
            return r, out
        r, out = dec(comp, comp.size)
        assert r == d.size and np.array_equal(out[:d.size], d) and (out[d.size:] == 0xA5).all()
        r, _ = dec(comp, comp.size - 1)                                   # truncated
        assert r == 0 and b"container" in L.trc_last_error()
        forged = comp.copy(); forged[24:32] = np.frombuffer(np.uint64(comp.size * 2).tobytes(), dtype=np.uint8)      # header.payload
        r, _ = dec(forged, comp.size)
        assert r == 0
        bad = comp.copy(); bad[32:36] = np.frombuffer(np.uint32(7).tobytes(), dtype=np.uint8)                        # clen[0]: sums no longer match
        r, _ = dec(bad, comp.size)
        assert r == 0
        before = L.trc_last_error()
        r, out = dec(d, d.size)                                           # stored raw: inlen == outlen
        assert r == d.size and np.array_equal(out[:d.size], d)
        assert L.trc_last_error() == before, "a raw copy is a success: it must not report a container error (ADVICE r4)"


def test_host_pointer_layer_is_thread_safe(torch_cuda):
    """host-pointer calls from several threads at once.  Calls on one device serialise on the device
    context's lock and use that context's own copy-thread pools (round 2 had process-wide pools behind per-device locks);
    inputs above 1 MB per slice go through the pools' threaded path.  Eight threads, three coders, sizes straddling the
    pool threshold: every container must equal the one the same call produces alone, and decode back."""
    import concurrent.futures as cf
    chunk = 1024
    assert trc.lib().trc_set_chunk(chunk) == 0
    try:
        jobs = []
        for t in range(8):
            codec = (trc.ANS4S, trc.RCS2, trc.RCA)[t % 3]
            n = (6000001, 900001, 2500003, 300000)[t % 4] + 4096 * t
            d = gen("text" if t % 2 else "zipf", n, 100 + t)
            _, cdf, cdfnum = T.orc_cdfini(d)
            jobs.append((codec, d, cdf, cdfnum))
        alone = [trc.host_encode(c, d, cdf, m) for c, d, cdf, m in jobs]

        def both(j):
            c, d, cdf, m = jobs[j]
            comp = trc.host_encode(c, d, cdf, m)
            back = trc.host_decode(c, comp, d.size, cdf, m)
            return comp, back
        for _ in range(3):
            with cf.ThreadPoolExecutor(8) as ex:
                res = list(ex.map(both, range(len(jobs))))
            for j, (comp, back) in enumerate(res):
                assert np.array_equal(comp, alone[j]), "thread %d: container differs from the single-threaded call" % j
                assert np.array_equal(back, jobs[j][1]), "thread %d: round trip" % j
    finally:
        trc.lib().trc_set_chunk(0)


@pytest.mark.parametrize("codec", trc.AVAILABLE, ids=lambda c: trc.CODEC_NAMES[c])
def test_baseline_size_properties(torch_cuda, codec):
    """100 MB (BASELINE.json configs): round trip on device, directory consistency, sampled chunks == oracle"""
    torch = torch_cuda
    n, chunk = 100 * 1000 * 1000, 4096
    kind = "runs" if codec in (trc.RCA, trc.ANSA) else "text"
    d = (T.nibble_bytes(n, 7, "runs") if codec in trc.NIBBLE_CODECS else
         T.int_bytes(n, trc.VLC_ELEM[codec], "walk", 7) if codec in trc.VLC_CODECS else gen(kind, n, 7))
    _, cdf, cdfnum = T.orc_cdfini(d)
    dc = trc.DeviceCoder(codec, n, chunk, "cuda:0")
    if codec in trc.STATIC:
        dc.set_cdf(cdf, cdfnum)
    d_in = to_dev(torch, d)
    dc.encode(d_in, n)
    d_out = torch.zeros(n + 512, dtype=torch.uint8, device="cuda:0")
    dc.decode(d_out, n)
    torch.cuda.synchronize()
    assert torch.equal(d_out[:n], d_in[:n]), "100 MB round trip failed"
    nch = trc.nchunks(n, chunk)
    clen = dc.clen[:nch].cpu().numpy().view(np.uint32).astype(np.int64)
    total = int(dc.total[0].item())
    assert clen.sum() == total and total < n
    off = np.concatenate([[0], np.cumsum(clen)])
    rng = np.random.default_rng(5)
    payload = dc.payload[:total].cpu().numpy()
    for c in list(rng.integers(0, nch, 48)) + [0, nch - 1]:
        sl = d[c * chunk:(c + 1) * chunk]
        exp = T.orc_enc(codec, sl, cdf, cdfnum)
        assert clen[c] == exp.size and np.array_equal(payload[off[c]:off[c + 1]], exp), "chunk %d" % c


@pytest.mark.parametrize("codec", [trc.ANS4S, trc.RCS1], ids=lambda c: trc.CODEC_NAMES[c])
def test_many_groups_uses_scan_kernel(torch_cuda, codec):
    """> 8192 groups of 64 chunks: the directory goes through the single-workgroup scan kernel instead of
    in-kernel group sums (150 MB at chunk 256 = 9156 groups)"""
    torch = torch_cuda
    n, chunk = 150 * 1000 * 1000, 256
    d = gen("zipf", n, 21)
    _, cdf, cdfnum = T.orc_cdfini(d)
    dc = trc.DeviceCoder(codec, n, chunk, "cuda:0")
    dc.set_cdf(cdf, cdfnum)
    d_in = to_dev(torch, d)
    dc.encode(d_in, n)
    d_out = torch.zeros(n + 512, dtype=torch.uint8, device="cuda:0")
    dc.decode(d_out, n, dir_ready=True)                     # the scanned offsets the encode left in the workspace
    torch.cuda.synchronize()
    assert torch.equal(d_out[:n], d_in[:n])
    d_out.zero_()
    dc.decode(d_out, n)                                     # ... and the ones the decode derives itself
    torch.cuda.synchronize()
    assert torch.equal(d_out[:n], d_in[:n])
    nch = trc.nchunks(n, chunk)
    clen = dc.clen[:nch].cpu().numpy().view(np.uint32).astype(np.int64)
    total = int(dc.total[0].item())
    assert clen.sum() == total
    off = np.concatenate([[0], np.cumsum(clen)])
    payload = dc.payload[:total].cpu().numpy()
    for c in [0, 1, 63, 64, 8192 * 64 - 1, 8192 * 64, nch - 1] + list(np.random.default_rng(1).integers(0, nch, 20)):
        exp = T.orc_enc(codec, d[c * chunk:(c + 1) * chunk], cdf, cdfnum)
        assert clen[c] == exp.size and np.array_equal(payload[off[c]:off[c + 1]], exp), "chunk %d" % c


def test_config5_shard_one_gigabyte(torch_cuda):
    """BASELINE config 5 gives every GPU a 1 GB shard of Zipf(1.1) bytes: static rANS at the bench chunk, 1 953 125 chunks,
    30 518 groups (scan-kernel directory), eight residency rounds.  Round trip on device, directory consistency, sampled
    chunks equal to the oracle.  (The shard is 20 rotated copies of a 50 MB Zipf sample: same statistics, one CDF.)"""
    torch = torch_cuda
    base_n, reps, chunk, codec = 50 * 1000 * 1000, 20, 512, trc.ANS4S
    n = base_n * reps
    base = gen("zipf", base_n, 11)
    _, cdf, cdfnum = T.orc_cdfini(base)
    b = torch.from_numpy(base).to("cuda:0")
    d_in = torch.empty(n + 512, dtype=torch.uint8, device="cuda:0")
    d_in[n:] = 0
    for r in range(reps):
        d_in[r * base_n:(r + 1) * base_n] = torch.roll(b, 4099 * r)
    dc = trc.DeviceCoder(codec, n, chunk, "cuda:0")
    dc.set_cdf(cdf, cdfnum)
    dc.encode(d_in, n)
    d_out = torch.zeros(n + 512, dtype=torch.uint8, device="cuda:0")
    dc.decode(d_out, n)
    torch.cuda.synchronize()
    assert torch.equal(d_out[:n], d_in[:n]), "1 GB round trip failed"
    nch = trc.nchunks(n, chunk)
    clen = dc.clen[:nch].to(torch.int64)
    total = int(dc.total[0].item())
    assert int(clen.sum().item()) == total and total < n
    off = torch.cumsum(clen, 0) - clen
    for c in [0, 1, 63, 64, nch // 2, nch - 65, nch - 1] + list(np.random.default_rng(9).integers(0, nch, 40)):
        c = int(c)
        sl = d_in[c * chunk:(c + 1) * chunk].cpu().numpy()
        exp = T.orc_enc(codec, sl, cdf, cdfnum)
        o, l = int(off[c].item()), int(clen[c].item())
        assert l == exp.size and np.array_equal(dc.payload[o:o + l].cpu().numpy(), exp), "chunk %d" % c


@pytest.mark.parametrize("codec", trc.AVAILABLE, ids=lambda c: trc.CODEC_NAMES[c])
def test_max_rate_bursts(torch_cuda, codec):
    """chunks that stay compressible overall but contain long bursts of the rarest symbols (frequency 1 in the
    static CDF): inside a burst every symbol renormalises, i.e. the coded stream is consumed/produced at the
    maximum rate the ring + look-ahead logic is sized for (32 bytes per 16-symbol period)"""
    rng = np.random.default_rng(7)
    n, chunk = 1 << 20, 4096
    d = np.zeros(n, dtype=np.uint8)
    d[rng.integers(0, n, n // 50)] = 1                         # two common symbols
    rare = np.arange(40, 250, dtype=np.uint8)
    for start in range(1000, n - 2000, 4096):                  # one burst per chunk, different lengths / phases
        ln = int(rng.integers(60, 700))
        d[start:start + ln] = rng.choice(rare, ln)
    d = fit(codec, d)
    _, cdf, cdfnum = T.orc_cdfini(d)
    clen, _ = device_roundtrip(torch_cuda, codec, d, chunk, cdf, cdfnum)
    assert (clen < chunk).mean() > 0.9                         # the chunks are coded, not stored raw
    clen2, _ = device_roundtrip(torch_cuda, codec, d, 512, cdf, cdfnum)


@pytest.mark.parametrize("codec", trc.AVAILABLE, ids=lambda c: trc.CODEC_NAMES[c])
def test_corrupt_input_stays_in_bounds(torch_cuda, codec):
    """The reference decoders are undefined on corrupt input; these must at least stay inside their buffers:
    directory entries above the chunk length read as raw, two-stream headers are clamped to the chunk's payload,
    stream fetches stop at the end of the chunk's payload.  Output is unspecified -- only survival is checked,
    followed by a clean round trip on the same context."""
    torch = torch_cuda
    rng = np.random.default_rng(99 + codec)
    n, chunk = 300007, 1024
    d = fit(codec, gen("zipf", n, 12))
    _, cdf, cdfnum = T.orc_cdfini(d)
    dc = trc.DeviceCoder(codec, n, chunk, "cuda:0")
    if codec in trc.STATIC:
        dc.set_cdf(cdf, cdfnum)
    d_in = to_dev(torch, d)
    dc.encode(d_in, n)
    clen, payload = dc.result(n)
    nch = trc.nchunks(n, chunk)
    d_out = torch.zeros(n + 512, dtype=torch.uint8, device="cuda:0")
    for trial in range(6):
        bad_clen = clen.copy().astype(np.uint32)
        bad_pay = np.concatenate([payload, np.zeros(dc.payload.numel() - payload.size, np.uint8)])
        idx = rng.integers(0, nch, 40)
        kind = trial % 3
        if kind == 0:
            bad_clen[idx] = rng.integers(0, 1 << 32, idx.size, dtype=np.uint64).astype(np.uint32)   # absurd directory entries
        elif kind == 1:
            bad_clen[idx] = rng.integers(0, 12, idx.size).astype(np.uint32)                          # shorter than any header
        else:
            pos = rng.integers(0, payload.size, 3000)
            bad_pay[pos] = rng.integers(0, 256, pos.size).astype(np.uint8)                           # flipped payload bytes
            bad_pay[:64] = 0xFF
        dc.clen[:nch].copy_(torch.from_numpy(bad_clen.view(np.int32)))
        dc.payload.copy_(torch.from_numpy(bad_pay))
        dc.decode(d_out, n)
        torch.cuda.synchronize()
    dc.encode(d_in, n)                                                   # the context still works
    dc.decode(d_out, n)
    torch.cuda.synchronize()
    assert torch.equal(d_out[:n], d_in[:n])


def test_device_layer_rejects_bad_arguments(torch_cuda):
    torch = torch_cuda
    l = trc.lib()
    buf = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda:0")
    p = buf.data_ptr()
    wb = l.trc_work_bytes(trc.RCB, 4096, 4096)
    assert l.trc_encode_dev(trc.RCB, p + 1, 4096, 4096, None, 0, p + 8192, p + 16384, p + 32768, p + 65536, wb, None) == -1     # misaligned input
    assert l.trc_encode_dev(trc.RCB, p, 4096, 100, None, 0, p + 8192, p + 16384, p + 32768, p + 65536, wb, None) == -1        # bad chunk
    assert l.trc_encode_dev(trc.RCB, p, 4096, 4096, None, 0, p + 8192, p + 16384, p + 32768, p + 65536, 16, None) == -3        # workspace too small
    assert l.trc_encode_dev(trc.ANS4S, p, 4096, 4096, None, 0, p + 8192, p + 16384, p + 32768, p + 65536, wb, None) == -4      # static coder without CDF
    assert l.trc_encode_dev(99, p, 4096, 4096, None, 0, p + 8192, p + 16384, p + 32768, p + 65536, wb, None) == -1             # unknown codec
    assert b"codec" in l.trc_last_error()


with open(os.path.join(GOLD, "bench_configs.json")) as _f:
    BENCH_GOLD = {e["name"]: e for e in json.load(_f)}


@pytest.mark.parametrize("cfg", [c for c in T.BENCH_CONFIGS if c["chunk"] is not None], ids=lambda c: c["name"])
def test_bench_config_total_parity(torch_cuda, cfg):
    """TOTAL (not sampled) parity at the exact configurations bench.py reports on: the whole length directory and the
    whole payload of the 100 MB workload equal (a) the committed SHA-256 of the REFERENCE's per-chunk outputs
    (tests/golden/bench_configs.json, generated through oracle/_ref) and (b) the oracle port, byte for byte."""
    import hashlib
    torch = torch_cuda
    g = BENCH_GOLD[cfg["name"]]
    n, chunk, codec = cfg["n"], cfg["chunk"], cfg["codec"]
    d = T.bench_input(cfg["kind"], n, cfg["seed"])
    assert hashlib.sha256(d.tobytes()).hexdigest() == g["in_sha256"], "workload generator drifted"
    d_in = to_dev(torch, d)
    dc = trc.DeviceCoder(codec, n, chunk, "cuda:0")
    cdf, cdfnum = None, 0
    if codec in trc.STATIC:
        dc.cdfini(d_in, n, 256)                                 # as bench.py does: cdfini on device
        torch.cuda.synchronize()
        cdf = np.zeros(257, dtype=np.uint16)
        cdf[:257] = dc.cdf[:257].cpu().numpy().view(np.uint16)
        cdfnum = 256
        assert hashlib.sha256(cdf[:257].tobytes()).hexdigest() == g["cdf_sha256"], "device cdfini differs from the reference's"
    dc.encode(d_in, n)
    clen, payload = dc.result(n)
    assert payload.size == g["payload_bytes"] and clen.size == g["nchunks"]
    assert hashlib.sha256(clen.astype("<u4").tobytes()).hexdigest() == g["clen_sha256"], "length directory differs from the reference"
    assert hashlib.sha256(payload.tobytes()).hexdigest() == g["payload_sha256"], "payload differs from the reference"
    exp_payload, exp_clen = T.orc_chunked_enc_mt(codec, d, chunk, cdf, cdfnum)
    assert np.array_equal(clen, exp_clen) and np.array_equal(payload, exp_payload), "differs from the oracle port"
    d_out = torch.zeros(n + 512, dtype=torch.uint8, device="cuda:0")
    dc.decode(d_out, n)
    torch.cuda.synchronize()
    assert torch.equal(d_out[:n], d_in[:n]), "round trip failed"


@pytest.mark.parametrize("name", ["rcs-text100m-1536", "anscdf-drift100m-1536", "anscdf1-drift100m-4096"])
def test_model_coders_next_to_another_coder_on_a_second_stream(torch_cuda, name):
    """the hand-written carry chains of `rcs` / `anscdf` were only ever exercised with their kernel alone on the
    device (one wave per SIMD).  Here the coder encodes and decodes its 100 MB bench configuration on one stream while a second stream
    keeps the static rANS coder (three waves per SIMD wherever it lands) and the bitwise rANS busy on the same device: every payload
    must hash to the committed SHA-256 of the reference's per-chunk outputs, every decode must return the input.  (The order-1 coder rides
    along since the end of round 5: its chain passes and its eight-lanes-per-chunk decoder keep per-chunk state in LDS and in HBM.)"""
    import hashlib
    torch = torch_cuda
    cfg = [c for c in T.BENCH_CONFIGS if c["name"] == name][0]
    g = BENCH_GOLD[name]
    n, chunk, codec = cfg["n"], cfg["chunk"], cfg["codec"]
    d = T.bench_input(cfg["kind"], n, cfg["seed"])
    d_in = to_dev(torch, d)
    dc = trc.DeviceCoder(codec, n, chunk, "cuda:0")
    d_out = torch.zeros(n + 512, dtype=torch.uint8, device="cuda:0")
    # the neighbours: static rANS on 64 MB of text, bitwise rANS on 32 MB
    nb = 64 * 10**6
    t_in = to_dev(torch, gen("text", nb, 5))
    _, cdf, cdfnum = T.orc_cdfini(t_in[:nb].cpu().numpy())
    n1 = trc.DeviceCoder(trc.ANS4S, nb, 512, "cuda:0"); n1.set_cdf(cdf, cdfnum)
    n2 = trc.DeviceCoder(trc.ANSB, nb // 2, 1536, "cuda:0")
    t_out = torch.zeros(nb + 512, dtype=torch.uint8, device="cuda:0")
    t_out2 = torch.zeros(nb // 2 + 512, dtype=torch.uint8, device="cuda:0")
    side = torch.cuda.Stream(device="cuda:0")
    torch.cuda.synchronize()
    for rep in range(4):
        with torch.cuda.stream(side):
            for _ in range(12):
                n1.encode(t_in, nb); n1.decode(t_out, nb, dir_ready=True)
            n2.encode(t_in, nb // 2); n2.decode(t_out2, nb // 2, dir_ready=True)
        dc.encode(d_in, n)
        d_out.zero_()
        dc.decode(d_out, n, dir_ready=True)
        torch.cuda.synchronize()
        clen, payload = dc.result(n)
        assert hashlib.sha256(clen.astype("<u4").tobytes()).hexdigest() == g["clen_sha256"], "length directory differs from the reference (rep %d)" % rep
        assert hashlib.sha256(payload.tobytes()).hexdigest() == g["payload_sha256"], "payload differs from the reference (rep %d)" % rep
        assert torch.equal(d_out[:n], d_in[:n]), "round trip failed next to a busy second stream (rep %d)" % rep
        assert torch.equal(t_out[:nb], t_in[:nb]) and torch.equal(t_out2[:nb // 2], t_in[:nb // 2]), "the neighbours' round trips failed"


ALIASES = {
    trc.RCS1: [("rccdfsenc", d) for d in ("rccdfsldec", "rccdfsbdec", "rccdfsvldec", "rccdfsvbdec")],
    trc.RCS2: [("rccdfs2enc", "rccdfsl2dec"), ("rccdfs2enc", "rccdfsb2dec")],
    trc.RCSM: [("rccdfsmenc", "rccdfsmldec"), ("rccdfsmenc", "rccdfsmbdec")],
    trc.ANS4S: [("anscdf4senc" + v, "anscdf4sdec" + v) for v in ("", "0", "s", "x")],
    trc.ANSA: [("anscdfenc" + v, "anscdfdec" + v) for v in ("", "0", "s", "x")],
    trc.ANSA4: [("anscdf4enc" + v, "anscdf4dec" + v) for v in ("", "0", "s", "x")],
    trc.ANSO1: [("anscdf1enc" + v, "anscdf1dec" + v) for v in ("", "0", "s", "x")],
    trc.VLAU16: [("anscdfuenc16" + v, "anscdfudec16" + v) for v in ("", "0", "s", "x")],
    trc.VLAUZ16: [("anscdfuzenc16" + v, "anscdfuzdec16" + v) for v in ("", "0", "s", "x")],
    trc.VLAV16: [("anscdfvenc16" + v, "anscdfvdec16" + v) for v in ("", "0", "s", "x")],
    trc.VLAVZ16: [("anscdfvzenc16" + v, "anscdfvzdec16" + v) for v in ("", "0", "s", "x")],
    trc.VLAV32: [("anscdfvenc32" + v, "anscdfvdec32" + v) for v in ("", "0", "s", "x")],
    trc.VLAVZ32: [("anscdfvzenc32" + v, "anscdfvzdec32" + v) for v in ("", "0", "s", "x")],
}


@pytest.mark.parametrize("codec", sorted(ALIASES), ids=lambda c: trc.CODEC_NAMES[c])
def test_every_alias_entry_point_is_called(torch_cuda, codec):
    """the reference exports several names per coder -- the linear/binary/division decoders of the static range coders
    (the harness picks the `l` forms when m<16, turborc.c:495-498), the per-ISA `0/s/x` builds of the rANS coders
    (ids 57/58) -- all of them must RUN here, not just exist: every (encoder, decoder) name pair is called through the
    host-pointer layer, its container compared with the oracle and decoded back."""
    chunk = 1024
    assert trc.lib().trc_set_chunk(chunk) == 0
    try:
        for kind, n in (("zipf", 50001), ("nibble", 20000)):
            d = fit(codec, gen(kind, n, 57))
            _, cdf, cdfnum = T.orc_cdfini(d)
            exp_payload, exp_clen, _ = T.orc_chunked_enc(codec, d, host_chunk(codec, chunk), cdf, cdfnum)
            for en, dn in ALIASES[codec]:
                comp = trc.host_encode(codec, d, cdf, cdfnum, name=en)
                _, clen, payload = trc.parse_container(comp)
                assert np.array_equal(clen, exp_clen) and np.array_equal(payload, exp_payload), en
                assert np.array_equal(trc.host_decode(codec, comp, n, cdf, cdfnum, name=dn), d), dn
    finally:
        trc.lib().trc_set_chunk(0)
