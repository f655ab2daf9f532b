"""CPU, development container only: fuzz the oracle restatement against the compiled reference
(oracle/_ref/libtrc_ref.so).  Skipped where the reference build is absent (e.g. the GPU box when
the prebuilt library did not travel); the golden vectors cover that case."""
import numpy as np
import pytest

import trc_testlib as T
from golden.make_golden import gen

pytestmark = pytest.mark.skipif(not T.have_ref(), reason="oracle/_ref/libtrc_ref.so not built")
CODECS = [T.ANS4S, T.RCS1, T.RCS2, T.RCA, T.ANSA, T.RCB, T.RCAI, T.RCSM, T.ANSO1, T.ANSB]


@pytest.mark.parametrize("kind", ["zipf", "text", "runs", "uniform", "nibble", "binary"])
def test_fuzz_against_reference(kind):
    rng = np.random.default_rng(12345)
    sizes = [1, 2, 3, 9, 10, 11, 70, 71, 4095, 4096, 4097] + [int(x) for x in rng.integers(12, 70000, 12)]
    for n in sizes:
        d = gen(kind, n, 7000 + n)
        r, cdf, cdfnum = T.orc_cdfini(d)
        r2, cdf2, _ = T.ref_cdfini(d)
        assert r == r2 and np.array_equal(cdf, cdf2)
        for codec in CODECS:
            if codec == T.RCS2 and n < 2:
                continue
            a = T.orc_enc(codec, d, cdf, cdfnum)
            b = T.ref_enc(codec, d, cdf, cdfnum)
            if codec == T.ANSB and b.size == n:      # ansbc: a total of exactly n is a coded stream the reference cannot
                assert a.size == n                   # round-trip (raw test `>`); the oracle stores it raw
                continue
            assert np.array_equal(a, b), (kind, n, T.CODEC_NAMES[codec])
            assert np.array_equal(T.orc_dec(codec, a, n, cdf, cdfnum), d)
            rd = T.ref_dec(codec, a, n, cdf, cdfnum)
            assert rd is None or np.array_equal(rd, d)


def test_all_static_rc_decoders_agree():
    d = gen("zipf", 50000, 3)
    _, cdf, cdfnum = T.orc_cdfini(d)
    a = T.orc_enc(T.RCS1, d, cdf, cdfnum)
    for s in ("l", "b", "vl", "vb"):
        assert np.array_equal(T.ref_dec(T.RCS1, a, d.size, cdf, cdfnum, search=s), d)


def test_lut_division_decoders_agree():
    """rccdfsm{b,l}dec (reciprocal-table division, turborc_.h:172-190) decode the oracle's stream"""
    for kind, n in (("zipf", 50000), ("nibble", 30000)):
        d = gen(kind, n, 3)
        _, cdf, cdfnum = T.orc_cdfini(d)
        a = T.orc_enc(T.RCSM, d, cdf, cdfnum)
        for s in ("b", "l"):
            assert np.array_equal(T.ref_dec(T.RCSM, a, d.size, cdf, cdfnum, search=s), d)


def test_multiblock_adaptive_rans():
    for n in ((1 << 22) - 1, (1 << 22) + 1):
        d = gen("zipf", n, 5)
        a = T.orc_enc(T.ANSA, d)
        assert np.array_equal(a, T.ref_enc(T.ANSA, d))
        assert np.array_equal(T.orc_dec(T.ANSA, a, n), d)


@pytest.mark.parametrize("kind", ["geo", "runs", "uniform"])
def test_nibble_coders_against_reference(kind):
    """`turborc -n` coders.  rccdf4ienc is compared from 64 bytes up (below, the reference returns meaningless
    lengths and crashes: oracle/trc_oracle.c); anscdf4dec of the reference only where n % 4 == 0 (tail defect)."""
    rng = np.random.default_rng(777)
    sizes = list(range(1, 72)) + [255, 256, 257, 4095, 4096, 4097, 4098] + [int(x) for x in rng.integers(72, 90000, 10)]
    for n in sizes:
        d = T.nibble_bytes(n, 9000 + n, kind)
        for codec in T.NIBBLE_CODECS:
            a = T.orc_enc(codec, d)
            assert np.array_equal(T.orc_dec(codec, a, n), d), (kind, n, T.CODEC_NAMES[codec])
            if codec == T.RCAI4 and n < 64:
                continue
            for v in ("s", "x") if codec == T.ANSA4 else ("",):
                assert np.array_equal(a, T.ref_enc(codec, d, variant=v)), (kind, n, T.CODEC_NAMES[codec], v)
                if a.size != n and not (codec == T.ANSA4 and n % 4):
                    assert np.array_equal(T.ref_dec(codec, a, n, variant=v), d)


def test_multiblock_nibble_rans():
    n = (1 << 22) + 4
    d = T.nibble_bytes(n, 11, "geo")
    a = T.orc_enc(T.ANSA4, d)
    assert np.array_equal(a, T.ref_enc(T.ANSA4, d, variant="x"))
    assert np.array_equal(T.orc_dec(T.ANSA4, a, n), d)
    assert np.array_equal(T.ref_dec(T.ANSA4, a, n, variant="s"), d)


@pytest.mark.parametrize("codec", T.VLC_CODECS, ids=lambda c: T.CODEC_NAMES[c])
def test_vlc_integer_coders_against_reference(codec):
    """Turbo-VLC coders on 16/32-bit series (sizes = multiples of the element size, where the reference is defined)"""
    es = T.VLC_ELEM[codec]
    rng = np.random.default_rng(4242)
    for kind in ("small", "walk", "mixed", "wide"):
        for ne in [1, 2, 3, 7, 8, 9, 63, 64, 65, 1023, 1024] + [int(x) for x in rng.integers(1025, 120000, 6)]:
            n = ne * es
            d = T.int_bytes(n, es, kind, 100 + ne)
            a = T.orc_enc(codec, d)
            for v in ("s", "x") if codec in T.VLA_CODECS else ("",):
                assert np.array_equal(a, T.ref_enc(codec, d, variant=v)), (T.CODEC_NAMES[codec], kind, n, v)
                if a.size != n:
                    assert np.array_equal(T.ref_dec(codec, a, n, variant=v), d)
            assert np.array_equal(T.orc_dec(codec, a, n), d)
    for n in (1, 3, 5, 7, 4097):                                 # partial last element: zero-extended, self-consistent
        d = T.int_bytes(n + 8, es, "small", n)[:n]
        assert np.array_equal(T.orc_dec(codec, T.orc_enc(codec, d), n), d)


def test_multiblock_vlc_rans():
    """more than 4 Mi elements: tables restart per block, the bit string and the zigzag predecessor run on"""
    n = ((1 << 22) + 5) * 2
    d = T.int_bytes(n, 2, "walk", 3)
    a = T.orc_enc(T.VLAVZ16, d)
    assert np.array_equal(a, T.ref_enc(T.VLAVZ16, d, variant="x"))
    assert np.array_equal(T.orc_dec(T.VLAVZ16, a, n), d)


def small_bytes(n, seed, kind):
    """what the vnibble coders are for: bytes that are mostly small values"""
    u = (T.splitmix64(n, seed) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    if kind == "geo":
        return np.minimum(np.floor(np.log1p(-u) / np.log(0.85)), 255).astype(np.uint8)
    if kind == "mid":                                            # mostly the two-symbol range 13..44
        return (13 + np.minimum(np.floor(np.log1p(-u) / np.log(0.9)), 200)).astype(np.uint8)
    return gen(kind, n, seed)


@pytest.mark.parametrize("codec", [T.RCV8, T.RCVI8], ids=lambda c: T.CODEC_NAMES[c])
def test_vnibble_coders_against_reference(codec):
    """rccdfenc8 / rccdfienc8 (`turborc -e48/-e49`).  Where the reference lets stream 0's tail run into stream 1 (its own
    decoder then fails: 4 + len0 > 4 + n*37/64) the oracle stores the chunk raw -- the one documented deviation."""
    rng = np.random.default_rng(4848)
    sizes = list(range(1, 80)) + [255, 256, 257, 4095, 4096, 4097] + [int(x) for x in rng.integers(80, 120000, 10)]
    overlaps = 0
    for kind in ("geo", "mid", "zipf", "text", "uniform", "const"):
        for n in sizes:
            d = small_bytes(n, 4800 + n, kind)
            a = T.orc_enc(codec, d)
            b = T.ref_enc(codec, d)
            assert np.array_equal(T.orc_dec(codec, a, n), d), (kind, n)
            if codec == T.RCVI8 and a.size == n and b.size != n:
                len0 = int(b[:4].view(np.uint32)[0])
                assert 4 + len0 > 4 + n * 37 // 64, (kind, n)           # stream 0 overran stream 1's base ...
                assert not np.array_equal(T.ref_dec(codec, b, n), d)    # ... and the reference cannot decode its own output
                overlaps += 1
                continue
            assert np.array_equal(a, b), (kind, n, T.CODEC_NAMES[codec])
            if b.size != n:
                assert np.array_equal(T.ref_dec(codec, a, n), d), (kind, n)
    assert codec == T.RCV8 or overlaps < 40
