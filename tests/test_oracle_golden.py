"""CPU: the oracle restatement (oracle/trc_oracle.c) against the committed golden vectors, which
were produced by the compiled reference (tests/golden/make_golden.py).  Bit-exact."""
import hashlib
import json
import os

import numpy as np
import pytest

import trc_testlib as T
from golden.make_golden import gen

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CODECS = [T.ANS4S, T.RCS1, T.RCS2, T.RCA, T.ANSA, T.RCB, T.RCAI, T.RCSM, T.ANSO1, T.ANSB]


@pytest.fixture(scope="module")
def vectors():
    z = np.load(os.path.join(GOLD, "vectors.npz"))
    index = json.loads(bytes(z["index"]).decode())
    return z, index


def test_generators_are_stable(vectors):
    """the seeded generators regenerate the committed inputs (so the GPU box can make big inputs)"""
    z, index = vectors
    for ent in index:
        assert np.array_equal(gen(ent["kind"], ent["n"], ent["seed"]), z["in_%d" % ent["case"]]), ent


def test_cdfini_matches_golden(vectors):
    z, index = vectors
    for ent in index:
        r, cdf, cdfnum = T.orc_cdfini(z["in_%d" % ent["case"]])
        assert r == ent["n"] and cdfnum == ent["cdfnum"]
        assert np.array_equal(cdf[:cdfnum + 1], z["cdf_%d" % ent["case"]]), ent


@pytest.mark.parametrize("codec", CODECS, ids=lambda c: T.CODEC_NAMES[c])
def test_encode_matches_golden_and_roundtrips(vectors, codec):
    z, index = vectors
    name = T.CODEC_NAMES[codec]
    seen = 0
    for ent in index:
        if name not in ent["out"]:
            continue
        d = z["in_%d" % ent["case"]]
        cdf = np.zeros(257, dtype=np.uint16); cdf[:ent["cdfnum"] + 1] = z["cdf_%d" % ent["case"]]
        o = T.orc_enc(codec, d, cdf, ent["cdfnum"])
        assert o.size == ent["out"][name], (ent["kind"], ent["n"], name)
        exp = d if o.size == ent["n"] else z["out_%d_%s" % (ent["case"], name)]
        assert np.array_equal(o, exp), (ent["kind"], ent["n"], name)
        assert np.array_equal(T.orc_dec(codec, o, ent["n"], cdf, ent["cdfnum"]), d)
        seen += 1
    assert seen > 100


def test_large_cases_hash():
    with open(os.path.join(GOLD, "large.json")) as f:
        large = json.load(f)
    for ent in large:
        d = gen(ent["kind"], ent["n"], ent["seed"])
        assert hashlib.sha256(d.tobytes()).hexdigest() == ent["in_sha256"]
        r, cdf, cdfnum = T.orc_cdfini(d)
        assert cdfnum == ent["cdfnum"] and hashlib.sha256(cdf[:cdfnum + 1].tobytes()).hexdigest() == ent["cdf_sha256"]
        for codec in CODECS:
            name = T.CODEC_NAMES[codec]
            o = T.orc_enc(codec, d, cdf, cdfnum)
            assert o.size == ent["out"][name]["len"], (ent["kind"], ent["n"], name)
            assert hashlib.sha256(o.tobytes()).hexdigest() == ent["out"][name]["sha256"], (ent["kind"], ent["n"], name)
        if "chunk4096" in ent:
            for codec in CODECS:
                name = T.CODEC_NAMES[codec]
                payload, clen, poff = T.orc_chunked_enc(codec, d, 4096, cdf, cdfnum)
                assert payload.size == ent["chunk4096"][name]["total"]
                assert hashlib.sha256(payload.tobytes()).hexdigest() == ent["chunk4096"][name]["sha256"]
                assert np.array_equal(T.orc_chunked_dec(codec, payload, clen, d.size, 4096, cdf, cdfnum), d)


# ---------------------------------------------------------------- `turborc -n` coders ---
@pytest.fixture(scope="module")
def nibble_vectors():
    z = np.load(os.path.join(GOLD, "nibble_vectors.npz"))
    return z, json.loads(bytes(z["index"]).decode())


@pytest.mark.parametrize("codec", T.NIBBLE_CODECS, ids=lambda c: T.CODEC_NAMES[c])
def test_nibble_encode_matches_golden_and_roundtrips(nibble_vectors, codec):
    z, index = nibble_vectors
    name = T.CODEC_NAMES[codec]
    seen = 0
    for ent in index:
        d = z["in_%d" % ent["case"]]
        assert np.array_equal(T.nibble_bytes(ent["n"], ent["seed"], ent["kind"]), d)
        o = T.orc_enc(codec, d)
        assert np.array_equal(T.orc_dec(codec, o, ent["n"]), d), (ent["kind"], ent["n"], name)
        if name not in ent["out"]:
            continue                                         # reference undefined there (rccdf4ienc, n < 64)
        assert o.size == ent["out"][name], (ent["kind"], ent["n"], name)
        exp = d if o.size == ent["n"] else z["out_%d_%s" % (ent["case"], name)]
        assert np.array_equal(o, exp), (ent["kind"], ent["n"], name)
        seen += 1
    assert seen >= 45


# ---------------------------------------------------------------- Turbo-VLC integer coders ---
@pytest.mark.parametrize("codec", T.VLC_CODECS, ids=lambda c: T.CODEC_NAMES[c])
def test_vlc_encode_matches_golden_and_roundtrips(codec):
    z = np.load(os.path.join(GOLD, "vlc_vectors.npz"))
    index = json.loads(bytes(z["index"]).decode())
    name, seen = T.CODEC_NAMES[codec], 0
    for ent in index:
        if name not in ent["out"]:
            continue
        d = z["in_%d" % ent["case"]]
        assert np.array_equal(T.int_bytes(ent["n"], ent["es"], ent["kind"], ent["seed"]), d)
        o = T.orc_enc(codec, d)
        assert o.size == ent["out"][name], (ent["kind"], ent["n"], name)
        exp = d if o.size == ent["n"] else z["out_%d_%s" % (ent["case"], name)]
        assert np.array_equal(o, exp), (ent["kind"], ent["n"], name)
        assert np.array_equal(T.orc_dec(codec, o, ent["n"]), d)
        seen += 1
    assert seen == 52


def test_headline_bench_config_hash():
    """the oracle port on the WHOLE headline workload (text100m, chunk 512, static rANS) reproduces the committed hash of
    the reference's per-chunk outputs (tests/golden/bench_configs.json, generated through oracle/_ref)"""
    import hashlib
    with open(os.path.join(GOLD, "bench_configs.json")) as f:
        g = {e["name"]: e for e in json.load(f)}["anscdf4s-text100m-512"]
    cfg = [c for c in T.BENCH_CONFIGS if c["name"] == g["name"]][0]
    d = T.bench_input(cfg["kind"], cfg["n"], cfg["seed"])
    assert hashlib.sha256(d.tobytes()).hexdigest() == g["in_sha256"]
    r, cdf, cdfnum = T.orc_cdfini(d, 256)
    assert r == d.size and hashlib.sha256(cdf[:257].tobytes()).hexdigest() == g["cdf_sha256"]
    payload, clen = T.orc_chunked_enc_mt(cfg["codec"], d, cfg["chunk"], cdf, cdfnum)
    assert payload.size == g["payload_bytes"]
    assert hashlib.sha256(clen.astype("<u4").tobytes()).hexdigest() == g["clen_sha256"]
    assert hashlib.sha256(payload.tobytes()).hexdigest() == g["payload_sha256"]


@pytest.mark.parametrize("codec", [T.RCV8, T.RCVI8], ids=lambda c: T.CODEC_NAMES[c])
def test_vnibble_golden_vectors(codec):
    """rccdfenc8 / rccdfienc8 (`turborc -e48/-e49`): the oracle against the committed reference outputs"""
    z = np.load(os.path.join(GOLD, "vnib_vectors.npz"))
    index = json.loads(bytes(z["index"]).decode())
    name = T.CODEC_NAMES[codec]
    for ent in index:
        d, n = z["in_%d" % ent["case"]], ent["n"]
        o = T.orc_enc(codec, d)
        assert o.size == ent["out"][name], (ent["kind"], n)
        assert np.array_equal(o, d if o.size == n else z["out_%d_%s" % (ent["case"], name)]), (ent["kind"], n)
        assert np.array_equal(T.orc_dec(codec, o, n), d)
