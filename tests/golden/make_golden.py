"""Generate the golden vectors of tests/golden/ from the REFERENCE itself.

Run in the development container only (needs oracle/_ref/libtrc_ref.so, i.e. /root/reference):

    make -C oracle && python tests/golden/make_golden.py

The reference repository ships no test vectors; these files are the pin.  They hold
data only: seeded synthetic inputs, the CDF the reference's cdfini computed, and the bytes the
reference encoders returned (small cases verbatim in vectors.npz, large cases as SHA-256 in
large.json).  Calls follow the reference quirks documented in tests/trc_testlib.py (out above in;
never decode a raw stream).
"""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import trc_testlib as T  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def gen(kind, n, seed):
    if kind == "zipf":
        return T.zipf_bytes(n, 1.1, 256, seed)
    if kind == "text":
        return T.text_bytes(n, seed)
    if kind == "runs":
        return T.runs_bytes(n, seed)
    if kind == "uniform":
        return T.uniform_bytes(n, seed)
    if kind == "nibble":
        return T.zipf_bytes(n, 1.0, 16, seed)
    if kind == "binary":
        return ((T.uniform_bytes(n, seed) & 1) * 7).astype(np.uint8)
    if kind == "const":
        return np.full(n, 65, dtype=np.uint8)
    raise ValueError(kind)


SMALL_SIZES = [1, 2, 3, 4, 5, 7, 8, 9, 10, 11, 12, 15, 16, 17, 31, 32, 33, 63, 64, 70, 71, 255, 256, 257, 1000, 4096, 4097]
MID = [(k, n) for k in ("zipf", "runs") for n in (16384, 65536)] + [("text", 65535), ("uniform", 8192), ("const", 5000)]
LARGE = [("zipf", 10**6, 1), ("text", 10**6, 7), ("runs", 10**6, 3), ("uniform", 10**6, 1),
         ("zipf", (1 << 22) + 1, 5), ("runs", 9 * (1 << 20) + 3, 5)]
CODECS = [T.ANS4S, T.RCS1, T.RCS2, T.RCA, T.ANSA, T.RCB, T.RCAI, T.RCSM, T.ANSO1, T.ANSB]
# `turborc -n` coders: own fixture file so that vectors.npz stays byte-stable
NIB_SIZES = [1, 2, 3, 4, 5, 6, 7, 8, 9, 15, 16, 17, 33, 63, 64, 65, 66, 67, 100, 255, 256, 257, 1000, 1001, 1002, 1003,
             4096, 4097, 4098, 4099, 16384, 65535]
NIB_RCAI4_MIN = 64          # below this the reference's rccdf4ienc returns meaningless lengths / crashes (see oracle)


def main_nibble():
    arrays, index = {}, []
    ci = 0
    for kind in ("geo", "runs", "uniform"):
        for n in NIB_SIZES:
            seed = 3000 + ci
            d = T.nibble_bytes(n, seed, kind)
            arrays["in_%d" % ci] = d
            ent = dict(case=ci, kind=kind, n=n, seed=seed, out={})
            for codec in T.NIBBLE_CODECS:
                if codec == T.RCAI4 and n < NIB_RCAI4_MIN:
                    continue
                o = T.ref_enc(codec, d, variant="s" if codec == T.ANSA4 else "")
                if codec == T.ANSA4:
                    assert np.array_equal(o, T.ref_enc(codec, d, variant="x")), "s/x builds differ"
                if o.size != n and not (codec == T.ANSA4 and n % 4):    # reference anscdf4dec mis-decodes n%4 tails
                    assert np.array_equal(T.ref_dec(codec, o, n, variant="s" if codec == T.ANSA4 else ""), d)
                name = T.CODEC_NAMES[codec]
                ent["out"][name] = int(o.size)
                if o.size != n:
                    arrays["out_%d_%s" % (ci, name)] = o
            index.append(ent)
            ci += 1
    arrays["index"] = np.frombuffer(json.dumps(index).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "nibble_vectors.npz"), **arrays)
    print("wrote", len(index), "nibble cases")


def main_vlc():
    """Turbo-VLC integer coders: 16/32-bit series, sizes that are multiples of the element size"""
    arrays, index = {}, []
    ci = 0
    for kind in ("small", "walk", "mixed", "wide"):
        for ne in (1, 2, 3, 5, 8, 16, 17, 64, 100, 255, 1000, 4096, 16384):
            for es in (2, 4):
                n, seed = ne * es, 5000 + ci
                d = T.int_bytes(n, es, kind, seed)
                arrays["in_%d" % ci] = d
                ent = dict(case=ci, kind=kind, n=n, es=es, seed=seed, out={})
                for codec in T.VLC_CODECS:
                    if T.VLC_ELEM[codec] != es:
                        continue
                    va = "s" if codec in T.VLA_CODECS else ""
                    o = T.ref_enc(codec, d, variant=va)
                    if va:
                        assert np.array_equal(o, T.ref_enc(codec, d, variant="x")), "s/x builds differ"
                    if o.size != n:
                        assert np.array_equal(T.ref_dec(codec, o, n, variant=va), d)
                    name = T.CODEC_NAMES[codec]
                    ent["out"][name] = int(o.size)
                    if o.size != n:
                        arrays["out_%d_%s" % (ci, name)] = o
                index.append(ent)
                ci += 1
    arrays["index"] = np.frombuffer(json.dumps(index).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "vlc_vectors.npz"), **arrays)
    print("wrote", len(index), "vlc cases")


def small_bytes(n, seed, kind):
    """bytes that are mostly small values (what the vnibble coders are for)"""
    u = (T.splitmix64(n, seed) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    if kind == "geo":
        return np.minimum(np.floor(np.log1p(-u) / np.log(0.85)), 255).astype(np.uint8)
    if kind == "mid":
        return (13 + np.minimum(np.floor(np.log1p(-u) / np.log(0.9)), 200)).astype(np.uint8)
    return gen(kind, n, seed)


def main_vnib():
    """"vnibble" coders rccdfenc8 / rccdfienc8 (`turborc -e48/-e49`): own fixture file.  Cases where the reference's
    two-stream output cannot be decoded by the reference itself (stream 0 ran into stream 1, see oracle/trc_oracle.c) are
    recorded as raw (`refbad`), which is what the oracle and the kernels produce."""
    arrays, index = {}, []
    ci = 0
    for kind in ("geo", "mid", "zipf", "text", "uniform"):
        for n in SMALL_SIZES + [16384, 65536]:
            seed = 7000 + ci
            d = small_bytes(n, seed, kind)
            arrays["in_%d" % ci] = d
            ent = dict(case=ci, kind=kind, n=n, seed=seed, out={}, refbad=[])
            for codec in (T.RCV8, T.RCVI8):
                o = T.ref_enc(codec, d)
                name = T.CODEC_NAMES[codec]
                if o.size != n and not np.array_equal(T.ref_dec(codec, o, n), d):
                    assert codec == T.RCVI8 and 4 + int(o[:4].view(np.uint32)[0]) > 4 + n * 37 // 64
                    ent["refbad"].append(name)
                    o = d                                      # raw
                ent["out"][name] = int(o.size)
                if o.size != n:
                    arrays["out_%d_%s" % (ci, name)] = o
            index.append(ent)
            ci += 1
    arrays["index"] = np.frombuffer(json.dumps(index).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "vnib_vectors.npz"), **arrays)
    print("wrote", len(index), "vnibble cases,", sum(len(e["refbad"]) for e in index), "stored raw where the reference output is undecodable")


def main():
    assert T.have_ref(), "reference build missing: make -C oracle"
    arrays, index = {}, []
    cases = [(k, n) for k in ("zipf", "text", "runs", "uniform", "nibble", "binary") for n in SMALL_SIZES] + MID
    for ci, (kind, n) in enumerate(cases):
        seed = 1000 + ci
        d = gen(kind, n, seed)
        r, cdf, cdfnum = T.ref_cdfini(d)
        assert r == n
        arrays["in_%d" % ci] = d
        arrays["cdf_%d" % ci] = cdf[:cdfnum + 1]
        ent = dict(case=ci, kind=kind, n=n, seed=seed, cdfnum=cdfnum, out={})
        for codec in CODECS:
            if codec == T.RCS2 and n < 2:
                continue                                    # reference crashes (wild pointer), see oracle
            o = T.ref_enc(codec, d, cdf, cdfnum)
            for v in ("s", "x") if codec in (T.ANS4S, T.ANSA, T.ANSO1) else ():
                assert np.array_equal(o, T.ref_enc(codec, d, cdf, cdfnum, v)), "s/x builds differ"
            # (ansbc returning exactly n hands back a CODED stream -- its raw test is `>` -- which its caller then treats
            # as raw: the one input class it cannot round-trip; the oracle stores those raw, see oracle/trc_oracle.c)
            rd = None if (codec == T.ANSB and o.size == n) else T.ref_dec(codec, o, n, cdf, cdfnum)
            assert rd is None or np.array_equal(rd, d)
            name = T.CODEC_NAMES[codec]
            ent["out"][name] = int(o.size)
            if o.size != n:                                  # raw outputs are the input itself: no need to store
                arrays["out_%d_%s" % (ci, name)] = o
        index.append(ent)
    arrays["index"] = np.frombuffer(json.dumps(index).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "vectors.npz"), **arrays)

    large = []
    for kind, n, seed in LARGE:
        d = gen(kind, n, seed)
        r, cdf, cdfnum = T.ref_cdfini(d)
        ent = dict(kind=kind, n=n, seed=seed, cdfnum=cdfnum, cdf_sha256=hashlib.sha256(cdf[:cdfnum + 1].tobytes()).hexdigest(),
                   in_sha256=hashlib.sha256(d.tobytes()).hexdigest(), out={})
        for codec in CODECS:
            o = T.ref_enc(codec, d, cdf, cdfnum)
            ent["out"][T.CODEC_NAMES[codec]] = dict(len=int(o.size), sha256=hashlib.sha256(o.tobytes()).hexdigest())
        # chunked totals (what the GPU container must reproduce per chunk) at 4 KiB, from the reference
        if n <= 10**6:
            ch = {}
            for codec in CODECS:
                tot, h = 0, hashlib.sha256()
                for c0 in range(0, n, 4096):
                    o = T.ref_enc(codec, d[c0:c0 + 4096], cdf, cdfnum)
                    tot += o.size; h.update(o.tobytes())
                ch[T.CODEC_NAMES[codec]] = dict(total=tot, sha256=h.hexdigest())
            ent["chunk4096"] = ch
        large.append(ent)
    with open(os.path.join(HERE, "large.json"), "w") as f:
        json.dump(large, f, indent=1)
    print("wrote", len(index), "small cases,", len(large), "large cases")


if __name__ == "__main__":
    if "--vnib-only" in sys.argv:
        main_vnib()
        sys.exit(0)
    if "--nibble-only" not in sys.argv and "--vlc-only" not in sys.argv:
        main()
    if "--vlc-only" not in sys.argv:
        main_nibble()
    main_vlc()
    main_vnib()
