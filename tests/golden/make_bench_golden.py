"""Generate tests/golden/bench_configs.json THROUGH THE REFERENCE (oracle/_ref/libtrc_ref.so; build container only):

    make -C oracle && python tests/golden/make_bench_golden.py

For every configuration bench.py reports on (trc_testlib.BENCH_CONFIGS: exact workload generator, seed, size, coder
and chunk) the reference function is called on every chunk of the 100 MB input, and the SHA-256 of the concatenated
per-chunk outputs (= the payload area of the TRC1 container) and of the u32 length directory are recorded.  The file
holds data only: hashes, sizes, seeds.  tests/test_gpu_parity.py::test_bench_config_total_parity compares the GPU
path with these hashes (and byte for byte with the oracle port); bench.py prints the same `payload_sha256`.
"""
import concurrent.futures as cf
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import trc_testlib as T  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def ref_chunks(codec, d, chunk, cdf, cdfnum, lo, hi):
    """reference outputs of chunks [lo, hi) -> (list of arrays, lengths)"""
    outs, lens = [], []
    for c in range(lo, hi):
        o = T.ref_enc(codec, d[c * chunk:(c + 1) * chunk], cdf, cdfnum, "s" if codec == T.ANSA else "")
        outs.append(o); lens.append(o.size)
    return np.concatenate(outs), np.asarray(lens, dtype=np.uint32)


def main():
    assert T.have_ref(), "reference build missing: make -C oracle"
    res, cache = [], {}
    path = os.path.join(HERE, "bench_configs.json")
    have = {e["name"]: e for e in json.load(open(path))} if os.path.exists(path) and "--all" not in sys.argv else {}
    whole = {}                                       # (codec, kind, n, seed) -> bytes the reference function returns for the WHOLE input
    def load(key):
        if key not in cache:
            d = T.bench_input(*key)
            r, cdf, cdfnum = T.ref_cdfini(d, 256)
            assert r == d.size
            cache[key] = (d, cdf, cdfnum)
        return cache[key]
    def whole_buffer(cfg):
        """round 4: what the chunking costs.  The reference semantics are ONE call over the whole input (rccdf.c:201-211,
        anscdf.c:567-586 in 4 MiB blocks, anscdf.c:57-73, rc_.c:47-58); its size goes beside the per-chunk payload size."""
        wk = (cfg["codec"], cfg["kind"], cfg["n"], cfg["seed"])
        if wk not in whole:
            d, cdf, cdfnum = load((cfg["kind"], cfg["n"], cfg["seed"]))
            whole[wk] = int(T.ref_enc(cfg["codec"], d, cdf, cdfnum, "s" if cfg["codec"] == T.ANSA else "").size)
        return whole[wk]
    for cfg in T.BENCH_CONFIGS:
        if cfg["name"] in have:                      # entries already committed are kept as they are (--all regenerates everything)
            e = have[cfg["name"]]
            if "whole_buffer_bytes" not in e:
                e["whole_buffer_bytes"] = whole_buffer(cfg)
                print(e["name"], "whole buffer", e["whole_buffer_bytes"])
            res.append(e); continue
        if cfg["chunk"] is None:                     # whole-buffer size only (1 GB inputs)
            ent = dict(name=cfg["name"], codec=T.CODEC_NAMES[cfg["codec"]], kind=cfg["kind"], seed=cfg["seed"], n=cfg["n"], chunk=None,
                       whole_buffer_bytes=whole_buffer(cfg))
            print(ent["name"], "whole buffer", ent["whole_buffer_bytes"])
            res.append(ent); continue
        key = (cfg["kind"], cfg["n"], cfg["seed"])
        d, cdf, cdfnum = load(key)
        n, chunk, codec = cfg["n"], cfg["chunk"], cfg["codec"]
        nch = (n + chunk - 1) // chunk
        nthr = min(64, os.cpu_count() or 1)
        per = (nch + nthr - 1) // nthr
        with cf.ThreadPoolExecutor(nthr) as ex:
            parts = list(ex.map(lambda lo: ref_chunks(codec, d, chunk, cdf, cdfnum, lo, min(nch, lo + per)), range(0, nch, per)))
        payload = np.concatenate([p[0] for p in parts]); clen = np.concatenate([p[1] for p in parts])
        # the oracle port must agree with the reference on the whole workload before the hash is trusted as a pin
        op, oc = T.orc_chunked_enc_mt(codec, d, chunk, cdf, cdfnum)
        assert np.array_equal(oc, clen) and np.array_equal(op, payload), cfg["name"]
        ent = dict(name=cfg["name"], codec=T.CODEC_NAMES[codec], kind=cfg["kind"], seed=cfg["seed"], n=n, chunk=chunk, nchunks=nch,
                   in_sha256=hashlib.sha256(d.tobytes()).hexdigest(), cdfnum=cdfnum,
                   cdf_sha256=hashlib.sha256(cdf[:cdfnum + 1].tobytes()).hexdigest(),
                   payload_bytes=int(payload.size), payload_sha256=hashlib.sha256(payload.tobytes()).hexdigest(),
                   clen_sha256=hashlib.sha256(clen.astype("<u4").tobytes()).hexdigest(), raw_chunks=int((clen == np.minimum(chunk, n - np.arange(nch) * chunk)).sum()),
                   whole_buffer_bytes=whole_buffer(cfg))
        print(ent["name"], ent["payload_bytes"], ent["payload_sha256"][:16])
        res.append(ent)
    with open(path, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
