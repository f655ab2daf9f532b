#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python - <<'PY' > gpurun_out/r06s.txt 2>&1
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "turbo-range-coder_amd")
import numpy as np, trc, trc_testlib as T
from golden.make_golden import gen
rng = np.random.default_rng(5)
for codec in (trc.RCA, trc.RCAI, trc.ANSA, trc.RCB):
    for it in range(7):
        n = int(rng.integers(1, 6 * 1000 * 1000)) if it else 4096 * 517
        if it == 1: n = 4096 * 64 * 3 + 1
        if it == 2: n = 4095
        d = gen(("text", "zipf", "runs")[it % 3], n, 900 + 31 * it + codec)
        if it == 4:
            u = gen("uniform", n, 77); d[n // 3:n // 3 + 50000] = u[n // 3:n // 3 + 50000]
        comp = trc.host_encode(codec, d)
        if comp.size == n:
            print(codec, it, n, "raw"); continue
        hdr, clen, payload = trc.parse_container(comp)
        chunk = hdr["chunk"]
        exp_payload, exp_clen = T.orc_chunked_enc_mt(codec, d, chunk, None, 0)
        enc_ok = np.array_equal(clen, exp_clen) and np.array_equal(payload, exp_payload)
        back = trc.host_decode(codec, comp, n)
        bad = np.nonzero(back != d)[0]
        lens = np.minimum(chunk, n - np.arange(clen.size) * chunk)
        rawc = np.nonzero(clen == lens)[0]
        print(codec, it, n, "chunk", chunk, "enc_ok", enc_ok, "dec_bad", bad.size, (bad[:4], bad[-2:], (bad[:4] // chunk)) if bad.size else "", "raw chunks", rawc[:5], rawc.size)
PY
cat gpurun_out/r06s.txt
