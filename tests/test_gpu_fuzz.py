"""GPU (-m gpu): randomized parity sweep -- random codec / length / chunk size / data kind, device layer,
bit-exact against the oracle per chunk.  Seeds are fixed, so a failure is reproducible."""
import os

import numpy as np
import pytest

import trc
import trc_testlib as T
from golden.make_golden import gen

pytestmark = pytest.mark.gpu
KINDS = ["zipf", "text", "runs", "uniform", "nibble", "binary", "const"]


# TRC_FUZZ_SEEDS=N widens the sweep for a soak run (default 6 seeds x 20 configurations); TRC_FUZZ_CODECS="1,3,12" restricts it to
# those codec ids (soaks of the kernels a round has touched)
_ONLY = [int(x) for x in os.environ.get("TRC_FUZZ_CODECS", "").split(",") if x.strip()]
@pytest.mark.parametrize("seed", range(int(os.environ.get("TRC_FUZZ_SEEDS", "6"))))
def test_random_configurations(seed):
    import torch
    assert torch.cuda.is_available()
    rng = np.random.default_rng(1000 + seed)
    for _it in range(20):
        codec = int(rng.choice(_ONLY or trc.AVAILABLE))
        chunk = int(rng.choice([256, 320, 512, 1024, 1984, 4096, 16384, 65536]))
        if codec == trc.ANSB:
            chunk = min(chunk, 8192)                         # one reference block per chunk
        n = int(rng.choice([int(rng.integers(1, 300)), int(rng.integers(300, 70000)), int(rng.integers(70000, 1500000))]))
        kind = str(rng.choice(KINDS))
        d = gen(kind, n, int(rng.integers(1, 1 << 30)))
        if rng.random() < 0.3 and n > 2000:                  # splice an incompressible / a constant stretch in
            a = int(rng.integers(0, n - 1000)); b = a + int(rng.integers(100, 1000))
            d = d.copy(); d[a:b] = T.uniform_bytes(b - a, seed) if rng.random() < 0.5 else 7
        if codec in trc.NIBBLE_CODECS:
            d = (d & 15).astype(np.uint8)
        if codec in trc.VLC_CODECS and rng.random() < 0.7:    # 16/32-bit series (else: arbitrary bytes read as integers)
            es = trc.VLC_ELEM[codec]
            d = np.concatenate([T.int_bytes(n // es * es, es, str(rng.choice(["small", "walk", "mixed", "wide"])), int(rng.integers(1, 1 << 30))),
                                d[:n % es]]).astype(np.uint8)
        r, cdf, cdfnum = T.orc_cdfini(d)
        if r < 0:                                            # distribution the reference's cdfini cannot normalise
            continue
        dc = trc.DeviceCoder(codec, n, chunk, "cuda:0")
        if codec in trc.STATIC:
            dc.set_cdf(cdf, cdfnum)
        d_in = torch.from_numpy(np.concatenate([d, np.zeros(512, np.uint8)])).to("cuda:0")
        dc.encode(d_in, n)
        clen, payload = dc.result(n)
        exp_payload, exp_clen, _x = T.orc_chunked_enc(codec, d, chunk, cdf, cdfnum)
        tag = (trc.CODEC_NAMES[codec], kind, n, chunk)
        assert np.array_equal(clen, exp_clen), tag
        assert np.array_equal(payload, exp_payload), tag
        d_out = torch.full((n + 512,), 0x5A, dtype=torch.uint8, device="cuda:0")
        dc.decode(d_out, n)
        torch.cuda.synchronize()
        o = d_out.cpu().numpy()
        assert np.array_equal(o[:n], d) and (o[n:] == 0x5A).all(), tag
