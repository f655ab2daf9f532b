"""CPU, development container only: fuzz the oracle restatement against the compiled reference
(oracle/_ref/libtrc_ref.so).  Skipped where the reference build is absent (e.g. the GPU box when
the prebuilt library did not travel); the golden vectors cover that case."""
import numpy as np
import pytest

import trc_testlib as T
from golden.make_golden import gen

pytestmark = pytest.mark.skipif(not T.have_ref(), reason="oracle/_ref/libtrc_ref.so not built")
CODECS = [T.ANS4S, T.RCS1, T.RCS2, T.RCA, T.ANSA, T.RCB, T.RCAI]


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


def test_multiblock_adaptive_rans():
    for n in ((1 << 22) - 1, (1 << 22) + 1):
        d = gen("zipf", n, 5)
        a = T.orc_enc(T.ANSA, d)
        assert np.array_equal(a, T.ref_enc(T.ANSA, d))
        assert np.array_equal(T.orc_dec(T.ANSA, a, n), d)
