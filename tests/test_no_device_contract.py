"""The boundary's failure contract (include/turborc.h:46-59 has no error codes; SURVEY 8b "Errors"): the library has NO CPU
coding path -- the product is the HIP path, and `oracle/` is test infrastructure that the library never links or calls.
What a caller of the reference-named functions gets on a box without a usable GPU is decided here and pinned by this test:
every encoder / decoder returns 0 (no length a valid call can return for n > 0), trc_last_error() says why, nothing is
written to the output, nothing crashes or exits (the reference's die() -> exit(-1), conf.h:369-383, is not reproduced), and
the *_dev entry points return TRC_E_NODEV.  Runs in the build container, which has no GPU; on the GPU box it is skipped."""
import ctypes as C

import numpy as np
import pytest

import trc


def _no_gpu():
    try:
        return trc.lib().trc_device_count() == 0
    except Exception:
        return False


@pytest.mark.skipif(not _no_gpu(), reason="a GPU is visible: the calls would succeed")
def test_reference_named_calls_without_a_device_return_zero_and_say_why():
    lib = trc.lib()
    lib.trc_last_error.restype = C.c_char_p
    n = 20000
    data = np.random.default_rng(5).integers(0, 40, n, dtype=np.uint8)
    cdf = np.arange(0, 257, dtype=np.uint32) * 128
    cdf = cdf.astype(np.uint16); cdf[256] = 32768
    for codec in trc.AVAILABLE:
        for table, what in ((trc._HOST_ENC, "enc"), (trc._HOST_DEC, "dec")):
            out = np.full(2 * n + 4096, 0xA5, dtype=np.uint8)
            f = trc._host_fn(table[codec], codec)
            pin, pout = data.ctypes.data_as(trc._u8p), out.ctypes.data_as(trc._u8p)
            if codec == trc.ANS4S:
                r = f(pin, n, pout, cdf.ctypes.data_as(trc._u16p))
            elif codec in (trc.RCS1, trc.RCS2, trc.RCSM):
                r = f(pin, n, pout, cdf.ctypes.data_as(trc._u16p), 256)
            else:
                r = f(pin, n, pout)
            assert r == 0, (table[codec], r)
            assert b"no HIP device" in lib.trc_last_error(), (table[codec], lib.trc_last_error())
            assert (out == 0xA5).all(), table[codec] + " wrote to its output"


@pytest.mark.skipif(not _no_gpu(), reason="a GPU is visible")
def test_cdfini_without_a_device():
    """cdfini (rccdf.c:50-68) returns the input length on success; without a device it returns 0 and leaves the CDF alone"""
    lib = trc.lib()
    d = np.arange(4096, dtype=np.uint8)
    cdf = np.full(260, 7, dtype=np.uint16)
    f = lib.cdfini
    f.restype = C.c_int; f.argtypes = [trc._u8p, C.c_size_t, trc._u16p, C.c_uint]
    assert f(d.ctypes.data_as(trc._u8p), d.size, cdf.ctypes.data_as(trc._u16p), 256) <= 0
    assert (cdf == 7).all()
