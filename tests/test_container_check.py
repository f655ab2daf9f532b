"""CPU: trc_container_check (include/trc_hip.h) -- validation of untrusted TRC1 containers, host-only.

The reference prototypes carry no input length, so a caller handed a file must be able to check that everything a
decoder will read lies inside its buffer (round-1 advisor finding: a crafted/truncated container made the host layer
read past the file buffer)."""
import ctypes as C
import struct

import numpy as np
import pytest

import trc


def make(codec=1, chunk=1024, n=5000, clens=None, payload=None, magic=0x31435254, version=1):
    nch = (n + chunk - 1) // chunk
    clens = list(clens if clens is not None else [100] * nch)
    pay = sum(min(l, min(chunk, n - i * chunk)) for i, l in enumerate(clens)) if payload is None else payload
    hdr = struct.pack("<IBBHIIQQ", magic, codec, version, 256, chunk, len(clens), n, pay)
    return hdr + struct.pack("<%dI" % len(clens), *clens) + bytes(pay)


@pytest.fixture(scope="module")
def check():
    f = trc.lib().trc_container_check
    f.restype = C.c_int
    f.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_size_t]
    return lambda b, blen=None, codec=0, outlen=2**64 - 1: f(b, len(b) if blen is None else blen, codec, outlen)


def test_accepts_well_formed(check):
    b = make()
    assert check(b) == 0 and check(b, codec=1, outlen=5000) == 0
    assert check(b + b"\0" * 100) == 0                       # slack behind the container is fine
    raw_last = make(clens=[100, 100, 100, 100, 904])          # last chunk stored raw (5000 - 4*1024 = 904)
    assert check(raw_last) == 0
    assert check(make(clens=[100, 100, 100, 100, 0xFFFFFFFF])) == 0     # above the chunk length reads as raw: 904 bytes


def test_rejects_malformed(check):
    b = make()
    assert check(b[:31]) != 0                                 # shorter than a header
    assert check(b, blen=40) != 0                             # directory runs past the buffer
    assert check(b, blen=len(b) - 1) != 0                     # payload runs past the buffer (truncated file)
    assert check(make(magic=0x12345678)) != 0
    assert check(make(version=2)) != 0
    assert check(make(codec=99)) != 0
    assert check(b, codec=3) != 0                             # other coder than the caller expects
    assert check(b, outlen=4999) != 0                         # other length than the caller expects
    assert check(make(chunk=1000)) != 0                       # chunk not a multiple of 64
    assert check(make(payload=10**9)) != 0                    # payload above n
    assert check(make(payload=400)) != 0                      # directory does not add up to the payload
    hdr = bytearray(b); hdr[12:16] = struct.pack("<I", 7)      # nchunks inconsistent with n / chunk
    assert check(bytes(hdr)) != 0
    assert b"container" in trc.lib().trc_last_error()
