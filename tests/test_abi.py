"""CPU: the C-ABI library loads and exports every symbol include/*.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "turbo-range-coder_amd", "libturborc_hip.so")
PROTO = re.compile(r"^\s*(?:LIBAPI\s+)?(?:const\s+)?(?:size_t|int|void|uint32_t|char\s*\*|const char\s*\*)\s+\*?(\w+)\s*\(", re.M)


def declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(PROTO.findall(txt)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        import __graft_entry__ as g
        g.build()
    return ctypes.CDLL(LIB)


@pytest.mark.parametrize("header", ["trc_hip.h", "turborc.h", "anscdf.h"])
def test_every_declared_symbol_is_exported(lib, header):
    names = declared(header)
    assert names, header
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, "declared in include/%s but not exported: %s" % (header, missing)


def test_expected_reference_names_present():
    names = set(declared("turborc.h")) | set(declared("anscdf.h"))
    for n in ("cdfini", "anscdf4senc", "anscdf4sdec", "anscdfini", "anscdfenc", "anscdfdec", "anscdfencs", "anscdfdecx",
              "rccdfsenc", "rccdfsbdec", "rccdfsldec", "rccdfsvbdec", "rccdfsvldec", "rccdfs2enc", "rccdfsb2dec", "rccdfsl2dec",
              "rccdfenc", "rccdfdec", "rcsenc", "rcsdec",
              # SURVEY 8f ranks 1-2
              "rccdfienc", "rccdfidec", "rccdf4enc", "rccdf4dec", "rccdf4ienc", "rccdf4idec", "rccdfsmenc", "rccdfsmbdec", "rccdfsmldec",
              "anscdf4enc", "anscdf4dec", "anscdf4encs", "anscdf4decx", "anscdf1enc", "anscdf1dec", "anscdf1encs", "anscdf1decx"):
        assert n in names


def test_reference_dispatch_globals_exported(lib):
    """include/anscdf.h declares the reference's dispatch globals (reference include/anscdf.h:32-35)"""
    for g in ("_anscdfenc", "_anscdfdec", "_anscdf4enc", "_anscdf4dec"):
        assert ctypes.c_void_p.in_dll(lib, g).value, g


def test_config_calls_work_without_gpu(lib):
    lib.trc_get_chunk.restype = ctypes.c_uint32
    assert lib.trc_get_chunk() % 64 == 0
    lib.trc_work_bytes.restype = ctypes.c_size_t
    lib.trc_work_bytes.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_uint32]
    assert lib.trc_work_bytes(1, 100 * 1000 * 1000, 4096) > 100 * 1000 * 1000
    assert lib.trc_set_chunk(100) != 0          # rejected: not a multiple of 64 in range
    lib.trc_last_error.restype = ctypes.c_char_p
    assert b"chunk" in lib.trc_last_error()
    # the chunk size of a host-pointer call (round 6): the largest chunk whose one-wave time hides behind the call's PCIe time
    # (budget max(1.7 ms, 0.35 n / 50 GB/s)) -- the caller sees the ratio, and the link, not the kernels, sets the pace
    lib.trc_auto_chunk.restype = ctypes.c_uint32
    lib.trc_auto_chunk.argtypes = [ctypes.c_size_t]
    assert [lib.trc_auto_chunk(n) for n in (1, 100 * 10**6, 8 * 10**9)] == [4096, 4096, 4096]          # static coders: 4096 (nothing to gain above)
    lib.trc_auto_chunk_codec.restype = ctypes.c_uint32
    lib.trc_auto_chunk_codec.argtypes = [ctypes.c_int, ctypes.c_size_t]
    GB = 10**9
    for codec in (1, 2, 3, 11):
        assert [lib.trc_auto_chunk_codec(codec, n) for n in (1, 100 * 10**6, 8 * GB)] == [4096, 4096, 4096]
    assert [lib.trc_auto_chunk_codec(4, n) for n in (1, 100 * 10**6, 200 * 10**6, 400 * 10**6, GB)] == [4096, 4096, 4096, 6144, 16384]    # rccdf
    assert [lib.trc_auto_chunk_codec(5, n) for n in (1, 100 * 10**6, GB)] == [4096, 4096, 16384]                                         # anscdf
    assert [lib.trc_auto_chunk_codec(6, n) for n in (1, 100 * 10**6, GB)] == [2048, 2048, 8192]                                         # rcs: 592 ns per chunk byte
    assert [lib.trc_auto_chunk_codec(12, n) for n in (1, 100 * 10**6, GB, 8 * GB)] == [4096, 4096, 6144, 16384]                                    # order-1 rANS: never below 4096
    assert [lib.trc_auto_chunk_codec(13, n) for n in (1, 100 * 10**6, 8 * GB)] == [2048, 2048, 8192]                                     # bitwise rANS: within one reference block
    for codec in range(1, 28):
        for n in (1, 10**6, 100 * 10**6, GB, 8 * GB):
            c = lib.trc_auto_chunk_codec(codec, n)
            assert c % 64 == 0 and 512 <= c <= 16384 and c >= lib.trc_auto_chunk_codec(codec, max(1, n // 2)), (codec, n, c)             # grows with the input
    # the chunk of a device-resident call: the input becomes a whole number of residency rounds of the coder's lanes, barely
    lib.trc_round_chunk.restype = ctypes.c_uint32
    lib.trc_round_chunk.argtypes = [ctypes.c_int, ctypes.c_size_t]
    MB = 10**6
    assert [lib.trc_round_chunk(1, n) for n in (1, 70 * MB, 100 * MB, 120 * MB, 333 * MB, 10**9)] == [512, 512, 512, 640, 1728, 5120]       # static rANS: 196 608 lanes
    assert [lib.trc_round_chunk(3, n) for n in (100 * MB,)] == [1024]                                                                       # -e45: two lanes per chunk
    assert [lib.trc_round_chunk(4, n) for n in (1, 70 * MB, 100 * MB, 120 * MB, 150 * MB, 333 * MB, 10**9)] == [512, 1088, 1536, 1856, 2304, 5120, 15296]   # 65 536 lanes (round 5: one round up to 16 KiB, not four of 3840)
    assert [lib.trc_round_chunk(13, n) for n in (100 * MB, 10**9)] == [1536, 7680]                                                           # bitwise rANS: within one reference block
    for codec in (1, 4, 5, 6, 7):
        for n in (70 * MB, 100 * MB, 120 * MB, 150 * MB, 333 * MB, 10**9, 8 * 10**9):
            c = lib.trc_round_chunk(codec, n)
            rc = {1: 196608, 4: 65536, 5: 65536, 6: 65536, 7: 65536}[codec]
            rounds = -(-(-(-n // c)) // rc)
            assert c % 64 == 0 and 512 <= c <= 16384
            assert c == 512 or -(-n // c) > 0.93 * rounds * rc, (codec, n, c)      # the last round is (nearly) full: 64-byte steps of the chunk
            assert c == 512 or -(-n // (c - 64)) > rounds * rc                    # ... and no smaller chunk is: c is the largest that fits
            assert rounds == 1 or -(-n // rc) > 16384                             # more than one round only when one round would need a chunk above the cap
    if "TRC_CHUNK" not in os.environ:
        assert lib.trc_get_chunk() == 0                                   # automatic by default
    assert lib.trc_set_chunk(2048) == 0 and lib.trc_get_chunk() == 2048
    assert lib.trc_set_chunk(0) == 0 and lib.trc_get_chunk() == 0         # back to automatic


def test_no_cpu_coding_path(lib):
    """without a HIP device every coder entry point fails loudly (message + return 0) instead of coding on the CPU"""
    import numpy as np
    lib.trc_device_count.restype = ctypes.c_int
    if lib.trc_device_count() > 0:
        pytest.skip("a GPU is present")
    lib.trc_last_error.restype = ctypes.c_char_p
    d = np.arange(4096, dtype=np.uint8)
    out = np.zeros(8192, dtype=np.uint8)
    for name in ("rcsenc", "rccdfenc", "anscdfenc", "rccdf4enc", "anscdf1enc"):
        f = getattr(lib, name)
        f.restype = ctypes.c_size_t
        f.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        assert f(d.ctypes.data, d.size, out.ctypes.data) == 0, name
        assert b"no HIP device" in lib.trc_last_error(), name
        assert not out.any(), name                           # nothing was written


def test_headers_compile_as_plain_c(tmp_path):
    """include/turborc.h + include/anscdf.h are plain C a TurboRC-style caller can include: prototypes, cdf_t, the
    dispatch typedefs of the reference (include/anscdf.h:27-30) and its globals (:32-35)"""
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('''
#include "turborc.h"
#include "anscdf.h"
#include "trc_hip.h"
fanscdfenc e = anscdfenc; fanscdfdec d = anscdfdec;
fanscdf4senc se = anscdf4senc; fanscdf4sdec sd = anscdf4sdec;
size_t f(unsigned char *a, size_t n, unsigned char *b, cdf_t *c) { return _anscdfenc(a, n, b) + se(a, n, b, c) + rccdfs2enc(a, n, b, c, 256); }
''')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)])
    subprocess.check_call(["g++", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), "-x", "c++", str(src)])


def test_host_call_plans(lib):
    """the slices (launches) and parts of a host-pointer call, for every coder / size / chunk / buffer kind (trc_host_plan needs no device):
    slices tile the chunk range and start on group boundaries (the decoders find a slice's payload through per-group sums: a slice that
    started inside a group decoded from the wrong offset in the first version of round 6's plan); striped / streamed calls only for the
    coders that support them, from page-locked input (encode), with parts that are multiples of 128 bytes of at least 1 KB, at most
    eight of them (pageable output: four), and launches of at most one residency round"""
    import numpy as np
    lib.trc_host_plan.restype = ctypes.c_int
    lib.trc_host_plan.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t), ctypes.c_int, ctypes.POINTER(ctypes.c_uint32)]
    lib.trc_auto_chunk_codec.restype = ctypes.c_uint32
    lib.trc_auto_chunk_codec.argtypes = [ctypes.c_int, ctypes.c_size_t]
    first = (ctypes.c_size_t * 4096)()
    part = ctypes.c_uint32(0)
    gated_codecs = {4, 5, 6, 7}                                  # rccdf, anscdf, rcs, rccdfi
    rounds = {1: 196608, 2: 196608, 3: 98304, 11: 196608, 4: 65536, 5: 65536, 6: 65536, 7: 65536, 13: 65536}
    seen_gated = 0
    for codec in (1, 2, 3, 4, 5, 6, 7, 11, 12, 13, 26):
        for n in (1, 4095, 4096, 70001, 1 << 20, 5 * 10**6 + 3, 100 * 10**6, 333 * 10**6 + 77, 10**9, 3 * 10**9 + 5):
            for chunk in (0, 512, 1024, 2048, 4096, 4160, 16384, 65536):
                for decode in (0, 1):
                    for pinned in (0, 1):
                        ch = chunk or lib.trc_auto_chunk_codec(codec, n)
                        if codec == 13 and ch > 8192:
                            continue
                        nsl = lib.trc_host_plan(codec, n, chunk, decode, pinned, first, 4096, ctypes.byref(part))
                        assert 1 <= nsl <= 2100, (codec, n, chunk, decode, pinned, nsl)
                        f = np.array(first[:nsl + 1], dtype=np.int64)
                        nch = -(-n // ch)
                        assert f[0] == 0 and f[-1] == nch and np.all(np.diff(f) > 0), (codec, n, chunk, f[:5])
                        assert np.all(f[:-1] % 64 == 0), (codec, n, chunk, decode, pinned, f[:8])
                        if part.value:
                            seen_gated += 1
                            assert codec in gated_codecs and ch % 128 == 0 and ch >= 2048 and n >= ch
                            assert decode or pinned, "striped input needs page-locked memory"
                            assert part.value % 128 == 0 and part.value >= 1024 and -(-ch // part.value) <= (8 if (pinned or not decode) else 4)
                            assert np.all(np.diff(f) <= rounds[codec]), "a gated launch is at most one residency round"
                            if decode:
                                assert n <= (512 << 20 if pinned else 256 << 20)
                        else:
                            assert not (codec in gated_codecs and ch % 128 == 0 and ch >= 2048 and n >= ch and pinned and (not decode or n <= (512 << 20)) and "TRC_HOST_NO_STRIPE" not in os.environ and "TRC_HOST_NO_GATE" not in os.environ and nsl <= 16), (codec, n, chunk, decode)
    assert seen_gated > 100
