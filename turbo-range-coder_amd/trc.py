"""ctypes binding of libturborc_hip.so for tests / bench.py (plumbing only).

The product is the C-ABI shared library (include/*.h).  This module only
  * builds/loads it (in-tree, so the GPU box sees the .so that was actually used),
  * wraps the device-resident entry points around torch tensors (torch = device memory + streams),
  * wraps the reference-signature host-pointer calls around numpy arrays.
There is no CPU fallback: loading fails loudly if the library is missing.
"""
import ctypes as C
import os
import subprocess

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
LIB = os.environ.get("TRC_LIB") or os.path.join(PKG, "libturborc_hip.so")   # TRC_LIB: A/B builds for ablations

ANS4S, RCS1, RCS2, RCA, ANSA, RCB, RCAI, RCA4, RCAI4, ANSA4, RCSM, ANSO1, ANSB = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13
VLCU16, VLCU32, VLCV16, VLCV32, VLCVZ16, VLCVZ32 = 14, 15, 16, 17, 18, 19   # Turbo-VLC integer coders
VLAU16, VLAUZ16, VLAV16, VLAVZ16, VLAV32, VLAVZ32 = 20, 21, 22, 23, 24, 25   # ... over the CDF rANS
RCV8, RCVI8 = 26, 27                                                          # "vnibble" coders (turborc -e48 / -e49)
CODEC_NAMES = {ANS4S: "anscdf4s", RCS1: "rccdfs", RCS2: "rccdfs2", RCA: "rccdf", ANSA: "anscdf", RCB: "rcs", RCAI: "rccdfi",
               RCA4: "rccdf4", RCAI4: "rccdf4i", ANSA4: "anscdf4", RCSM: "rccdfsm", ANSO1: "anscdf1", ANSB: "ansb",
               VLCU16: "rccdfu16", VLCU32: "rccdfu32", VLCV16: "rccdfv16", VLCV32: "rccdfv32", VLCVZ16: "rccdfvz16", VLCVZ32: "rccdfvz32",
               VLAU16: "anscdfu16", VLAUZ16: "anscdfuz16", VLAV16: "anscdfv16", VLAVZ16: "anscdfvz16", VLAV32: "anscdfv32", VLAVZ32: "anscdfvz32",
               RCV8: "rccdf8", RCVI8: "rccdfi8"}
VLC_CODECS = (VLCU16, VLCU32, VLCV16, VLCV32, VLCVZ16, VLCVZ32, VLAU16, VLAUZ16, VLAV16, VLAVZ16, VLAV32, VLAVZ32)
VLC_ELEM = {VLCU16: 2, VLCU32: 4, VLCV16: 2, VLCV32: 4, VLCVZ16: 2, VLCVZ32: 4,
            VLAU16: 2, VLAUZ16: 2, VLAV16: 2, VLAVZ16: 2, VLAV32: 4, VLAVZ32: 4}
NIBBLE_CODECS = (RCA4, RCAI4, ANSA4)                           # `turborc -n` coders: input values 0..15
STATIC = (ANS4S, RCS1, RCS2, RCSM)
TABLES_READY = 0x100                                          # include/trc_hip.h
DIR_READY = 0x200
AVAILABLE = (ANS4S, RCS1, RCS2, RCB, RCA, ANSA, RCAI, RCA4, RCAI4, ANSA4, RCSM, ANSO1, ANSB) + VLC_CODECS + (RCV8, RCVI8)          # codecs with HIP kernels behind them (grows per round)
PAD = 256
HDR = 32

_u8p = C.POINTER(C.c_uint8)
_u16p = C.POINTER(C.c_uint16)
_vp = C.c_void_p
_sz = C.c_size_t


def build(force=False):
    """Compile every HIP translation unit for gfx950 into turbo-range-coder_amd/libturborc_hip.so."""
    if force:
        subprocess.check_call(["make", "-s", "-C", PKG, "clean"])
    subprocess.check_call(["make", "-s", "-j8", "-C", PKG])
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise RuntimeError("libturborc_hip.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                               "there is no CPU fallback")
        # The process must end up with ONE HIP runtime.  PyTorch's wheel brings its own libamdhip64; if this library is
        # loaded first it pulls in /opt/rocm's copy, torch then loads its own, and device pointers of one runtime reach
        # launches of the other ("no ROCm-capable device is detected" on the first call).  With torch imported first the
        # library's dependency resolves to the runtime that is already there.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        l = C.CDLL(LIB)
        l.trc_last_error.restype = C.c_char_p
        l.trc_device_count.restype = C.c_int
        l.trc_set_chunk.restype = C.c_int; l.trc_set_chunk.argtypes = [C.c_uint32]
        l.trc_get_chunk.restype = C.c_uint32
        l.trc_round_chunk.restype = C.c_uint32; l.trc_round_chunk.argtypes = [C.c_int, _sz]
        l.trc_auto_chunk_codec.restype = C.c_uint32; l.trc_auto_chunk_codec.argtypes = [C.c_int, _sz]
        l.trc_work_bytes.restype = _sz; l.trc_work_bytes.argtypes = [C.c_int, _sz, C.c_uint32]
        l.trc_cdfini_dev.restype = C.c_int
        l.trc_cdfini_dev.argtypes = [_vp, _sz, _vp, C.c_uint, _vp, _vp, _vp]
        l.trc_hist_dev.restype = C.c_int; l.trc_hist_dev.argtypes = [_vp, _sz, _vp, _vp]
        l.trc_cdf_from_hist_dev.restype = C.c_int; l.trc_cdf_from_hist_dev.argtypes = [_vp, _sz, _vp, C.c_uint, _vp, _vp]
        l.trc_encode_dev.restype = C.c_int
        l.trc_encode_dev.argtypes = [C.c_int, _vp, _sz, C.c_uint32, _vp, C.c_uint, _vp, _vp, _vp, _vp, _sz, _vp]
        l.trc_tables_dev.restype = C.c_int
        l.trc_tables_dev.argtypes = [_vp, C.c_uint, _vp, _sz, _vp]
        l.trc_decode_dev.restype = C.c_int
        l.trc_decode_dev.argtypes = [C.c_int, _vp, _vp, _sz, C.c_uint32, _vp, C.c_uint, _vp, _vp, _sz, _vp]
        l.trc_timing_enable.restype = C.c_int; l.trc_timing_enable.argtypes = [C.c_int]
        l.trc_timing_pause.restype = C.c_int; l.trc_timing_pause.argtypes = [C.c_int]
        l.trc_timing_read.restype = C.c_int
        l.trc_timing_read.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]
        l.trc_kernel_name.restype = C.c_char_p; l.trc_kernel_name.argtypes = [C.c_int, C.c_int]
        l.trc_set_devices.restype = C.c_int; l.trc_set_devices.argtypes = [C.POINTER(C.c_int), C.c_int]
        l.trc_get_devices.restype = C.c_int; l.trc_get_devices.argtypes = [C.POINTER(C.c_int), C.c_int]
        _lib = l
    return _lib


class TrcError(RuntimeError):
    pass


def _chk(rc):
    if rc != 0:
        raise TrcError("libturborc_hip rc=%d: %s" % (rc, lib().trc_last_error().decode()))


def timing_enable(on=True):
    _chk(lib().trc_timing_enable(1 if on else 0))


def timing_pause(paused=True):
    """suspend / resume the event pairs without resetting what has been collected"""
    _chk(lib().trc_timing_pause(1 if paused else 0))


def timing_read(decode):
    """-> (total_ms, launches) of the coder kernels since timing_enable(); decode = 2: the encode path's scan + gather kernels"""
    ms, cnt = C.c_double(0), C.c_int(0)
    _chk(lib().trc_timing_read(2 if decode == 2 else 1 if decode else 0, C.byref(ms), C.byref(cnt)))
    return ms.value, cnt.value


def nchunks(n, chunk):
    return (n + chunk - 1) // chunk


# ---------------------------------------------------------------- device-resident layer (torch) ---
class DeviceCoder:
    """Pre-allocated HBM buffers for repeated encode/decode of up to `n` bytes on the current device."""

    def __init__(self, codec, n, chunk=4096, device="cuda"):
        import torch
        self.torch = torch
        self.codec, self.n, self.chunk = codec, n, chunk
        self.nch = nchunks(n, chunk)
        self.dev = torch.device(device)
        u8 = torch.uint8
        self.work_bytes = lib().trc_work_bytes(codec, n, chunk)
        if self.work_bytes == 0:
            raise TrcError("bad (codec, n, chunk)")
        self.work = torch.empty(self.work_bytes + PAD, dtype=u8, device=self.dev)
        self.clen = torch.zeros(max(self.nch, 1) + 64, dtype=torch.int32, device=self.dev)
        self.payload = torch.zeros(n + PAD + 64, dtype=u8, device=self.dev)
        self.total = torch.zeros(2, dtype=torch.int64, device=self.dev)
        self.cdf = torch.zeros(264, dtype=torch.int16, device=self.dev)
        self.status = torch.zeros(4, dtype=torch.int32, device=self.dev)
        self.cdfnum = 0
        self.tables_ready = 0                                  # TABLES_READY once trc_tables_dev ran for the current CDF

    def _stream(self):
        return self.torch.cuda.current_stream(self.dev).cuda_stream

    def _tables(self):
        """derive the coder tables from the current CDF once (enqueued), instead of at every encode/decode"""
        if self.codec in STATIC:
            _chk(lib().trc_tables_dev(self.cdf.data_ptr(), self.cdfnum, self.work.data_ptr(), self.work_bytes, self._stream()))
            self.tables_ready = TABLES_READY

    def set_cdf(self, cdf_np, cdfnum):
        t = self.torch.from_numpy(np.ascontiguousarray(cdf_np[:cdfnum + 1]).view(np.int16))
        self.cdf[:cdfnum + 1].copy_(t)
        self.cdfnum = cdfnum
        self._tables()

    def cdfini(self, d_in, n, cdfnum):
        """Device cdfini: histogram of d_in[:n] -> self.cdf (stays on device)."""
        _chk(lib().trc_cdfini_dev(d_in.data_ptr(), n, self.cdf.data_ptr(), cdfnum, self.status.data_ptr(),
                                  self.work.data_ptr(), self._stream()))
        self.cdfnum = cdfnum
        self._tables()

    def hist(self, d_in, n, d_hist):
        """byte histogram of d_in[:n] into d_hist (int64[256] device tensor)"""
        _chk(lib().trc_hist_dev(d_in.data_ptr(), n, d_hist.data_ptr(), self._stream()))

    def cdf_from_hist(self, d_hist, n_total, cdfnum):
        _chk(lib().trc_cdf_from_hist_dev(d_hist.data_ptr(), n_total, self.cdf.data_ptr(), cdfnum, self.status.data_ptr(), self._stream()))
        self.cdfnum = cdfnum
        self._tables()

    def encode(self, d_in, n=None):
        """Enqueue encode of d_in[:n]; results in self.clen / self.payload / self.total (device)."""
        n = self.n if n is None else n
        st = self.codec in STATIC
        _chk(lib().trc_encode_dev(self.codec | self.tables_ready, d_in.data_ptr(), n, self.chunk,
                                  self.cdf.data_ptr() if st else None, self.cdfnum if st else 0,
                                  self.clen.data_ptr(), self.payload.data_ptr(), self.total.data_ptr(),
                                  self.work.data_ptr(), self.work_bytes, self._stream()))

    def decode(self, d_out, n=None, clen=None, payload=None, dir_ready=False):
        """dir_ready: the workspace still holds the group sums of `clen` (the encode or decode just before this call
        was for the same directory): TRC_DIR_READY, include/trc_hip.h"""
        n = self.n if n is None else n
        st = self.codec in STATIC
        clen = self.clen if clen is None else clen
        payload = self.payload if payload is None else payload
        _chk(lib().trc_decode_dev(self.codec | self.tables_ready | (DIR_READY if dir_ready else 0), clen.data_ptr(), payload.data_ptr(), n, self.chunk,
                                  self.cdf.data_ptr() if st else None, self.cdfnum if st else 0,
                                  d_out.data_ptr(), self.work.data_ptr(), self.work_bytes, self._stream()))

    def result(self, n=None):
        """Synchronise and fetch (clen[nchunks] u32, payload bytes) to the host."""
        n = self.n if n is None else n
        nch = nchunks(n, self.chunk)
        self.torch.cuda.synchronize(self.dev)
        tot = int(self.total[0].item())
        clen = self.clen[:nch].cpu().numpy().view(np.uint32).copy()
        payload = self.payload[:tot].cpu().numpy().copy()
        return clen, payload


# ------------------------------------------------------ reference-signature layer (host pointers) ---
_HOST_ENC = {ANS4S: "anscdf4senc", RCS1: "rccdfsenc", RCS2: "rccdfs2enc", RCA: "rccdfenc", ANSA: "anscdfenc", RCB: "rcsenc", RCAI: "rccdfienc",
             RCA4: "rccdf4enc", RCAI4: "rccdf4ienc", ANSA4: "anscdf4enc", RCSM: "rccdfsmenc", ANSO1: "anscdf1enc", ANSB: "ansbc",
             VLCU16: "rccdfuenc16", VLCU32: "rccdfuenc32", VLCV16: "rccdfvenc16", VLCV32: "rccdfvenc32", VLCVZ16: "rccdfvzenc16", VLCVZ32: "rccdfvzenc32",
             VLAU16: "anscdfuenc16", VLAUZ16: "anscdfuzenc16", VLAV16: "anscdfvenc16", VLAVZ16: "anscdfvzenc16", VLAV32: "anscdfvenc32", VLAVZ32: "anscdfvzenc32",
             RCV8: "rccdfenc8", RCVI8: "rccdfienc8"}
_HOST_DEC = {ANS4S: "anscdf4sdec", RCS1: "rccdfsbdec", RCS2: "rccdfsb2dec", RCA: "rccdfdec", ANSA: "anscdfdec", RCB: "rcsdec", RCAI: "rccdfidec",
             RCA4: "rccdf4dec", RCAI4: "rccdf4idec", ANSA4: "anscdf4dec", RCSM: "rccdfsmbdec", ANSO1: "anscdf1dec", ANSB: "ansbd",
             VLCU16: "rccdfudec16", VLCU32: "rccdfudec32", VLCV16: "rccdfvdec16", VLCV32: "rccdfvdec32", VLCVZ16: "rccdfvzdec16", VLCVZ32: "rccdfvzdec32",
             VLAU16: "anscdfudec16", VLAUZ16: "anscdfuzdec16", VLAV16: "anscdfvdec16", VLAVZ16: "anscdfvzdec16", VLAV32: "anscdfvdec32", VLAVZ32: "anscdfvzdec32",
             RCV8: "rccdfdec8", RCVI8: "rccdfidec8"}


def _host_fn(name, codec):
    f = getattr(lib(), name)
    f.restype = _sz
    if codec == ANS4S:
        f.argtypes = [_u8p, _sz, _u8p, _u16p]
    elif codec in (RCS1, RCS2, RCSM):
        f.argtypes = [_u8p, _sz, _u8p, _u16p, C.c_uint]
    else:
        f.argtypes = [_u8p, _sz, _u8p]
    return f


def host_encode(codec, data, cdf=None, cdfnum=256, name=None):
    """Call the reference-named encoder with host pointers -> np.uint8 array of the returned length."""
    data = np.ascontiguousarray(data, dtype=np.uint8)
    n = data.size
    out = np.zeros(n + n // 3 + 1024, dtype=np.uint8)          # the harness's OSIZE (turborc.c:418)
    f = _host_fn(name or _HOST_ENC[codec], codec)
    pin, pout = data.ctypes.data_as(_u8p), out.ctypes.data_as(_u8p)
    if codec == ANS4S:
        l = f(pin, n, pout, cdf.ctypes.data_as(_u16p))
    elif codec in (RCS1, RCS2, RCSM):
        l = f(pin, n, pout, cdf.ctypes.data_as(_u16p), cdfnum)
    else:
        l = f(pin, n, pout)
    if l == 0 and n != 0:
        raise TrcError(lib().trc_last_error().decode())
    return out[:l].copy()


def host_decode(codec, comp, n, cdf=None, cdfnum=256, name=None):
    comp = np.ascontiguousarray(comp, dtype=np.uint8)
    if comp.size == n:
        return comp.copy()                                      # CCPY rule (turborc.c:434)
    src = np.zeros(comp.size + 1024, dtype=np.uint8); src[:comp.size] = comp
    out = np.full(n + 64, 0xA5, dtype=np.uint8)
    f = _host_fn(name or _HOST_DEC[codec], codec)
    pin, pout = src.ctypes.data_as(_u8p), out.ctypes.data_as(_u8p)
    if codec == ANS4S:
        l = f(pin, n, pout, cdf.ctypes.data_as(_u16p))
    elif codec in (RCS1, RCS2, RCSM):
        l = f(pin, n, pout, cdf.ctypes.data_as(_u16p), cdfnum)
    else:
        l = f(pin, n, pout)
    if l != n:
        raise TrcError(lib().trc_last_error().decode())
    return out[:n].copy()


def set_devices(devs):
    """devices of the host-pointer calls ([] = the caller's current device; an entry may repeat: include/trc_hip.h)"""
    arr = (C.c_int * max(len(devs), 1))(*devs)
    _chk(lib().trc_set_devices(arr, len(devs)))


def host_cdfini(data, cdfnum=None):
    data = np.ascontiguousarray(data, dtype=np.uint8)
    if cdfnum is None:
        cdfnum = int(data.max()) + 1
    cdf = np.zeros(257, dtype=np.uint16)
    f = lib().cdfini
    f.restype = C.c_int; f.argtypes = [_u8p, _sz, _u16p, C.c_uint]
    r = f(data.ctypes.data_as(_u8p), data.size, cdf.ctypes.data_as(_u16p), cdfnum)
    return r, cdf, cdfnum


def parse_container(buf):
    """-> dict(hdr fields), clen (u32 array), payload (u8 array)"""
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    magic, codec, ver, cdfnum, chunk, nch = np.frombuffer(buf[:16].tobytes(), dtype="<u4,u1,u1,<u2,<u4,<u4")[0]
    n, pay = np.frombuffer(buf[16:32].tobytes(), dtype="<u8")
    clen = buf[HDR:HDR + 4 * int(nch)].view("<u4").copy()
    payload = buf[HDR + 4 * int(nch):HDR + 4 * int(nch) + int(pay)].copy()
    return dict(magic=int(magic), codec=int(codec), version=int(ver), cdfnum=int(cdfnum), chunk=int(chunk),
                nchunks=int(nch), n=int(n), payload=int(pay)), clen, payload
