"""Per-wave clocks of trc_o1_walk_kernel (a -DTRC_O1W_PROF build selected with TRC_LIB): start / loop start / end on the 100 MHz wall clock
and the rounds each wave made.  usage: TRC_LIB=.../libNAME.so python scripts/probe/o1w_prof.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "turbo-range-coder_amd")]
import trc  # noqa: E402
import trc_testlib as T  # noqa: E402

n, chunk = 100 * 1000 * 1000, 4096
d = T.drift_bytes(n, 3)
dc = trc.DeviceCoder(trc.ANSO1, n, chunk, "cuda:0")
d_in = torch.from_numpy(np.concatenate([d, np.zeros(512, np.uint8)])).to("cuda:0")
for _ in range(3):
    dc.encode(d_in, n)
torch.cuda.synchronize()
lib = trc.lib()
buf = np.zeros(4 * 4096, dtype=np.uint64)
lib.trc_o1w_prof_read.argtypes = [C.c_void_p, C.c_size_t]
r = lib.trc_o1w_prof_read(buf.ctypes.data, buf.nbytes)
assert r == 0, r
p = buf.reshape(4096, 4)
p = p[p[:, 0] != 0]
t0 = p[:, 0].min()
st, ls, en, rd = (p[:, 0] - t0) / 100.0, (p[:, 1] - t0) / 100.0, (p[:, 2] - t0) / 100.0, p[:, 3].astype(np.int64)
q = lambda a: " ".join("%8.1f" % np.percentile(a, x) for x in (0, 10, 50, 90, 100))
print("waves %d" % len(p))
print("start us      (min p10 p50 p90 max): " + q(st))
print("loop start us                      : " + q(ls))
print("end us                             : " + q(en))
print("rounds                             : " + q(rd))
print("us per round (loop)                : " + q((en - ls) / np.maximum(rd, 1)))
o = np.argsort(en)[-8:]
print("last waves: end us / rounds / us per round: " + "; ".join("%.0f/%d/%.2f" % (en[i], rd[i], (en[i] - ls[i]) / rd[i]) for i in o))
