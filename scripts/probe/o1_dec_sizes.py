"""Order-1 decoder kernel time against input size (how much two waves sharing a SIMD cost each other): drift data, chunk 4096.
usage: [TRC_O1_ROWS=4|1] python scripts/probe/o1_dec_sizes.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "turbo-range-coder_amd")]
import trc  # noqa: E402
import trc_testlib as T  # noqa: E402

full = T.drift_bytes(100 * 1000 * 1000, 3)
rows = int(os.environ.get("TRC_O1_ROWS", "0"))
per_wave = {1: 8, 4: 16, 2: 32, 16: 16, 8: 8, 64: 64}.get(rows, 0)      # chunks per wave of the forced form (0: the default picks by size)
for mb in (25, 50, 67, 100):
    n = mb * 1000 * 1000
    d = full[:n]
    dc = trc.DeviceCoder(trc.ANSO1, n, 4096, "cuda:0")
    d_in = torch.from_numpy(np.concatenate([d, np.zeros(512, np.uint8)])).to("cuda:0")
    out = torch.zeros(n + 512, dtype=torch.uint8, device="cuda:0")
    dc.encode(d_in, n)
    ts = []
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); dc.decode(out, n, dir_ready=True); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    assert torch.equal(out[:n], d_in[:n])
    nch = (n + 4095) // 4096
    waves = ("%5d waves of %d chunks" % ((nch + per_wave - 1) // per_wave, per_wave)) if per_wave else "default form"
    print("%4d MB: %6d chunks, %s  decode %.3f ms (min of 4; %s)" % (mb, nch, waves, min(ts), " ".join("%.3f" % t for t in ts)))
