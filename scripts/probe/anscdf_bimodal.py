"""Probe (round 5): the anscdf encoder runs 0.59 ms in most processes and 0.68 ms in some (both passes slower together, the decoder
unchanged).  One process: allocate as bench.py does, time the encode, print the addresses of the buffers (modulo 2 MiB / 1 GiB)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "turbo-range-coder_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import trc, trc_testlib as T
n, chunk = 100 * 1000 * 1000, 1536
pre = int(os.environ.get("PROBE_PREALLOC_MB", "0"))
hold = torch.empty(pre << 20, dtype=torch.uint8, device="cuda:0") if pre else None     # shifts where the work buffer lands
d = T.bench_input("drift", n, 1) if hasattr(T, "bench_input") else None
d_in = torch.from_numpy(np.concatenate([d, np.zeros(512, np.uint8)])).to("cuda:0")
dc = trc.DeviceCoder(trc.ANSA, n, chunk, "cuda:0")
for _ in range(20): dc.encode(d_in, n)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): dc.encode(d_in, n)
e1.record(); torch.cuda.synchronize()
p = dc.work.data_ptr()
print("encode %.3f ms  work %#x (mod 2MiB %#x, mod 1GiB %#x)  in %#x  payload %#x" % (e0.elapsed_time(e1) / 50, p, p & 0x1fffff, p & 0x3fffffff, d_in.data_ptr(), dc.payload.data_ptr()))
