"""measuring aid: device histogram (cdfini) time on 100 MB of text"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "turbo-range-coder_amd"), os.path.join(ROOT, "tests")]
import trc, trc_testlib as T
n = 100 * 1000 * 1000
d = T.text_bytes(n, 7)
d_in = torch.from_numpy(np.concatenate([d, np.zeros(512, np.uint8)])).cuda()
dc = trc.DeviceCoder(trc.ANS4S, n, 512, "cuda:0")
hist = torch.zeros(256, dtype=torch.int64, device="cuda:0")
for _ in range(3): dc.hist(d_in, n, hist)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): dc.hist(d_in, n, hist)
b.record(); torch.cuda.synchronize()
assert np.array_equal(hist.cpu().numpy(), np.bincount(d, minlength=256))
print("hist (memset + kernel): %.1f us per 100 MB = %.2f TB/s" % (a.elapsed_time(b) / 20 * 1e3, n / (a.elapsed_time(b) / 20 * 1e-3) / 1e12))
