"""measuring aid: does keeping two steps in flight (two contexts, two streams) hide the gather and the launch gaps?"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "turbo-range-coder_amd"), os.path.join(ROOT, "tests")]
import trc, trc_testlib as T
n, chunk, dev = 100 * 1000 * 1000, int(sys.argv[1]) if len(sys.argv) > 1 else 512, torch.device("cuda", 0)
d = T.text_bytes(n, 7)
d_in = torch.from_numpy(np.concatenate([d, np.zeros(512, np.uint8)])).to(dev)
for nctx in (1, 2, 3):
    dcs = [trc.DeviceCoder(trc.ANS4S, n, chunk, dev) for _ in range(nctx)]
    outs = [torch.zeros(n + 512, dtype=torch.uint8, device=dev) for _ in range(nctx)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(nctx)]
    for dc in dcs:
        dc.cdfini(d_in, n, 256)
    torch.cuda.synchronize()
    def run(K):
        for k in range(K):
            i = k % nctx
            with torch.cuda.stream(streams[i]):
                dcs[i].encode(d_in, n); dcs[i].decode(outs[i], n, dir_ready=True)
    run(6); torch.cuda.synchronize()
    assert all(torch.equal(o[:n], d_in[:n]) for o in outs)
    t0 = time.perf_counter(); run(40); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("contexts/streams %d: %.4f ms per step, %.0f MB/s" % (nctx, dt / 40 * 1e3, n * 40 / dt / 1e6))
