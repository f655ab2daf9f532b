"""GPU (-m gpu): the static rANS encoder that gathers its own payload (csrc/trc_gather.h, round 5).

The twelve-wave encoder workgroups of a one-round launch take tickets, publish their byte totals in the sync area of the workspace,
poll the totals of the lower tickets and move their own pieces to the container -- no gather kernel behind them.  What can go wrong
is what these cases aim at: the sync area reused call after call (it must be zero again after every launch), workgroups whose last
waves have no chunks, raw (incompressible) chunks among coded ones, one workgroup / a few / a chip full of them, a second stream
keeping the device busy while the workgroups wait for each other.  Everything is compared byte for byte with the oracle's chunked
encode (lengths, payload, total) and with the same call under TRC_ENC_FUSED=0 (the gather kernel); decodes return the input.
Each environment runs in a process of its own (the switches are read once)."""
import os
import subprocess
import sys
import textwrap

import pytest

import trc

pytestmark = pytest.mark.gpu

CODE = textwrap.dedent("""
    import sys, hashlib, numpy as np, torch
    sys.path[:0] = [%r, %r]
    import trc, trc_testlib as T
    from golden.make_golden import gen
    def check(d, chunk, reps=3, busy=False):
        n = d.size
        _, cdf, cdfnum = T.orc_cdfini(d)
        ep, ec, _ = T.orc_chunked_enc(trc.ANS4S, d, chunk, cdf, cdfnum)
        dc = trc.DeviceCoder(trc.ANS4S, n, chunk, "cuda:0")
        dc.work.fill_(0xA5)                                       # whatever the allocator handed out: the table prep must leave the sync area zero
        dc.set_cdf(cdf, cdfnum)
        d_in = torch.from_numpy(np.concatenate([d, np.zeros(512, np.uint8)])).to("cuda:0")
        side = None
        if busy:                                                  # another coder's launches on a second stream while the workgroups wait for each other
            side = torch.cuda.Stream()
            d2 = torch.from_numpy(gen("text", 1 << 22, 3)).to("cuda:0")
            dc2 = trc.DeviceCoder(trc.RCB, 1 << 22, 1536, "cuda:0")
            with torch.cuda.stream(side):
                for _ in range(6): dc2.encode(d2, 1 << 22)
        for rep in range(reps):                                   # the second and third call find the sync area as the call before left it
            dc.payload.fill_(0x5A); dc.total.fill_(-1)
            dc.encode(d_in, n)
            clen, payload = dc.result(n)
            assert np.array_equal(clen, ec), ("clen", n, chunk, rep)
            assert payload.size == ep.size and np.array_equal(payload, ep), ("payload", n, chunk, rep, payload.size, ep.size)
            assert int(dc.payload[ep.size:ep.size + 64].cpu().numpy().max()) == 0x5A and int(dc.payload[ep.size:ep.size + 64].cpu().numpy().min()) == 0x5A, "wrote past the payload"
            out = torch.full((n + 512,), 0xA5, dtype=torch.uint8, device="cuda:0")
            dc.decode(out, n, dir_ready=True); torch.cuda.synchronize()
            o = out.cpu().numpy()
            assert np.array_equal(o[:n], d) and (o[n:] == 0xA5).all(), ("roundtrip", n, chunk, rep)
        if side is not None: side.synchronize()
        sync = dc.work[41984:41984 + 2112].cpu().numpy()
        assert not sync.any(), "sync area not zero after the call"
        return hashlib.sha256(payload.tobytes()).hexdigest()
    mixed = np.concatenate([gen("zipf", 512 * 700, 1), gen("uniform", 512 * 37, 2), gen("const", 512 * 90, 3), gen("uniform", 512 * 3, 4),
                            gen("text", 512 * 64 * 30 + 77, 5)])
    hs = []
    hs.append(check(gen("text", 300001, 11), 512))               # 10 groups: one workgroup, two waves without chunks
    hs.append(check(gen("zipf", 64 * 512 * 41 + 5, 12), 512))    # 42 groups: four workgroups, the last with six waves; ragged last chunk
    hs.append(check(mixed, 512))                                  # raw chunks among coded ones
    hs.append(check(gen("text", 64 * 256 * 12 * 7, 13), 256))     # exactly seven full workgroups
    hs.append(check(gen("uniform", 64 * 512 * 13 + 1, 14), 512))  # nothing compresses: every piece is the input chunk
    hs.append(check(gen("text", 70 * 1000 * 1000, 15), 512, reps=2, busy=True))   # 2137 groups: the shape the library picks by itself at this size
    print("hashes", " ".join(hs))
    print("ok")
""")


def run(env):
    code = CODE % (os.path.dirname(os.path.abspath(trc.__file__)), os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
    assert r.returncode == 0 and "ok" in r.stdout, (env, r.stdout[-2000:] + r.stderr[-3000:])
    return [l for l in r.stdout.splitlines() if l.startswith("hashes")][0]


def test_encoder_gathers_its_own_payload():
    fused = run(dict(TRC_ENC_WPB="12", TRC_ENC_FUSED="1"))
    plain = run(dict(TRC_ENC_WPB="12", TRC_ENC_FUSED="0"))
    assert fused == plain
