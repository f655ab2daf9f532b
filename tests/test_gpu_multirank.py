"""GPU (-m gpu): the device-side schedule of `bench.py --gpus N` against REAL peer processes on one GPU.

RCCL cannot put two ranks on one device, and no 8-GPU node was available in any round, so before this test the N > 1 device path
(shard.StepPipeline on CudaRuntime: coder kernels on the main stream, grouped exchanges on a side stream, events between them, two
banks of CUDA result buffers, preallocated CUDA receive buffers, rotating roots) had only ever run with world = 1 or with CPU
tensors.  `bench.py --backend gloo --same-device` runs it with N ranks on cuda:0: gloo carries the bytes (device tensors staged
through host memory by shard.StagedDist), everything else is the code an 8-GPU job runs.  `--verify-gather`: after the timed
region two full groups and a ragged one, every root hashing each piece it received against the SHA-256 its owner computed."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU (and must not silently fall back)"
    return torch


def run_bench(world, extra, env=None, timeout=900):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--backend", "gloo", "--same-device", "--verify-gather",
           "--no-cpu", "--clock-warmup-ms", "0", "--warmup", "1"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=dict(os.environ, **(env or {})), cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("world,codec,schedule", [(2, "anscdf4s", "rotate"), (4, "anscdf4s", "rotate"), (2, "anscdf4s", "root0"), (4, "rccdf", "rotate"), (3, "rcs", "root0")])
def test_ranks_on_one_device_gather_whole_results(torch_cuda, world, codec, schedule):
    j = run_bench(world, ["--codec", codec, "--steps", str(2 * world + 1), "--size", "20000000"], {"TRC_BENCH_EXCHANGE": schedule})
    assert j["n_gpus"] == world and j["config"]["exchange_schedule"] == schedule and j["config"]["ranks_share_device"] is True
    g = j["gather_check"]
    assert g["ok"] and g["mismatches"] == 0
    G = world if schedule == "rotate" else 1
    assert g["steps_run"] == g["steps_checked_on_their_roots"] == 2 * G + (1 if G > 1 else 0)
    assert g["pieces_hashed"] == g["steps_run"] * world


def test_config5_shard_size_with_a_peer(torch_cuda):
    """BASELINE config 5's per-GPU shard (10^9 Zipf bytes generated on the device) with two ranks: the 0.7 GB payloads of every step
    cross between the ranks' CUDA buffers; sampled chunks of rank 0 against the oracle as in the single-rank run"""
    j = run_bench(2, ["--workload", "zipf1g", "--steps", "2"], timeout=1500)
    assert j["n_gpus"] == 2 and j["gather_check"]["ok"] and j["gather_check"]["pieces_hashed"] == 5 * 2
    assert j.get("oracle_checked_chunks", 0) >= 50
