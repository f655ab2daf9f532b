"""bench.py --gpus N started by hand launches its own N ranks: the launch is asserted on the CPU
with --dry-launch, and a 2-rank launch is run for real with a stub in place of the GPU work (the launcher's rc handling)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    e.update(kw)
    return e


def test_dry_launch_command_and_environment():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "20", "--warmup", "5", "--workload", "zipf1g", "--dry-launch"],
                       capture_output=True, text=True, env=_env(), timeout=120)
    assert r.returncode == 0, r.stderr
    j = json.loads(r.stdout.strip().splitlines()[-1])
    cmd = j["launch"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(BENCH)
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5", "--workload", "zipf1g"]      # flags pass through, --dry-launch does not
    assert j["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # BASELINE config 5 on 8 GPUs: what one rank allocates (1 GB shard; two banks of 8 result sets, two banks of 7 receive
    # sets, the coder's scratch) must fit a 288 GB GPU with a wide margin
    b = j["hbm_bytes_per_rank"]
    assert b["chunk"] == 5120 and b["receive_banks"] == 2 * 7 * (4 * 195313 + 10**9 + 1024)      # (round 5: one residency round of chunk 5120, not two of 2560)
    assert 30e9 < b["total"] < 40e9, b


def test_dry_launch_honours_master_port():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-launch"], capture_output=True, text=True, env=_env(MASTER_PORT="29611"), timeout=120)
    cmd = json.loads(r.stdout.strip().splitlines()[-1])["launch"]
    assert cmd[cmd.index("--master-port") + 1] == "29611"


def test_no_self_launch_inside_a_job_or_with_one_gpu():
    # under torch.distributed.run (WORLD_SIZE set) bench.py must not launch again; with one GPU there is nothing to launch
    for args, env in ((["--gpus", "2"], _env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")), (["--gpus", "1"], _env())):
        r = subprocess.run([sys.executable, BENCH] + args + ["--dry-launch"], capture_output=True, text=True, env=env, timeout=120)
        assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["launch"] is None, r.stdout + r.stderr


def test_self_launch_runs_the_ranks_and_reports_failure(tmp_path):
    """the real launcher path on the CPU: no GPU here, so every rank stops at bench.py's own 'needs a GPU' assertion --
    what is checked is that two ranks were started under torch.distributed.run and that their failure is the launcher's rc."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, env=_env(), timeout=300)
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present: the launch would run the real bench")
    assert r.returncode != 0
    assert "bench.py needs a GPU" in r.stderr and "local_rank: 1" in r.stderr.replace("local_rank : 1", "local_rank: 1"), r.stderr[-3000:]
