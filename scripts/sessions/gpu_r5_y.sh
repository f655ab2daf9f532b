#!/bin/bash
# round 5 (second session): the static rANS encoder gathering its own payload (trc_gather.h): parity first, then A/B against the gather kernel
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fused_gather.py -x -q 2>&1 | tail -15
for rep in 1 2 3; do for f in 0 1; do
  echo "== TRC_ENC_FUSED=$f"
  TRC_ENC_FUSED=$f timeout 300 python bench.py --no-cpu --no-beyond --no-configs 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print({k: j.get(k) for k in ('value', 'ms_per_step', 'value_cold_clocks', 'value_cold', 'enc_kernel_ms', 'dec_kernel_ms', 'payload_matches_reference_sha256')})"
done; done
TRC_ENC_FUSED=1 bash scripts/gpu_kstats.sh fused1 --no-beyond --no-configs
TRC_ENC_FUSED=0 bash scripts/gpu_kstats.sh fused0 --no-beyond --no-configs
