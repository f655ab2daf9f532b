#!/bin/bash
# full parity suite on the main build, then A/B timing of the listed variants on the headline and the static range coders
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -5
bash scripts/gpu_ab.sh "${1:-main v_old}" "anscdf4s" "512 1024 4096" 2
bash scripts/gpu_ab.sh "${1:-main v_old}" "rccdfs rccdfs2 rccdfsm anscdf" "512" 1
