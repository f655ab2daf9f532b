#!/bin/bash
# round 3, decoder work: parity subset on the main build, then A/B of the variants on the headline and the static range coders
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "anscdf4s or rccdfs or cdfini or config5 or buckets or golden" 2>&1 | tail -5
bash scripts/gpu_ab.sh "${1:-main v_old v_lean v_dpp}" "anscdf4s" "512 1024" 2
bash scripts/gpu_ab.sh "main v_old" "rccdfs rccdfs2 rccdfsm" "512" 1
