#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "anscdf1 or fuzz" 2>&1 | tail -4
bash scripts/gpu_ab.sh "main v_old" "anscdf1" "4096 2048 1024" 1
