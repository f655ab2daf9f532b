#!/bin/bash
# round 5: one 16-wave coding-pass workgroup per CU (LDS request padded) -- is the anscdf / ansb encode still bimodal?  parity of the touched coders
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "anscdf or ansb or rccdf" > gpurun_out/r05n_parity.log 2>&1; tail -1 gpurun_out/r05n_parity.log
for rep in 1 2 3 4 5 6; do bash scripts/gpu_codec_sweep.sh "anscdf ansb" "1536"; done 2>&1 | tee gpurun_out/r05n_var.txt
bash scripts/gpu_codec_sweep.sh "anscdf1" "4096"; bash scripts/gpu_codec_sweep.sh "rccdf rcs" "1536"
