#!/bin/bash
# round 5: the nibble coders in 12-wave workgroups with TrcPace -- parity (all kernels on the shared prologue), A/B via TRC_NIB_BIG
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "shapes or rccdf or anscdf or ansb or rcs or nibble or kernel_forms" > gpurun_out/r05q_parity.log 2>&1; tail -2 gpurun_out/r05q_parity.log
for rep in 1 2 3; do for b in 0 1; do export TRC_NIB_BIG=$b; echo "--- TRC_NIB_BIG=$b (rep $rep)"; bash scripts/gpu_codec_sweep.sh "rccdf4 rccdf4i anscdf4" "512"; done; done 2>&1 | tee gpurun_out/r05q_ab.txt
