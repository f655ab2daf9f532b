#!/bin/bash
# round 5: order-1 rANS decoder with R chunks per wave (TRC_O1_ROWS = 64 / 16 / 8) -- parity at each, timing
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for r in 16 8 64; do TRC_O1_ROWS=$r timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "anscdf1" 2>&1 | tail -1; done
for rep in 1 2; do for r in 64 16 8; do
  export TRC_O1_ROWS=$r
  echo "--- TRC_O1_ROWS=$r (rep $rep)"
  bash scripts/gpu_codec_sweep.sh "anscdf1" "4096 2048 1024"
done; done 2>&1 | tee gpurun_out/r05j_ab.txt
