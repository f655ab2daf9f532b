#!/bin/bash
# round 4: order-1 coder with the two-wave model pass + planar four-lane coding pass
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "anscdf1" 2>&1 | tail -3
for m in 1 0; do echo "TRC_O1_MC=$m"; TRC_O1_MC=$m bash scripts/gpu_codec_sweep.sh "anscdf1" "4096 2048 1024"; done
bash scripts/gpu_kstats.sh r4_o1 --codec anscdf1 --no-beyond
} > gpurun_out/r04_o1.log 2>&1
cut -c1-200 gpurun_out/r04_o1.log
