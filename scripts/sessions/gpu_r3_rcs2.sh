#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "rccdfs2 or fuzz" 2>&1 | tail -6
for rep in 1 2; do
echo "--- pair lanes"; bash scripts/gpu_codec_sweep.sh "rccdfs2" "512 896 1024 2048"
echo "--- one lane (TRC_RCS2_PAIR=0)"; TRC_RCS2_PAIR=0 bash scripts/gpu_codec_sweep.sh "rccdfs2" "512 896 1024 2048"
done
bash scripts/gpu_codec_sweep.sh "rccdfs" "512 1024"
