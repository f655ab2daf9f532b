#!/bin/bash
# round 5: static range decoders -- symbol from the estimated quotient, verified against its bounds; parity (incl. corrupt input), A/B vs the exact quotient
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "rccdfs or total_parity or alias or corrupt or alignment or shapes or golden or host_pointer" > gpurun_out/r05r_parity.log 2>&1; tail -2 gpurun_out/r05r_parity.log
for rep in 1 2 3; do for v in exactq main; do
  if [ "$v" = "main" ]; then unset TRC_LIB; else export TRC_LIB=$GRAFT_REPO_ROOT/turbo-range-coder_amd/build/ab/lib$v.so; fi
  echo "--- variant $v (rep $rep)"
  bash scripts/gpu_codec_sweep.sh "rccdfs" "512 4096"; bash scripts/gpu_codec_sweep.sh "rccdfs2" "1024"
done; done 2>&1 | tee gpurun_out/r05r_ab.txt
