#!/bin/bash
# round 5: TrcPace in the static range coders -- parity, A/B (TRC_RCS_ENC_WPB=1 forces the small-workgroup encoders)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "rccdfs or total_parity or alias or mixed_raw" > gpurun_out/r05g_parity.log 2>&1; tail -2 gpurun_out/r05g_parity.log
for rep in 1 2 3; do for v in nobal main; do
  if [ "$v" = "main" ]; then unset TRC_LIB TRC_RCS_ENC_WPB; else export TRC_LIB=$GRAFT_REPO_ROOT/turbo-range-coder_amd/build/ab/lib$v.so; fi
  echo "--- variant $v (rep $rep)"
  bash scripts/gpu_codec_sweep.sh "rccdfs rccdfsm" "512"; bash scripts/gpu_codec_sweep.sh "rccdfs2" "1024"
done; done 2>&1 | tee gpurun_out/r05g_ab.txt
