#!/bin/bash
# round 5: headline decoder start-up (two-step prime, directory loads hoisted) -- parity, in-kernel clocks, A/B against round 4
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "anscdf4s or static_rans or total_parity or corrupt or mixed_raw" > gpurun_out/r05b_parity.log 2>&1; tail -3 gpurun_out/r05b_parity.log
TRC_LIB=$GRAFT_REPO_ROOT/turbo-range-coder_amd/build/ab/libprof.so python bench.py --no-cpu --no-beyond --steps 64 --warmup 5 2> gpurun_out/r05b_decprof.txt > /dev/null; grep "dec prof" gpurun_out/r05b_decprof.txt | tail -2
for rep in 1 2 3; do for v in r4base main; do
  if [ "$v" = "main" ]; then unset TRC_LIB; else export TRC_LIB=$GRAFT_REPO_ROOT/turbo-range-coder_amd/build/ab/lib$v.so; fi
  echo "--- variant $v (rep $rep)"
  bash scripts/gpu_codec_sweep.sh "anscdf4s" "512 1024 4096"
done; done 2>&1 | tee gpurun_out/r05b_ab.txt
