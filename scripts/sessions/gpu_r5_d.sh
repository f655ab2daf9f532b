#!/bin/bash
# round 5: pace-keeping policies of the headline decoder (TRC_DEC_BALANCE variants)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for v in bal4p; do TRC_LIB=$GRAFT_REPO_ROOT/turbo-range-coder_amd/build/ab/lib$v.so python bench.py --no-cpu --no-beyond --steps 64 --warmup 5 2> gpurun_out/r05e_$v.txt | tail -1 | cut -c1-160; grep "dec wall" gpurun_out/r05e_$v.txt | tail -2; done
for rep in 1 2 3; do for v in main bal1 bal2 bal3 bal4 bal5; do
  if [ "$v" = "main" ]; then unset TRC_LIB; else export TRC_LIB=$GRAFT_REPO_ROOT/turbo-range-coder_amd/build/ab/lib$v.so; fi
  echo "--- variant $v (rep $rep)"
  bash scripts/gpu_codec_sweep.sh "anscdf4s" "512"
done; done 2>&1 | tee gpurun_out/r05e_ab.txt
unset TRC_LIB
for v in bal2 bal4; do TRC_LIB=$GRAFT_REPO_ROOT/turbo-range-coder_amd/build/ab/lib$v.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "anscdf4s or static_rans or total_parity" 2>&1 | tail -2; done
