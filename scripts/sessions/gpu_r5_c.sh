#!/bin/bash
# round 5: do the waves of a SIMD end together when they keep each other's pace (s_setprio)?
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for v in prof2 bal1p bal2p; do TRC_LIB=$GRAFT_REPO_ROOT/turbo-range-coder_amd/build/ab/lib$v.so python bench.py --no-cpu --no-beyond --steps 64 --warmup 5 2> gpurun_out/r05d_$v.txt | tail -1 | cut -c1-160; grep "dec wall" gpurun_out/r05d_$v.txt | tail -2; done
for rep in 1 2; do for v in main bal1 bal2; do
  if [ "$v" = "main" ]; then unset TRC_LIB; else export TRC_LIB=$GRAFT_REPO_ROOT/turbo-range-coder_amd/build/ab/lib$v.so; fi
  echo "--- variant $v (rep $rep)"
  bash scripts/gpu_codec_sweep.sh "anscdf4s" "512 1024"
done; done 2>&1 | tee gpurun_out/r05d_ab.txt
