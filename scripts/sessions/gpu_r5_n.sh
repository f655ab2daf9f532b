#!/bin/bash
# round 5: the encoder's staged payload written with the nontemporal hint (no end-of-kernel write-back of dirty L2 lines?) -- A/B incl. the gather behind it
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
TRC_LIB=$GRAFT_REPO_ROOT/turbo-range-coder_amd/build/ab/libdrainnt_prof.so python bench.py --no-cpu --no-beyond --steps 64 --warmup 5 2> gpurun_out/r05_drainnt_prof.txt | tail -1 | cut -c1-120; grep "enc wall" gpurun_out/r05_drainnt_prof.txt | tail -1
for rep in 1 2 3; do for v in main drainnt; do
  if [ "$v" = "main" ]; then unset TRC_LIB; else export TRC_LIB=$GRAFT_REPO_ROOT/turbo-range-coder_amd/build/ab/lib$v.so; fi
  echo "--- variant $v (rep $rep)"
  bash scripts/gpu_codec_sweep.sh "anscdf4s rccdfs" "512"; bash scripts/gpu_codec_sweep.sh "rccdfs2" "1024"
done; done 2>&1 | tee gpurun_out/r05o_ab.txt
for v in main drainnt; do
  if [ "$v" = "main" ]; then unset TRC_LIB; else export TRC_LIB=$GRAFT_REPO_ROOT/turbo-range-coder_amd/build/ab/lib$v.so; fi
  echo "--- kstats $v"; bash scripts/gpu_kstats.sh r5o_$v --no-beyond 2>&1 | head -5
done
