#!/bin/bash
# round 6, session ze: walk kernel A/B -- K rows requested a window ahead, the compiler told it has the SIMD to itself
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r06ze_o1_walk_ab.txt; : > $out
for v in base $VARIANTS; do
  lib=turbo-range-coder_amd/build/ab/lib$v.so; [ $v = base ] && lib=turbo-range-coder_amd/libturborc_hip.so
  echo "== $v" >> $out
  TRC_LIB=$PWD/$lib bash scripts/gpu_kstats.sh ze_$v --codec anscdf1 --no-beyond --no-configs --no-host 2>&1 | grep -E "o1_walk|o1_sort|value" | cut -c1-150 >> $out
done
cat $out
