#!/bin/bash
# round 6, session z: anscdf1 walk kernel -- waves per workgroup x chunks per workgroup (is it its longest chain at two waves per SIMD?)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r06z_o1_walk.txt; : > $out
for v in base w1g16 w1g24 w1g32 w2g48 w2g64 w4g64 w4g96; do
  lib=turbo-range-coder_amd/build/ab/lib$v.so; [ $v = base ] && lib=turbo-range-coder_amd/libturborc_hip.so
  echo "== $v" >> $out
  TRC_LIB=$PWD/$lib bash scripts/gpu_kstats.sh z_$v --codec anscdf1 --no-beyond --no-configs --no-host 2>&1 | grep -E "o1_|ansa_code|value" | cut -c1-150 >> $out
done
cat $out
