#!/bin/bash
# round 6, session zh: order-1 decoder (four lanes per chunk by default) in the bench loop: renormalisations early / late, three repetitions
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r06zh_o1_dec_ab.txt; : > $out
for rep in 1 2 3; do
for v in base rn0; do
  lib=turbo-range-coder_amd/build/ab/lib$v.so; [ $v = base ] && lib=turbo-range-coder_amd/libturborc_hip.so
  echo "== $v rep $rep" >> $out
  TRC_LIB=$PWD/$lib bash scripts/gpu_kstats.sh zh_$v --codec anscdf1 --no-beyond --no-configs --no-host 2>&1 | grep -E "o1_dec|value" | cut -c1-150 >> $out
done; done
cat $out
