#!/bin/bash
# host-pointer calls of the model-bound coders (4 waves per CU resident): does a slice of one full residency round
# (1024 waves x 64 chunks x 512 B = 32 MB) beat the 16 MB default?
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
out=gpurun_out/r03_slice.log; : > $out
for sl in 8388608 16777216 33554432 50331648 67108864; do
  echo "== TRC_HOST_SLICE=$sl page-locked" >> $out
  TRC_HOST_SLICE=$sl ./harness/trcbench -e 46,56,66,1,65 --text 100000000 --pin 2>&1 | grep -v "^synthetic" >> $out
done
echo "== default, 1 GB" >> $out
./harness/trcbench -e 46 --text 1000000000 --pin 2>&1 >> $out
TRC_HOST_SLICE=33554432 ./harness/trcbench -e 46 --text 1000000000 --pin 2>&1 >> $out
cat $out
