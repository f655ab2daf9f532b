#!/bin/bash
# slice size of host-pointer calls, the remaining coder families
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
out=gpurun_out/r03_slice2.log; : > $out
for sl in 16777216 33554432 67108864; do
  echo "== TRC_HOST_SLICE=$sl page-locked" >> $out
  TRC_HOST_SLICE=$sl ./harness/trcbench -e 44,47,48,49 --text 100000000 --pin 2>&1 | grep -v "^synthetic\|C Size" >> $out
  TRC_HOST_SLICE=$sl ./harness/trcbench --nibble 100000000 --pin 2>&1 | grep -v "^synthetic\|C Size" >> $out
  TRC_HOST_SLICE=$sl ./harness/trcbench -e 50,52,53,62,63 --int32 100000000 --pin 2>&1 | grep -v "^synthetic\|C Size" >> $out
  TRC_HOST_SLICE=$sl ./harness/trcbench -e 50,52,53,60,61,62,63 --int16 100000000 --pin 2>&1 | grep -v "^synthetic\|C Size" >> $out
done
for sl in 16777216 67108864 268435456; do
  echo "== order-1, TRC_HOST_SLICE=$sl" >> $out
  TRC_HOST_SLICE=$sl ./harness/trcbench -e 64 --text 100000000 --pin 2>&1 | grep -v "^synthetic\|C Size" >> $out
done
cat $out
