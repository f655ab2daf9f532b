#!/bin/bash
# host-pointer layer after the per-coder slice / chunk policy: every coder family through the reference-named calls
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
out=gpurun_out/r03_host2.log; : > $out
python -m pytest tests/test_zz_gpu_harness.py -x -q -k "slice_plan or c_harness or file_tool" 2>&1 | tail -3 >> $out
python -m pytest tests/test_gpu_parity.py -x -q -k "host" 2>&1 | tail -3 >> $out
for pin in "" "--pin"; do
  echo "== 100 MB ${pin:-pageable}" >> $out
  ./harness/trcbench -e 1,42,44,45,46,47,48,49,56,64,65,66,79 --text 100000000 $pin 2>&1 >> $out
  ./harness/trcbench --nibble 100000000 -e 46,47,56 $pin 2>&1 >> $out
  ./harness/trcbench -e 50,52,53,62,63 --int32 100000000 $pin 2>&1 >> $out
  ./harness/trcbench -e 50,52,53,60,61,62,63 --int16 100000000 $pin 2>&1 >> $out
done
echo "== 1 GB page-locked" >> $out
./harness/trcbench -e 1,46,56,65 --text 1000000000 --pin 2>&1 >> $out
cat $out
