#!/bin/bash
# round 6, session zi: order-1 decoder, four lanes per chunk and TWO chunks per row (TRC_O1_ROWS=42): parity, time against size, bench loop
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r06zi_o1_rows42.txt; : > $out
TRC_O1_ROWS=42 timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "order1 or anscdf1 or o1 or golden or total or shape" 2>&1 | tail -3 >> $out
for r in 42 4; do echo "== TRC_O1_ROWS=$r" >> $out; TRC_O1_ROWS=$r timeout 250 python scripts/probe/o1_dec_sizes.py 2>&1 | grep -v amdgpu.ids >> $out; done
for r in 42 4; do
  echo "== bench loop, TRC_O1_ROWS=$r" >> $out
  TRC_O1_ROWS=$r bash scripts/gpu_kstats.sh zi_$r --codec anscdf1 --no-beyond --no-configs --no-host 2>&1 | grep -E "o1_dec|value" | cut -c1-150 >> $out
done
cat $out
