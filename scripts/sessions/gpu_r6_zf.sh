#!/bin/bash
# round 6, session zf: the order-1 decoder with four lanes per chunk (TRC_O1_ROWS=4): parity, then kernel times next to the eight-lane form
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r06zf_o1_rows4.txt; : > $out
TRC_O1_ROWS=4 timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "order1 or anscdf1 or o1 or golden or total or shape" 2>&1 | tail -5 >> $out
for r in 4 1; do
  echo "== TRC_O1_ROWS=$r" >> $out
  TRC_O1_ROWS=$r bash scripts/gpu_kstats.sh zf_$r --codec anscdf1 --no-beyond --no-configs --no-host 2>&1 | grep -E "o1_dec|value" | cut -c1-150 >> $out
done
cat $out
