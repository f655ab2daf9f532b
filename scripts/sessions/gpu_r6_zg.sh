#!/bin/bash
# round 6, session zg: the order-1 decoder with two / four lanes per chunk: parity under each forced form, then time against input size
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r06zg_o1_rowsn.txt; : > $out
for r in 2 4; do
  echo "== parity, TRC_O1_ROWS=$r" >> $out
  TRC_O1_ROWS=$r timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "order1 or anscdf1 or o1 or golden or total or shape" 2>&1 | tail -3 >> $out
done
for r in 2 4 1; do echo "== TRC_O1_ROWS=$r" >> $out; TRC_O1_ROWS=$r timeout 250 python scripts/probe/o1_dec_sizes.py 2>&1 | grep -v amdgpu.ids >> $out; done
cat $out
