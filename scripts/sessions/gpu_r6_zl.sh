#!/bin/bash
# round 6, session zl: order-1 decoders with the first-touch bits read early (next to the early table load): parity under the forced forms, bench loop A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r06zl_o1_seen_early.txt; : > $out
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "order1 or anscdf1 or o1 or golden or total or shape" 2>&1 | tail -3 >> $out
for rep in 1 2; do for v in base prev; do
  lib=turbo-range-coder_amd/build/ab/lib$v.so; [ $v = base ] && lib=turbo-range-coder_amd/libturborc_hip.so
  for r in 4 1; do
  echo "== $v TRC_O1_ROWS=$r rep $rep" >> $out
  TRC_O1_ROWS=$r TRC_LIB=$PWD/$lib bash scripts/gpu_kstats.sh zl_$v --codec anscdf1 --no-beyond --no-configs --no-host 2>&1 | grep -E "o1_dec" | cut -c1-150 >> $out
  done
done; done
cat $out
