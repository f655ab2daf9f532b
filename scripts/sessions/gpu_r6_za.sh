#!/bin/bash
# round 6, session za: anscdf1 walk kernel with the unit list (one request per round, ahead of need): order-1 parity, then kernel times
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r06za_o1_walk_list.txt; : > $out
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "order1 or anscdf1 or o1 or golden or total" 2>&1 | tail -5 >> $out
for v in base $VARIANTS; do
  lib=turbo-range-coder_amd/build/ab/lib$v.so; [ $v = base ] && lib=turbo-range-coder_amd/libturborc_hip.so
  echo "== $v" >> $out
  TRC_LIB=$PWD/$lib bash scripts/gpu_kstats.sh za_$v --codec anscdf1 --no-beyond --no-configs --no-host 2>&1 | grep -E "o1_|ansa_code|value" | cut -c1-150 >> $out
done
cat $out
