#!/bin/bash
# round 6, session zd: walk kernel as four waves per workgroup, one workgroup per CU: order-1 parity, kernel times, per-wave clocks
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r06zd_o1_walk.txt; : > $out
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "order1 or anscdf1 or o1 or golden or total" 2>&1 | tail -5 >> $out
bash scripts/gpu_kstats.sh zd --codec anscdf1 --no-beyond --no-configs --no-host 2>&1 | grep -E "o1_|ansa_code|value" | cut -c1-150 >> $out
TRC_LIB=$PWD/turbo-range-coder_amd/build/ab/libpw4.so timeout 300 python scripts/probe/o1w_prof.py >> $out 2>&1
cat $out
