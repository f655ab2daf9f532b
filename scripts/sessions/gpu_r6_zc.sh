#!/bin/bash
# round 6, session zc: where the walk kernel's waves spend their time (per-wave wall clocks and rounds, -DTRC_O1W_PROF builds)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
out=gpurun_out/r06zc_o1_walk_prof.txt; : > $out
for v in pw2g32 pw4g96 pw2g24; do
  echo "== $v" >> $out
  TRC_LIB=$PWD/turbo-range-coder_amd/build/ab/lib$v.so timeout 300 python scripts/probe/o1w_prof.py >> $out 2>&1
done
cat $out
