#!/bin/bash
# round 5: which pass of the anscdf encoder is bimodal between processes (0.59 / 0.68 ms)?  kernel stats of six processes
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2 3 4 5 6 7 8; do
  bash scripts/gpu_kstats.sh zh_$i --codec anscdf --no-beyond 2>&1 | grep -E "model2|codeq|ansa_dec" | cut -c1-100 | tr '\n' ' '; echo
done
