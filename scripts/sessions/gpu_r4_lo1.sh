#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
bash scripts/gpu_ab.sh "main lo1" "rccdf anscdf" "1536" 2 > gpurun_out/r04_lo1.log 2>&1
grep -v "^$" gpurun_out/r04_lo1.log | cut -c1-170
