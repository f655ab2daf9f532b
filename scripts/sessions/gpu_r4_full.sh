#!/bin/bash
# round 4: the whole -m gpu suite on the current tree + schedule cost + default bench line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
bash scripts/sessions/gpu_r4_sched.sh
python bench.py --no-cpu 2>/dev/null | grep '^{' | tail -1 | cut -c1-1500
} > gpurun_out/r04_full.log 2>&1
tail -40 gpurun_out/r04_full.log
