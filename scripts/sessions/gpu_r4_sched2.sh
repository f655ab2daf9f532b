#!/bin/bash
# round 4: where the schedule's remaining cost is (kernel stats of plain vs --force-dist --group 8) + histogram forms
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
bash scripts/gpu_kstats.sh r4_plain --no-beyond --steps 40
bash scripts/gpu_kstats.sh r4_fd8 --no-beyond --force-dist --group 8 --lag 1 --steps 40
echo "### hist forms"
for f in 1 2; do echo "TRC_HIST_FORM=$f"; TRC_HIST_FORM=$f python scripts/probe/hist_time.py 2>&1 | tail -1; done
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cdfini or bench_config_total_parity and anscdf4s" 2>&1 | tail -2
} > gpurun_out/r04_sched2.log 2>&1
cut -c1-200 gpurun_out/r04_sched2.log
