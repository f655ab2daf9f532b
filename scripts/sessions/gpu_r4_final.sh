#!/bin/bash
# round 4, final tree: the whole -m gpu suite, then the bench lines / kernel stats / counters that changed since gpu_r4_profile.sh
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputest_final.log 2>&1; tail -4 gpurun_out/r04_gputest_final.log
for c in rccdf anscdf rcs; do python bench.py --codec $c 2>/dev/null | grep '^{' | tail -1 > gpurun_out/r04_bench_$c.json; done
python bench.py 2>/dev/null | grep '^{' | tail -1 > gpurun_out/r04_bench.json
bash scripts/gpu_all_codecs.sh > /dev/null 2>&1; cp gpurun_out/all_codecs.txt gpurun_out/r04_all_codecs.txt
bash scripts/sessions/gpu_r4_pmc34.sh > /dev/null 2>&1
for c in rccdf anscdf rcs anscdf1 ansb rccdfs2; do bash scripts/gpu_kstats.sh r4f_$c --codec $c --no-beyond; done > gpurun_out/r04_kernel_stats_cfg34.txt 2>&1
python scripts/probe/hist_time.py 2>&1 | tail -1
grep "rcs \|rccdf \|anscdf \|anscdf1 \|anscdf4s " gpurun_out/r04_all_codecs.txt | head -12
cat gpurun_out/r04_kernel_stats_cfg34.txt | cut -c1-120
