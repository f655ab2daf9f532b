#!/bin/bash
# Round profile: default bench JSON, rocprofv3 kernel-trace stats of the same command, and HBM-traffic PMC
# passes (each in its own run, no trace domains mixed in).  Summaries land in gpurun_out/ -> copy to profiles/.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r01}
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 600 gpurun_out/${TAG}_bench.json
rm -rf gpurun_out/prof_$TAG
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o $TAG -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-beyond > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1)
cp gpurun_out/prof_$TAG/${TAG}_kernel_stats.csv gpurun_out/${TAG}_kernel_stats.csv 2>/dev/null; head -8 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-200
bash scripts/gpu_pmc.sh ${TAG}t "" "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum" > gpurun_out/${TAG}_pmc_traffic.txt 2>&1
grep -A6 "ans4s_\(enc\|dec\)_kernel\|gather" gpurun_out/${TAG}_pmc_traffic.txt | grep -v "^--" | head -60
