#!/bin/bash
# round 4 evidence: default bench JSON + rocprofv3 kernel stats + HBM-traffic PMC passes (scripts/gpu_profile_round.sh), the
# zipf1g traffic pass (is the decoder's re-read traffic cache hits beyond the Infinity Cache too?), every coder at its chunk
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
bash scripts/gpu_profile_round.sh r04 > gpurun_out/r04_profile_round.log 2>&1
{
echo "### traffic PMC on --workload zipf1g (1 GB on the device: beyond the 256 MiB Infinity Cache)"
bash scripts/gpu_pmc.sh r04z "--workload zipf1g --no-verify" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum"
} > gpurun_out/r04_pmc_traffic_zipf1g.txt 2>&1
bash scripts/gpu_all_codecs.sh > /dev/null 2>&1; cp gpurun_out/all_codecs.txt gpurun_out/r04_all_codecs.txt
for c in rccdf anscdf rcs; do python bench.py --codec $c 2>/dev/null | grep '^{' | tail -1 > gpurun_out/r04_bench_$c.json; done
python bench.py --workload zipf1g --no-cpu 2>/dev/null | grep '^{' | tail -1 > gpurun_out/r04_bench_zipf1g.json
python bench.py --steps 20 --warmup 5 --clock-warmup-ms 0 --no-cpu --no-beyond 2>/dev/null | grep '^{' | tail -1 > gpurun_out/r04_bench_contract_shape.json
tail -c 400 gpurun_out/r04_bench.json; echo; cat gpurun_out/r04_all_codecs.txt | head -40
