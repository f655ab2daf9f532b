#!/bin/bash
# round 4: HBM-side traffic counters of the config 3 / 4 kernels (same counter sets as the headline's pmc_traffic pass)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for c in anscdf rccdf rcs; do
  echo "#### $c"
  bash scripts/gpu_pmc.sh tr_$c "--codec $c" "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum"
done
} > gpurun_out/r04_pmc_traffic_cfg34.txt 2>&1
grep -A4 "model2\|codeq\|dec_kernel\|enc_mc" gpurun_out/r04_pmc_traffic_cfg34.txt | grep -v "^--" | cut -c1-100 | head -120
