#!/bin/bash
# round 4: SQ counters of the config 3 / 4 kernels at their bench chunk (same counter sets as profiles/r03_pmc_sq_cfg34.txt)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for c in rccdf rcs anscdf; do
  echo "#### $c"
  bash scripts/gpu_pmc.sh sq_$c "--codec $c" \
    "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
    "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA"
done
} > gpurun_out/r04_pmc_sq_cfg34.txt 2>&1
grep -A9 "enc_mc_kernel\|model2\|codeq\|dec_kernel" gpurun_out/r04_pmc_sq_cfg34.txt | cut -c1-100 | head -150
