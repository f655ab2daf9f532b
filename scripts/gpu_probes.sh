#!/bin/bash
# gpu_probes.sh TAG -- the two single-question probes of round 5 on this box:
#   vcc_hazard (back-to-back VALU write -> VALU read of VCC / SGPR pairs under co-residency), valu_occ (aggregate issue rates)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
TAG=${1:-r05}
( hostname; date; /opt/rocm/bin/rocm-smi --showuniqueid 2>/dev/null | grep -i "unique" | head -2 ) > gpurun_out/${TAG}_vcc_hazard.txt
timeout 600 scripts/probe/vcc_hazard --trips ${TRIPS:-100000} --reps ${REPS:-1} >> gpurun_out/${TAG}_vcc_hazard.txt 2>&1; echo "exit $?" >> gpurun_out/${TAG}_vcc_hazard.txt
tail -5 gpurun_out/${TAG}_vcc_hazard.txt
if [ -z "$NO_RATES" ]; then
  timeout 600 scripts/probe/valu_occ > gpurun_out/${TAG}_valu_rates.txt 2>&1
  timeout 600 scripts/probe/valu_occ --half >> gpurun_out/${TAG}_valu_rates.txt 2>&1
  cat gpurun_out/${TAG}_valu_rates.txt
fi
