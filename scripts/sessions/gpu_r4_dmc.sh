#!/bin/bash
# round 4: two-wave decoders (rccdf / rccdfi / anscdf) -- parity + A/B -> gpurun_out/r04_dmc.log
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
echo "### parity"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "(rccdf or anscdf) and not bench_config and not gigabyte and not host_pointer" 2>&1 | tail -5
for mc in 1 0; do
  echo "### two-wave decoders: $mc"
  TRC_RCA_DMC=$mc TRC_ANSA_DMC=$mc bash scripts/gpu_codec_sweep.sh "rccdf rccdfi anscdf" "1536 512"
done
} > gpurun_out/r04_dmc.log 2>&1
cat gpurun_out/r04_dmc.log
