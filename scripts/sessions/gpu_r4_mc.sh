#!/bin/bash
# round 4: the two-wave (model wave + coder wave) encoders against the one-wave forms -> gpurun_out/r04_mc.log
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
echo "### parity (two-wave form is the default)"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rccdf-  or rccdfi- or rccdf] or rccdfi] or rccdf-bwt or rccdfi" 2>&1 | tail -5
for mc in 0 1; do
  echo "### TRC_RCA_MC=$mc"
  TRC_RCA_MC=$mc bash scripts/gpu_codec_sweep.sh "rccdf rccdfi" "1536 512 768 1024"
done
} > gpurun_out/r04_mc.log 2>&1
cat gpurun_out/r04_mc.log
