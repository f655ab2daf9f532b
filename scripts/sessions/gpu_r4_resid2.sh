#!/bin/bash
# round 4: where the waves land (SIMD level) + the two-wave encoders again -> gpurun_out/r04_resid2.log
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
R=scripts/probe/residency
$R 1018 64 36352
$R 1018 128 40448
$R 1018 64 36352 200 40448 128
$R 1018 64 36352 200 36864 128
$R 1018 64 36352 200 36352 64
$R 1018 128 40448 200 36352 64
$R 1018 128 36352 200 36352 128
$R 509 256 80896
echo "### parity"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rcs or anscdf- or anscdf] or rccdf- or rccdf]" 2>&1 | tail -5
for mc in 0 1; do
  echo "### TRC_RCB_MC=$mc TRC_ANSA_MC=$mc"
  TRC_RCB_MC=$mc TRC_ANSA_MC=$mc bash scripts/gpu_codec_sweep.sh "rcs anscdf" "1536 512 1024"
done
echo "### hist"
python scripts/probe/hist_time.py 2>&1 | tail -1
} > gpurun_out/r04_resid2.log 2>&1
cat gpurun_out/r04_resid2.log
