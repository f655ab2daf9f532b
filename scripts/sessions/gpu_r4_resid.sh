#!/bin/bash
# round 4: residency probe (scripts/probe/residency.hip) + rcs two-wave encoder A/B -> gpurun_out/r04_resid.log
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
R=scripts/probe/residency
echo "### residency: 1018 workgroups"
$R 1018 64 36352
$R 1018 64 32768
$R 1280 64 32768
$R 1018 128 40448
$R 1018 128 38400
$R 1018 128 36864
$R 1018 128 40960
$R 1018 64 36352 200 40448
$R 1018 64 36352 200 36352
$R 1018 64 36352 200 2048
$R 977 128 40448
$R 977 64 36352 200 40448
$R 3052 64 36352 100
$R 3052 128 40448 100
echo "### rcs parity (two-wave form is the default)"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rcs" 2>&1 | tail -5
for mc in 0 1; do
  echo "### TRC_RCB_MC=$mc"
  TRC_RCB_MC=$mc bash scripts/gpu_codec_sweep.sh "rcs" "1536 1600 512 768 1024"
done
echo "### rccdf at 1600 / 1536, two-wave vs one-wave"
for mc in 0 1; do TRC_RCA_MC=$mc bash scripts/gpu_codec_sweep.sh "rccdf" "1600 1536 1664"; done
echo "### hist"
python scripts/probe/hist_time.py 2>&1 | tail -5
} > gpurun_out/r04_resid.log 2>&1
cat gpurun_out/r04_resid.log
