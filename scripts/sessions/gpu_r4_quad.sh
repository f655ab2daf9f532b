#!/bin/bash
# round 4: workgroups of four (eight) waves -- parity of everything touched + the config 3 / 4 coders at 1536 / 512
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
echo "### parity (device layer, goldens, mixed, bursts, corrupt) for the coders of the four converted files"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "not bench_config and not gigabyte and not host_pointer" 2>&1 | tail -5
for mc in 1 0; do
  echo "### two-wave encoders: $mc"
  TRC_RCA_MC=$mc TRC_RCB_MC=$mc TRC_ANSA_MC=$mc bash scripts/gpu_codec_sweep.sh "rccdf anscdf rcs" "1536 512"
done
bash scripts/gpu_codec_sweep.sh "rccdfi ansb rccdf4 anscdf4 anscdf1" "0"
echo "### hist"
python scripts/probe/hist_time.py 2>&1 | tail -1
} > gpurun_out/r04_quad.log 2>&1
cat gpurun_out/r04_quad.log
