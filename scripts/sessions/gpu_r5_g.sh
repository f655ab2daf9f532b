#!/bin/bash
# round 5: TrcPace in the four-lanes-per-chunk rANS coding passes (anscdf / ansb / anscdf1) -- parity, A/B (TRC_CODEQ_GPW=1: small workgroups)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "anscdf or ansb or total_parity" > gpurun_out/r05h_parity.log 2>&1; tail -2 gpurun_out/r05h_parity.log
for rep in 1 2; do for g in 1 4; do
  export TRC_CODEQ_GPW=$g
  echo "--- TRC_CODEQ_GPW=$g (rep $rep)"
  bash scripts/gpu_codec_sweep.sh "anscdf ansb" "1536"; bash scripts/gpu_codec_sweep.sh "anscdf1" "4096"
done; done 2>&1 | tee gpurun_out/r05h_ab.txt
unset TRC_CODEQ_GPW
bash scripts/gpu_kstats.sh r5h_anscdf --codec anscdf --no-beyond 2>&1 | head -12
