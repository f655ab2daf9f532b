#!/bin/bash
# round 5: TrcPace between the hi wave and the lo wave of the anscdf model pass -- parity, A/B, traffic
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "anscdf and not anscdf4s" > gpurun_out/r05i_parity.log 2>&1; tail -2 gpurun_out/r05i_parity.log
for rep in 1 2 3; do for v in m2nopace main; do
  if [ "$v" = "main" ]; then unset TRC_LIB; else export TRC_LIB=$GRAFT_REPO_ROOT/turbo-range-coder_amd/build/ab/lib$v.so; fi
  echo "--- variant $v (rep $rep)"
  bash scripts/gpu_codec_sweep.sh "anscdf" "1536 512"
done; done 2>&1 | tee gpurun_out/r05i_ab.txt
