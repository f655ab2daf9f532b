#!/bin/bash
# round 5, first kernel session: cndmask run lengths (probe), parity of the hazard-safe kernels and of the new quad transposes,
# A/B against the round-4 library (scripts/build_variant.sh r4base from the round-4 sources)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
scripts/probe/valu_occ --only cnd > gpurun_out/r05_cnd_runs.txt 2>&1; tail -8 gpurun_out/r05_cnd_runs.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "device_layer_matches_oracle or total_parity or golden" > gpurun_out/r05a_parity.log 2>&1; tail -3 gpurun_out/r05a_parity.log
for rep in 1 2; do for v in r4base main; do
  if [ "$v" = "main" ]; then unset TRC_LIB; else export TRC_LIB=$GRAFT_REPO_ROOT/turbo-range-coder_amd/build/ab/lib$v.so; fi
  echo "--- variant $v (rep $rep)"
  bash scripts/gpu_codec_sweep.sh "anscdf4s rccdfs" "512"; bash scripts/gpu_codec_sweep.sh "rccdfs2" "1024"; bash scripts/gpu_codec_sweep.sh "rcs anscdf ansb rccdf" "1536"
done; done 2>&1 | tee gpurun_out/r05a_ab.txt
