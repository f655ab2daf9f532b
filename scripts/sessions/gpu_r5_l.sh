#!/bin/bash
# round 5: the CDF16 search as bit-selects under masks (no runs of VOP2 v_cndmask) -- parity, A/B against -DTRC_NIB_SEARCH_CND=1
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rccdf or anscdf or total_parity or golden or vnib or vlc or nibble" > gpurun_out/r05m_parity.log 2>&1; tail -2 gpurun_out/r05m_parity.log
for rep in 1 2 3; do for v in nibcnd main; do
  if [ "$v" = "main" ]; then unset TRC_LIB; else export TRC_LIB=$GRAFT_REPO_ROOT/turbo-range-coder_amd/build/ab/lib$v.so; fi
  echo "--- variant $v (rep $rep)"
  bash scripts/gpu_codec_sweep.sh "anscdf rccdf rccdfi" "1536"; bash scripts/gpu_codec_sweep.sh "rccdf4 anscdf4 rccdf8" "512"; bash scripts/gpu_codec_sweep.sh "anscdf1" "4096"
done; done 2>&1 | tee gpurun_out/r05m_ab.txt
