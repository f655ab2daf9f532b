#!/bin/bash
# round 5: refill segments aligned to the payload's 64-byte sectors (StreamInT::align_start) -- parity, A/B against -DTRC_DEC_NOALIGN, traffic
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "anscdf4s or rccdfs or static_rans or total_parity or corrupt or mixed_raw or alias or host_pointer" > gpurun_out/r05l_parity.log 2>&1; tail -2 gpurun_out/r05l_parity.log
for rep in 1 2 3; do for v in noalign main; do
  if [ "$v" = "main" ]; then unset TRC_LIB; else export TRC_LIB=$GRAFT_REPO_ROOT/turbo-range-coder_amd/build/ab/lib$v.so; fi
  echo "--- variant $v (rep $rep)"
  bash scripts/gpu_codec_sweep.sh "anscdf4s rccdfs" "512"; bash scripts/gpu_codec_sweep.sh "rccdfs2" "1024"; bash scripts/gpu_codec_sweep.sh "anscdf4s" "4096"
done; done 2>&1 | tee gpurun_out/r05l_ab.txt
unset TRC_LIB
bash scripts/gpu_pmc.sh r05al "" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" > gpurun_out/r05l_pmc_traffic.txt 2>&1
grep -A5 "ans4s_dec" gpurun_out/r05l_pmc_traffic.txt | head -8
python bench.py --no-cpu --no-beyond --force-dist --group 8 2> gpurun_out/r05_forcedist.err | tail -1 > gpurun_out/r05_bench_forcedist_g8.json; tail -5 gpurun_out/r05_forcedist.err; cut -c1-200 gpurun_out/r05_bench_forcedist_g8.json
