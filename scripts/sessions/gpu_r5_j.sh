#!/bin/bash
# round 5: anscdf model pass -- both halves of an input line requested together (QuadIn::take_fwd): time A/B, parity, fabric traffic
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "anscdf and not anscdf4s and not anscdf1" > gpurun_out/r05k_parity.log 2>&1; tail -1 gpurun_out/r05k_parity.log
for rep in 1 2 3; do for v in m2nopair main; do
  if [ "$v" = "main" ]; then unset TRC_LIB; else export TRC_LIB=$GRAFT_REPO_ROOT/turbo-range-coder_amd/build/ab/lib$v.so; fi
  echo "--- variant $v (rep $rep)"
  bash scripts/gpu_codec_sweep.sh "anscdf" "1536 1088"
done; done 2>&1 | tee gpurun_out/r05k_ab.txt
unset TRC_LIB
bash scripts/gpu_kstats.sh r5k_anscdf --codec anscdf --no-beyond 2>&1 | head -8
{ echo "#### anscdf"; bash scripts/gpu_pmc.sh tr5_anscdf "--codec anscdf" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum"; } > gpurun_out/r05_pmc_traffic_anscdf.txt 2>&1
grep -A5 "model2\|codeq" gpurun_out/r05_pmc_traffic_anscdf.txt | cut -c1-100 | head -40
