#!/bin/bash
# round 5: TrcPace in the static rANS encoder (12-wave workgroups) and decoder -- parity, A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "anscdf4s or static_rans or total_parity or mixed_raw" > gpurun_out/r05f_parity.log 2>&1; tail -2 gpurun_out/r05f_parity.log
for rep in 1 2 3; do for v in nobal main; do
  if [ "$v" = "main" ]; then unset TRC_LIB; else export TRC_LIB=$GRAFT_REPO_ROOT/turbo-range-coder_amd/build/ab/lib$v.so; fi
  echo "--- variant $v (rep $rep)"
  bash scripts/gpu_codec_sweep.sh "anscdf4s" "512 1024 2048"
done; done 2>&1 | tee gpurun_out/r05f_ab.txt
unset TRC_LIB
python bench.py --no-cpu 2>/dev/null | tail -1 > gpurun_out/r05f_bench.json; python -c "
import json; r=json.load(open('gpurun_out/r05f_bench.json')); print(r['value'], r['ms_per_step'], r['roofline'], r.get('value_cold_clocks'), r.get('value_cold'))"
python bench.py --no-cpu --workload zipf1g 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('zipf1g', r['value'], r['ms_per_step'], r['roofline']['enc_kernel_ms'], r['roofline']['dec_kernel_ms'])"
