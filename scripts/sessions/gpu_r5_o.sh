#!/bin/bash
# round 5: write-through drains (StreamOut<.., WT>) -- parity of every coder that uses them, kernel stats before / after
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "anscdf4s or rccdfs or anscdf or ansb or static_rans or total_parity or mixed_raw or alias or shapes or golden" > gpurun_out/r05p_parity.log 2>&1; tail -2 gpurun_out/r05p_parity.log
for c in "" "--codec rccdfs" "--codec anscdf" "--codec ansb"; do echo "--- kstats main $c"; bash scripts/gpu_kstats.sh r5p_x $c --no-beyond 2>&1 | head -5 | tail -4; done
python bench.py --no-cpu --no-beyond 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('bench', r['value'], r['ms_per_step'], r['value_cold_clocks'], r['roofline']['enc_kernel_ms'], r['roofline']['dec_kernel_ms'])"
