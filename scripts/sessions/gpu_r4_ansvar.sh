#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for i in 1 2 3; do bash scripts/gpu_codec_sweep.sh "anscdf rccdf" "1536"; done
python bench.py --codec anscdf --no-cpu 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('20 steps, steady clocks:', r['value'], r['roofline']['enc_kernel_ms'], r['roofline']['dec_kernel_ms'])"
TRC_ANSA_MC=0 bash scripts/gpu_codec_sweep.sh "anscdf" "1536"
TRC_ANSA_CODEQ=0 bash scripts/gpu_codec_sweep.sh "anscdf" "1536"
} > gpurun_out/r04_ansvar.log 2>&1
cat gpurun_out/r04_ansvar.log
