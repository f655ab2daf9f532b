#!/bin/bash
# round 4: the rANS step with a quotient-only correction (all run-time-divisor coding passes), ansb coding pass with four lanes
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
L=gpurun_out/r04_rans_step.log
b() { python bench.py --codec $1 --no-cpu --no-beyond $3 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', '$2', 'value', d['value'], 'ms', d['ms_per_step'], 'enc', r['enc_kernel_ms'], 'dec', r['dec_kernel_ms'])"; }
{
echo "### parity (rANS coders with a run-time divisor)"
timeout 2000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "ans or random or forms" 2>&1 | tail -3
echo "### bench"
for i in 1 2; do b ansb; b anscdf; b anscdf1; done
bash scripts/gpu_kstats.sh r4_ansb --codec ansb --no-beyond
bash scripts/gpu_kstats.sh r4_anscdf --codec anscdf --no-beyond
} > $L 2>&1
cat $L
