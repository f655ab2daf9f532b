#!/bin/bash
# round 4: ansb coding pass with four lanes per chunk
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
L=gpurun_out/r04_ansb_codeq.log
b() { python bench.py --codec $1 --no-cpu --no-beyond $3 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', '$2', 'value', d['value'], 'ms', d['ms_per_step'], 'enc', r['enc_kernel_ms'], 'dec', r['dec_kernel_ms'])"; }
{
echo "### parity (ansb)"
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ansb" 2>&1 | tail -3
TRC_FUZZ_SEEDS=25 TRC_FUZZ_CODECS="$(python -c 'import sys; sys.path.insert(0,"turbo-range-coder_amd"); import trc; print(trc.ANSB)')" timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -2
echo "### bench"
for i in 1 2; do b ansb; done
b ansb c512 "--chunk 512"
} > $L 2>&1
cat $L
bash scripts/gpu_kstats.sh r4_ansb --codec ansb --no-beyond >> gpurun_out/r04_ansb_codeq.log 2>&1; tail -8 gpurun_out/r04_ansb_codeq.log
