#!/bin/bash
# round 4: order-1 rANS with a per-lane write-back table cache in LDS
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
L=gpurun_out/r04_o1_cache.log
b() { python bench.py --codec $1 --no-cpu --no-beyond $3 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', '$2', 'value', d['value'], 'ms', d['ms_per_step'], 'enc', r['enc_kernel_ms'], 'dec', r['dec_kernel_ms'])"; }
{
echo "### parity (anscdf1)"
timeout 2000 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "anscdf1 or forms" 2>&1 | tail -3
TRC_FUZZ_SEEDS=25 TRC_FUZZ_CODECS="$(python -c 'import sys; sys.path.insert(0,"turbo-range-coder_amd"); import trc; print(trc.ANSO1)')" timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -2
echo "### bench"
for i in 1 2; do b anscdf1; done
b anscdf1 text "--input text"
b anscdf1 c2048 "--chunk 2048"
bash scripts/gpu_kstats.sh r4_o1 --codec anscdf1 --no-beyond
} > $L 2>&1
cat $L
