#!/bin/bash
# round 4: RcEncV::sym<false> by hand (rccdf / rccdfi coder wave)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
L=gpurun_out/r04_rcencv_asm.log
b() { python bench.py --codec $1 --no-cpu --no-beyond 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', '$2', 'value', d['value'], 'ms', d['ms_per_step'], 'enc', r['enc_kernel_ms'], 'dec', r['dec_kernel_ms'])"; }
{
echo "### parity (rccdf)"
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rccdf or forms" 2>&1 | tail -3
TRC_FUZZ_SEEDS=20 TRC_FUZZ_CODECS="$(python -c 'import sys; sys.path.insert(0,"turbo-range-coder_amd"); import trc; print("%d,%d"%(trc.RCA,trc.RCAI))')" timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -2
echo "### bench"
for i in 1 2; do b rccdf; b rccdfi; done
} > $L 2>&1
cat $L
