#!/bin/bash
# round 4: rcs decoder with the bit kept on the vector side (carry chain -> VGPR mask -> bit-selects)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
L=gpurun_out/r04_rcs_dec_vector_mask.log
{
echo "### parity (rcs)"
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "rcs or rcb or bit" 2>&1 | tail -3
echo "### bench"
for i in 1 2; do python bench.py --codec rcs --no-cpu --no-beyond 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('rcs value', d['value'], 'ms', d['ms_per_step'], 'enc', r['enc_kernel_ms'], 'dec', r['dec_kernel_ms'])"; done
} > $L 2>&1
cat $L
