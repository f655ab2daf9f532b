#!/bin/bash
# round 4: payload gather at 64 VGPRs (eight waves per SIMD)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
L=gpurun_out/r04_gather_vgprs.log
{
echo "### parity (every coder goes through the gather)"
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -3
echo "### bench"
for i in 1 2 3; do python bench.py --no-cpu --no-beyond 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('headline value', d['value'], 'ms', d['ms_per_step'], 'enc', r['enc_kernel_ms'], 'dec', r['dec_kernel_ms'], 'cold_clocks', d.get('value_cold_clocks'))"; done
bash scripts/gpu_kstats.sh r4_gather --no-beyond
bash scripts/gpu_kstats.sh r4_gather2 --codec rccdfs2 --no-beyond
} > $L 2>&1
cat $L
