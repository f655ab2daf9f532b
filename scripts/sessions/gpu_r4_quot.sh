#!/bin/bash
# round 4: quotient15 without the normalising shifts (static range decoders)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
L=gpurun_out/r04_quotient15.log
b() { python bench.py --codec $1 --no-cpu --no-beyond $3 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', '$2', 'value', d['value'], 'ms', d['ms_per_step'], 'enc', r['enc_kernel_ms'], 'dec', r['dec_kernel_ms'])"; }
{
echo "### parity (static range coders)"
timeout 2000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "rccdfs or random or corrupt or vlc or rccdfu or rccdfv" 2>&1 | tail -3
echo "### bench"
for i in 1 2; do b rccdfs2; b rccdfs; done
} > $L 2>&1
cat $L
