#!/bin/bash
# quick per-codec throughput table: bench.py (no CPU leg) for the given codecs x chunks -> one line each
# usage: gpu_codec_sweep.sh "codec ..." "chunk ..."
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for cdc in $1; do for ch in $2; do
  timeout 300 python bench.py --no-cpu --steps 5 --warmup 1 --codec $cdc --chunk $ch 2>gpurun_out/sweep_err.log | tail -1 > gpurun_out/sweep_tmp.json
  python -c "
import json
try:
    r=json.load(open('gpurun_out/sweep_tmp.json')); rf=r['roofline']
    print('%-10s chunk %5d  encdec %8.0f MB/s  enc %8.0f dec %8.0f  kernels %.3f/%.3f ms  ratio %.4f' % ('$cdc', r['config']['chunk'], r['value'], r['enc_MBps'], r['dec_MBps'], rf['enc_kernel_ms'], rf['dec_kernel_ms'], r['config']['ratio']))
except Exception as e:
    print('$cdc', $ch, 'FAILED', e); print(open('gpurun_out/sweep_err.log').read()[-600:])
"
done; done
