#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "(rccdf- or rccdf] or rccdfi or forms or bench_config) and not gigabyte and not host_pointer" 2>&1 | tail -4
bash scripts/gpu_codec_sweep.sh "rccdf rccdfi" "1536 512 4096"
} > gpurun_out/r04_rca3.log 2>&1
cut -c1-200 gpurun_out/r04_rca3.log
