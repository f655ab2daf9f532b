#!/bin/bash
# round 4: rccdf decoder trims (stream side once per two bytes, index carry chain) + the forms test
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "(rccdf or anscdf or forms or vlc or rccdf8) and not gigabyte and not host_pointer" 2>&1 | tail -4
bash scripts/gpu_codec_sweep.sh "rccdf rccdfi anscdf rccdf4 anscdf4 rccdf8 rccdfv16 anscdfv16" "0"
} > gpurun_out/r04_dec.log 2>&1
cut -c1-200 gpurun_out/r04_dec.log
