#!/bin/bash
# CDF16 model: conflict-free LDS layout + encoder-side walk with the LDS round trips off the dependency chain: parity, then timing
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
out=gpurun_out/r03_model.log; : > $out
python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3 >> $out
TRC_FUZZ_CODECS=${FUZZ_CODECS:-4,5,7,8,9,10,14,15,16,17,18,19,20,21,22,23,24,25,26,27} TRC_FUZZ_SEEDS=100 python -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3 >> $out
bash scripts/gpu_codec_sweep.sh "rccdf rccdfi anscdf rccdf4 rccdf4i anscdf4 rccdf8 rccdfi8 rccdfu16 rccdfv16 rccdfvz32 anscdfv16 anscdfvz32" "0" >> $out 2>&1
bash scripts/gpu_codec_sweep.sh "rccdf anscdf" "512" >> $out 2>&1
cat $out
