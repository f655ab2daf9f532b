#!/bin/bash
# tables in registers (decoders of anscdf / anscdf4 / Turbo-VLC; encoders of Turbo-VLC / vnibble): parity, then timing
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
out=gpurun_out/r03_model2.log; : > $out
python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3 >> $out
TRC_FUZZ_CODECS=5,10,14,15,16,17,18,19,20,21,22,23,24,25,26,27 TRC_FUZZ_SEEDS=200 python -m pytest tests/test_gpu_fuzz.py -x -q -n 4 2>&1 | tail -3 >> $out
bash scripts/gpu_codec_sweep.sh "anscdf anscdf4 rccdf8 rccdfi8 rccdfu16 rccdfu32 rccdfv16 rccdfv32 rccdfvz16 rccdfvz32 anscdfu16 anscdfuz16 anscdfv16 anscdfvz16 anscdfv32 anscdfvz32" "0" >> $out 2>&1
cat $out
