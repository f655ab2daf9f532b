#!/bin/bash
# round 6, session zj: order-1 decoder (four lanes per chunk): the pair (t[x], t[x+1]) picked by DPP instead of ds_bpermute
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r06zj_o1_dpp_pick.txt; : > $out
L=$PWD/turbo-range-coder_amd/build/ab/libdpp1.so
TRC_LIB=$L TRC_O1_ROWS=4 timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "order1 or anscdf1 or o1 or golden or total" 2>&1 | tail -3 >> $out
for v in dpp1 base; do lib=$L; [ $v = base ] && lib=$PWD/turbo-range-coder_amd/libturborc_hip.so
  echo "== $v, TRC_O1_ROWS=4" >> $out; TRC_LIB=$lib TRC_O1_ROWS=4 timeout 250 python scripts/probe/o1_dec_sizes.py 2>&1 | grep -v amdgpu.ids >> $out; done
for rep in 1 2; do for v in dpp1 base; do lib=$L; [ $v = base ] && lib=$PWD/turbo-range-coder_amd/libturborc_hip.so
  echo "== bench loop, $v rep $rep" >> $out
  TRC_LIB=$lib bash scripts/gpu_kstats.sh zj_$v --codec anscdf1 --no-beyond --no-configs --no-host 2>&1 | grep -E "o1_dec|value" | cut -c1-150 >> $out
done; done
cat $out
