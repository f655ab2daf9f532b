#!/bin/bash
# round 5: headline decoder pair step -- v_add + v_alignbyte instead of v_lshlrev + v_alignbit (one instruction of the 4-cycle class less): A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
AB=$GRAFT_REPO_ROOT/turbo-range-coder_amd/build/ab
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "anscdf4s" 2>&1 | tail -2
for rep in 1 2 3; do for v in main noalignbyte; do
  if [ $v = main ]; then unset TRC_LIB; else export TRC_LIB=$AB/lib$v.so; fi
  echo "== $v (rep $rep)"
  bash scripts/gpu_kstats.sh zi_${v}_$rep --no-beyond --no-configs 2>&1 | grep -E "ans4s_dec|value" | cut -c1-120
done; done
