#!/bin/bash
# round 5: anscdf1 decoder -- how much of a wave's chain is table traffic?  Timing ablation -DTRC_O1_ABL_NOMEM (no table ever moves;
# output wrong by construction) against the shipped kernel, 64 / 16 chunks per wave, chunk 4096 / 1024.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
AB=$GRAFT_REPO_ROOT/turbo-range-coder_amd/build/ab
for v in main o1nomem; do for rows in 16 64; do for ch in 4096 1024; do
  if [ $v = main ]; then unset TRC_LIB; else export TRC_LIB=$AB/lib$v.so; fi
  TRC_O1_ROWS=$rows timeout 300 python bench.py --no-cpu --no-verify --steps 5 --warmup 1 --codec anscdf1 --chunk $ch 2>gpurun_out/za_err.log | tail -1 | python -c "
import sys, json
try:
    r = json.loads(sys.stdin.readline()); rf = r['roofline']
    print('$v rows $rows chunk $ch: enc %.3f dec %.3f ms' % (rf['enc_kernel_ms'], rf['dec_kernel_ms']))
except Exception as e:
    print('$v rows $rows chunk $ch FAILED', e); print(open('gpurun_out/za_err.log').read()[-400:])
"
done; done; done
