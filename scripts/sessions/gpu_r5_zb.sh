#!/bin/bash
# round 5: anscdf1 decoder with eight lanes per chunk (trc_o1_dec_rows_kernel, TRC_O1_ROWS=1): parity, then time against 16 / 64 chunks per wave
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
TRC_O1_ROWS=1 timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "anscdf1 or order1 or round4_kernel_forms or round5_workgroup_shapes or any_legal_alignment" 2>&1 | tail -8
for rows in 1; do for ch in 4096 2048 1024; do
  TRC_O1_ROWS=$rows timeout 300 python bench.py --no-cpu --steps 5 --warmup 1 --codec anscdf1 --chunk $ch 2>gpurun_out/zb_err.log | tail -1 | python -c "
import sys, json
try:
    r = json.loads(sys.stdin.readline()); rf = r['roofline']
    print('rows $rows chunk $ch: enc %.3f dec %.3f ms  sha %s' % (rf['enc_kernel_ms'], rf['dec_kernel_ms'], r.get('payload_matches_reference_sha256')))
except Exception as e:
    print('rows $rows chunk $ch FAILED', e); print(open('gpurun_out/zb_err.log').read()[-600:])
"
done; done
