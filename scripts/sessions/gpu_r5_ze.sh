#!/bin/bash
# round 5: anscdf1 rows decoder -- table slots swizzled by the chunk number (L2 channel spreading) against unswizzled
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
AB=$GRAFT_REPO_ROOT/turbo-range-coder_amd/build/ab
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "anscdf1 or order1 or round4_kernel_forms or round5_workgroup_shapes or any_legal_alignment" 2>&1 | tail -3
for rep in 1 2; do for v in main o1noswz; do for ch in 4096 2048 1024; do
  if [ $v = main ]; then unset TRC_LIB; else export TRC_LIB=$AB/lib$v.so; fi
  timeout 300 python bench.py --no-cpu --steps 5 --warmup 1 --codec anscdf1 --chunk $ch 2>gpurun_out/ze_err.log | tail -1 | python -c "
import sys, json
try:
    r = json.loads(sys.stdin.readline()); rf = r['roofline']
    print('$v chunk $ch: enc %.3f dec %.3f ms  sha %s' % (rf['enc_kernel_ms'], rf['dec_kernel_ms'], r.get('payload_matches_reference_sha256')))
except Exception as e:
    print('$v chunk $ch FAILED', e); print(open('gpurun_out/ze_err.log').read()[-600:])
"
done; done; done
