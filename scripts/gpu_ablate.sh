#!/bin/bash
# timing ablations (results are wrong by construction: --no-verify): gpu_ablate.sh "variants" "bench args"
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for v in $1; do
  if [ "$v" = "main" ]; then unset TRC_LIB; else export TRC_LIB=$GRAFT_REPO_ROOT/turbo-range-coder_amd/build/ab/lib$v.so; fi
  python bench.py --no-cpu --no-verify --no-cold --steps 10 --warmup 2 $2 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); rf=r['roofline']
print('%-10s enc %.4f ms dec %.4f ms step %.4f ms' % ('$v', rf['enc_kernel_ms'], rf['dec_kernel_ms'], r['ms_per_step']))"
done; done
