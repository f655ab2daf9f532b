#!/bin/bash
# round 4: where the time of the rcs two-wave encoder is (ablation builds: results wrong by construction, timing only)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for v in main rcb_NOEMIT rcb_NOMODEL rcb_NOCODER; do
  if [ "$v" = "main" ]; then unset TRC_LIB; else export TRC_LIB=$GRAFT_REPO_ROOT/turbo-range-coder_amd/build/ab/lib$v.so; fi
  python bench.py --no-cpu --no-verify --steps 5 --warmup 1 --codec rcs 2>gpurun_out/abl.err | grep '^{' | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('%-12s enc %.3f ms dec %.3f ms' % ('$v', r['roofline']['enc_kernel_ms'], r['roofline']['dec_kernel_ms']))" || tail -3 gpurun_out/abl.err
done
} > gpurun_out/r04_rcb_abl.log 2>&1
cat gpurun_out/r04_rcb_abl.log
