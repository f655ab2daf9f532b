#!/bin/bash
# round 5 (third session): what bounds the gather kernel -- residency (VGPRs: 84 -> 5 waves per SIMD, 2.4 rounds of workgroups),
# the in-kernel group-base sums, or the fabric?  Variants: vectors per thread per trip 4 / 3 (64 / 52 VGPRs), waves_per_eu(8), scan kernel forced.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
AB=$GRAFT_REPO_ROOT/turbo-range-coder_amd/build/ab
for rep in 1 2; do
  for v in base gvpt4 gvpt3 gocc8; do
    echo "== $v (rep $rep)"
    if [ $v = base ]; then unset TRC_LIB; else export TRC_LIB=$AB/lib$v.so; fi
    bash scripts/gpu_kstats.sh z_${v}_$rep --no-beyond --no-configs 2>&1 | grep -E "gather|ans4s|value" | cut -c1-160
  done
  unset TRC_LIB
  echo "== base, scan kernel forced (rep $rep)"
  TRC_SCAN_MAX=0 bash scripts/gpu_kstats.sh z_scan_$rep --no-beyond --no-configs 2>&1 | grep -E "gather|ans4s|scan|value" | cut -c1-160
done
