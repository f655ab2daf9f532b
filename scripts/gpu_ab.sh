#!/bin/bash
# A/B of library variants (scripts/build_variant.sh): gpu_ab.sh "variants" "codecs" "chunks" [reps]
cd "$GRAFT_REPO_ROOT"
for rep in $(seq 1 ${4:-2}); do for v in $1; do
  if [ "$v" = "main" ]; then unset TRC_LIB; else export TRC_LIB=$GRAFT_REPO_ROOT/turbo-range-coder_amd/build/ab/lib$v.so; fi
  echo "--- variant $v (rep $rep)"; bash scripts/gpu_codec_sweep.sh "$2" "$3"
done; done
