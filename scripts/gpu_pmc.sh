#!/bin/bash
# PMC passes (separate from kernel-trace/stats, per the guide) on one bench configuration.
# usage: gpu_pmc.sh TAG "bench args" "CTRSET1" "CTRSET2" ...
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=$1; shift; BARGS=$1; shift
if [ ! -f gpurun_out/counters_list.txt ]; then (cd /tmp && rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/counters_list.txt 2>&1); fi
i=0
for CT in "$@"; do
  i=$((i+1))
  rm -rf gpurun_out/pmc_${TAG}_$i
  (cd /tmp && timeout 600 rocprofv3 --pmc $CT --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_$i -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-beyond $BARGS > $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_$i.log 2>&1)
  f=$(find gpurun_out/pmc_${TAG}_$i -name "*counter_collection.csv" | head -1)
  echo "== pass $i: $CT -> $f"
  [ -n "$f" ] && python scripts/pmc_summary.py "$f"
done
