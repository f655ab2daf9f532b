#!/bin/bash
# rocprofv3 kernel-trace stats of the default bench + chunk-size sweep
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r01}
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
for c in 512 1024 2048 4096 8192; do
  timeout 300 python bench.py --steps 10 --warmup 2 --chunk $c --no-cpu 2>/dev/null | tail -1 > gpurun_out/bench_chunk$c.json
  python - <<PY
import json; r=json.load(open("gpurun_out/bench_chunk$c.json")); print("chunk",$c,"value",r["value"],"enc",r["enc_MBps"],"dec",r["dec_MBps"],"ratio",r["config"]["ratio"],"ms",r["ms_per_step"], r["roofline"]["enc_kernel_ms"], r["roofline"]["dec_kernel_ms"])
PY
done
rm -rf gpurun_out/prof_$TAG
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o $TAG -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1)
find gpurun_out/prof_$TAG -name "*stats*" | head; 
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
