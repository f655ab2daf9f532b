#!/bin/bash
# first GPU pass: smoke, parity tests, short bench, rocprof kernel stats
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6 > gpurun_out/rocminfo.txt 2>&1
nproc >> gpurun_out/rocminfo.txt; lscpu | grep "Model name" >> gpurun_out/rocminfo.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -3 gpurun_out/bench.log
