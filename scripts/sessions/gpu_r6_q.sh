#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_host_layer.py -q -m gpu -x -k "striped or gate" --durations=5 2>&1 | tail -30 > gpurun_out/r06q.txt; cat gpurun_out/r06q.txt
