#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "host or bounded or raw or thread" --durations=8 -s 2>&1 | grep -v "^$" | tail -40 > gpurun_out/r06p.txt; cat gpurun_out/r06p.txt
