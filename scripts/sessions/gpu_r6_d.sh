#!/bin/bash
# round 6, session d: the new host-layer tests
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
out=gpurun_out/r06d_tests.txt; : > $out
timeout 1500 python -m pytest tests/test_gpu_host_layer.py -q -m gpu 2>&1 | tail -40 >> $out
cat $out
