#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( timeout 400 python scripts/soak_host_layer.py 150 1; timeout 300 python scripts/soak_host_layer.py 100 7 ) > gpurun_out/r06w_soak.txt 2>&1; tail -25 gpurun_out/r06w_soak.txt
