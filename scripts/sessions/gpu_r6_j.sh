#!/bin/bash
# round 6, session j: the whole -m gpu suite on the current tree
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 3300 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/r06j_gputest.log 2>&1; tail -40 gpurun_out/r06j_gputest.log
