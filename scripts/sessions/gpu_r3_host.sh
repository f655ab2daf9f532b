#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests/test_zz_gpu_harness.py -m gpu -x -q -k "c_harness or slice_plan or file_tool" 2>&1 | tail -4
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "host_pointer" 2>&1 | tail -3
echo "== pageable (chunk auto = 512)"; timeout 300 ./harness/trcbench -I 5 -e 42,45,65,79 --text 100000000
echo "== page-locked"; timeout 300 ./harness/trcbench -I 5 --pin -e 1,42,45,46,56,65,79 --text 100000000
echo "== page-locked 1 GB"; timeout 300 ./harness/trcbench -I 3 --pin -e 65,79 --text 1000000000
