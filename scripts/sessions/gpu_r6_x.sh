#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_host_layer.py -q -m gpu 2>&1 | tail -6; timeout 400 python scripts/soak_host_layer.py 150 1; timeout 300 python scripts/soak_host_layer.py 100 7; timeout 300 python scripts/soak_host_layer.py 100 11 ) > gpurun_out/r06x_soak.txt 2>&1; tail -25 gpurun_out/r06x_soak.txt
python - <<'PY'
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "turbo-range-coder_amd")
import trc_testlib as T
T.drift_bytes(100 * 1000 * 1000, 3).tofile("/tmp/drift100m.bin")
PY
./harness/trcbench -I 7 -e 46,56,1 --pin /tmp/drift100m.bin 2>&1 | grep -v "^file"; ./harness/trcbench -I 7 -e 46,56,1 /tmp/drift100m.bin 2>&1 | grep -v "^file"
