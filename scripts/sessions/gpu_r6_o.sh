#!/bin/bash
# round 6, session o: trace of the striped encode (slices with arrival gates)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
python - <<'PY'
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "turbo-range-coder_amd")
import trc_testlib as T
d = T.drift_bytes(100 * 1000 * 1000, 3); d.tofile("/tmp/drift100m.bin")
PY
rm -rf gpurun_out/trace_o
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_o -o t -- $GRAFT_REPO_ROOT/harness/trcbench -I 3 -e 46 --pin /tmp/drift100m.bin > /dev/null 2>&1)
python scripts/trace_timeline.py gpurun_out/trace_o enc > gpurun_out/r06o_trace_enc_striped.txt 2>&1
tail -75 gpurun_out/r06o_trace_enc_striped.txt
rm -rf gpurun_out/trace_o
timeout 600 python -m pytest tests/test_gpu_host_layer.py -q -m gpu -x -k environment 2>&1 | tail -30
