#!/bin/bash
# round 6, session b: host layer after the slice-plan fix (group-aligned slices, equal slices without ramps for kernel-bound coders)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
out=gpurun_out/r06b_host.txt; : > $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "host or bounded or raw" 2>&1 | tail -3 >> $out
timeout 900 python -m pytest tests/test_zz_gpu_harness.py -x -q -m gpu -k "not gather and not rccl" 2>&1 | tail -5 >> $out
python - <<'PY' >> $out 2>&1
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "turbo-range-coder_amd")
import trc_testlib as T
T.drift_bytes(100 * 1000 * 1000, 3).tofile("/tmp/drift100m.bin")
T.text_bytes(100 * 1000 * 1000, 7).tofile("/tmp/text100m.bin")
PY
run() { # label, env...
  echo "== $*" >> $out
  env "$@" timeout 300 ./harness/trcbench -I 7 -e 46,56,1 /tmp/drift100m.bin 2>&1 | grep -v "^file\|C Size" >> $out
  env "$@" timeout 300 ./harness/trcbench -I 7 -e 46,56,1 --pin /tmp/drift100m.bin 2>&1 | grep -v "^file\|C Size" | sed 's/$/  [pin]/' >> $out
  env "$@" timeout 300 ./harness/trcbench -I 7 -e 65,42 /tmp/text100m.bin 2>&1 | grep -v "^file\|C Size" >> $out
  env "$@" timeout 300 ./harness/trcbench -I 7 -e 65,42 --pin /tmp/text100m.bin 2>&1 | grep -v "^file\|C Size" | sed 's/$/  [pin]/' >> $out
}
run TRC_HOST_STREAMS=1
run TRC_HOST_STREAMS=2
run TRC_HOST_STREAMS=4
run TRC_HOST_STREAMS=8
run TRC_HOST_STREAMS=4 GPU_MAX_HW_QUEUES=8
run TRC_HOST_STREAMS=8 GPU_MAX_HW_QUEUES=8
run TRC_HOST_STREAMS=8 GPU_MAX_HW_QUEUES=12
run TRC_HOST_STREAMS=4 TRC_CHUNK=512
cat $out
