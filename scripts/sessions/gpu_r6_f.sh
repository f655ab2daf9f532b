#!/bin/bash
# round 6, session f: host layer with pieces (copies) decoupled from slices (kernels) and the out-worker thread
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
out=gpurun_out/r06f_host.txt; : > $out
timeout 1500 python -m pytest tests/test_gpu_host_layer.py -q -m gpu 2>&1 | tail -15 >> $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "host or bounded or raw or thread" 2>&1 | tail -3 >> $out
timeout 900 python -m pytest tests/test_zz_gpu_harness.py -x -q -m gpu -k "not gather and not rccl" 2>&1 | tail -5 >> $out
python - <<'PY' >> $out 2>&1
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "turbo-range-coder_amd")
import trc_testlib as T
T.drift_bytes(100 * 1000 * 1000, 3).tofile("/tmp/drift100m.bin")
T.text_bytes(100 * 1000 * 1000, 7).tofile("/tmp/text100m.bin")
PY
run() {
  echo "== $*" >> $out
  env "$@" timeout 300 ./harness/trcbench -I 7 -e 46,56,1,79 /tmp/drift100m.bin 2>&1 | grep -v "^file\|C Size" >> $out
  env "$@" timeout 300 ./harness/trcbench -I 7 -e 46,56,1 --pin /tmp/drift100m.bin 2>&1 | grep -v "^file\|C Size" | sed 's/$/  [pin]/' >> $out
  env "$@" timeout 300 ./harness/trcbench -I 7 -e 65,42 /tmp/text100m.bin 2>&1 | grep -v "^file\|C Size" >> $out
  env "$@" timeout 300 ./harness/trcbench -I 7 -e 65,42 --pin /tmp/text100m.bin 2>&1 | grep -v "^file\|C Size" | sed 's/$/  [pin]/' >> $out
}
run TRC_HOST_STREAMS=2
run TRC_HOST_STREAMS=2 TRC_HOST_PIECE=4194304
run TRC_HOST_STREAMS=2 TRC_HOST_PIECE=16777216
run TRC_HOST_STREAMS=1
run TRC_HOST_STREAMS=3
run TRC_HOST_STREAMS=2 TRC_CHUNK=512
run TRC_HOST_STREAMS=2 TRC_COPY_THREADS=10
cat $out
