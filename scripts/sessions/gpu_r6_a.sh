#!/bin/bash
# round 6, session a: the host-pointer layer with the ratio-aware chunk and concurrent slices.
# (1) correctness: host-layer parity tests + the harness tests; (2) e2e rate and ratio through the reference-named calls,
# 100 MB drift (ids 46, 56, 1) and text (65), pageable and page-locked, for 1 / 2 / 4 / 8 coder streams.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
out=gpurun_out/r06a_host.txt; : > $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "host or bounded or raw" 2>&1 | tail -5 >> $out
timeout 900 python -m pytest tests/test_zz_gpu_harness.py -x -q -m gpu -k "not gather and not rccl" 2>&1 | tail -5 >> $out
python - <<'PY' >> $out 2>&1
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, "turbo-range-coder_amd")
import trc_testlib as T
t = time.time()
T.drift_bytes(100 * 1000 * 1000, 3).tofile("/tmp/drift100m.bin")
T.text_bytes(100 * 1000 * 1000, 7).tofile("/tmp/text100m.bin")
print("inputs generated in %.1f s" % (time.time() - t))
PY
for ns in 1 2 4 8; do
  for pin in "" "--pin"; do
    echo "== TRC_HOST_STREAMS=$ns $pin drift100m" >> $out
    TRC_HOST_STREAMS=$ns timeout 300 ./harness/trcbench -I 5 -e 46,47,56,1,66 $pin /tmp/drift100m.bin 2>&1 | grep -v "^file" >> $out
    echo "== TRC_HOST_STREAMS=$ns $pin text100m" >> $out
    TRC_HOST_STREAMS=$ns timeout 300 ./harness/trcbench -I 5 -e 65,42,45,44 $pin /tmp/text100m.bin 2>&1 | grep -v "^file" >> $out
  done
done
echo "== old policy for comparison: TRC_CHUNK=512, streams 4, --pin" >> $out
TRC_CHUNK=512 timeout 300 ./harness/trcbench -I 5 -e 46,56,1 --pin /tmp/drift100m.bin 2>&1 | grep -v "^file" >> $out
TRC_CHUNK=512 timeout 300 ./harness/trcbench -I 5 -e 65 --pin /tmp/text100m.bin 2>&1 | grep -v "^file" >> $out
echo "== GPU_MAX_HW_QUEUES=8, streams 8, --pin" >> $out
GPU_MAX_HW_QUEUES=8 TRC_HOST_STREAMS=8 timeout 300 ./harness/trcbench -I 5 -e 46,56,1 --pin /tmp/drift100m.bin 2>&1 | grep -v "^file" >> $out
cat $out
