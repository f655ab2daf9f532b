#!/bin/bash
# round 6, session k: 1 GB and small calls through the host-pointer layer; a kernel + copy trace of one 100 MB rccdfenc call
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r06k_host_sizes.txt; : > $out
python - <<'PY' >> $out 2>&1
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "turbo-range-coder_amd")
import trc_testlib as T
d = T.drift_bytes(1000 * 1000 * 1000, 3); d.tofile("/tmp/drift1g.bin")
for n in (1, 10, 100):
    d[:n * 1000 * 1000].tofile("/tmp/drift%dm.bin" % n)
t = T.text_bytes(1000 * 1000 * 1000, 7); t.tofile("/tmp/text1g.bin")
for n in (1, 10, 100):
    t[:n * 1000 * 1000].tofile("/tmp/text%dm.bin" % n)
PY
for sz in 1m 10m 100m 1g; do
  for pin in "" "--pin"; do
    echo "== drift$sz $pin" >> $out
    timeout 600 ./harness/trcbench -I 5 -e 46,56,1 $pin /tmp/drift$sz.bin 2>&1 | grep -v "^file\|C Size" >> $out
    echo "== text$sz $pin" >> $out
    timeout 600 ./harness/trcbench -I 5 -e 65,45 $pin /tmp/text$sz.bin 2>&1 | grep -v "^file\|C Size" >> $out
  done
done
echo "== the old policy on the small and the large call: TRC_CHUNK=512 --pin" >> $out
for sz in 1m 10m 1g; do
  echo "-- drift$sz" >> $out
  TRC_CHUNK=512 timeout 600 ./harness/trcbench -I 5 -e 46 --pin /tmp/drift$sz.bin 2>&1 | grep -v "^file\|C Size" >> $out
done
cat $out
rm -rf gpurun_out/trace_k
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_k -o t -- $GRAFT_REPO_ROOT/harness/trcbench -I 3 -e 46 --pin /tmp/drift100m.bin > /dev/null 2>&1)
python scripts/trace_timeline.py gpurun_out/trace_k enc > gpurun_out/r06k_trace_enc_pinned.txt 2>&1
python scripts/trace_timeline.py gpurun_out/trace_k dec > gpurun_out/r06k_trace_dec_pinned.txt 2>&1
rm -rf gpurun_out/trace_k2
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_k2 -o t -- $GRAFT_REPO_ROOT/harness/trcbench -I 3 -e 46 /tmp/drift100m.bin > /dev/null 2>&1)
python scripts/trace_timeline.py gpurun_out/trace_k2 enc > gpurun_out/r06k_trace_enc_pageable.txt 2>&1
head -40 gpurun_out/r06k_trace_enc_pinned.txt; head -50 gpurun_out/r06k_trace_enc_pageable.txt
rm -rf gpurun_out/trace_k gpurun_out/trace_k2
