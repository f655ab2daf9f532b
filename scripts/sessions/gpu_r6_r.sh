#!/bin/bash
# round 6, session r: streamed decode (progress flags) -- correctness, then rate
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
out=gpurun_out/r06r_streamed.txt; : > $out
timeout 1200 python -m pytest tests/test_gpu_host_layer.py -q -m gpu -x 2>&1 | tail -15 >> $out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "host or bounded or raw or thread" 2>&1 | tail -3 >> $out
python - <<'PY' >> $out 2>&1
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "turbo-range-coder_amd")
import trc_testlib as T
d = T.drift_bytes(1000 * 1000 * 1000, 3); d.tofile("/tmp/drift1g.bin"); d[:100 * 1000 * 1000].tofile("/tmp/drift100m.bin"); d[:10 * 1000 * 1000].tofile("/tmp/drift10m.bin")
PY
for e in "X=1" "TRC_HOST_NO_STRIPE=1"; do
  echo "== $e" >> $out
  env $e timeout 300 ./harness/trcbench -I 7 -e 46,47,56,1 --pin /tmp/drift100m.bin 2>&1 | grep -v "^file\|C Size" | sed 's/$/  [pin]/' >> $out
  env $e timeout 300 ./harness/trcbench -I 7 -e 46,47,56,1 /tmp/drift100m.bin 2>&1 | grep -v "^file\|C Size" >> $out
  env $e timeout 300 ./harness/trcbench -I 3 -e 46,56,1 --pin /tmp/drift1g.bin 2>&1 | grep -v "^file\|C Size" | sed 's/$/  [pin 1g]/' >> $out
  env $e timeout 300 ./harness/trcbench -I 3 -e 46 /tmp/drift1g.bin 2>&1 | grep -v "^file\|C Size" | sed 's/$/  [1g]/' >> $out
  env $e timeout 300 ./harness/trcbench -I 7 -e 46,56,1 --pin /tmp/drift10m.bin 2>&1 | grep -v "^file\|C Size" | sed 's/$/  [pin 10m]/' >> $out
done
cat $out
