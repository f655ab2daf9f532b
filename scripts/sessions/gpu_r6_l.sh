#!/bin/bash
# round 6, session l: encode with a short last slice (80 + 20) against even slices (TRC_HOST_EVEN=1)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
out=gpurun_out/r06l_uneven.txt; : > $out
python - <<'PY' >> $out 2>&1
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "turbo-range-coder_amd")
import trc_testlib as T
d = T.drift_bytes(1000 * 1000 * 1000, 3); d.tofile("/tmp/drift1g.bin"); d[:100 * 1000 * 1000].tofile("/tmp/drift100m.bin")
PY
for rep in 1 2; do
for e in "TRC_HOST_EVEN=1" "X=1"; do
  echo "== $e" >> $out
  env $e timeout 300 ./harness/trcbench -I 7 -e 46,56,1 --pin /tmp/drift100m.bin 2>&1 | grep -v "^file\|C Size" | sed 's/$/  [pin]/' >> $out
  env $e timeout 300 ./harness/trcbench -I 7 -e 46,56,1 /tmp/drift100m.bin 2>&1 | grep -v "^file\|C Size" >> $out
  env $e timeout 300 ./harness/trcbench -I 3 -e 46 --pin /tmp/drift1g.bin 2>&1 | grep -v "^file\|C Size" | sed 's/$/  [pin 1g]/' >> $out
done
done
timeout 600 python -m pytest tests/test_gpu_host_layer.py -q -m gpu -x 2>&1 | tail -3 >> $out
cat $out
