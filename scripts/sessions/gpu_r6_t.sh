#!/bin/bash
# round 6, session t: the 1 GB anscdf decode under streaming: part size
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
out=gpurun_out/r06t_parts.txt; : > $out
python - <<'PY' >> $out 2>&1
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "turbo-range-coder_amd")
import trc_testlib as T
d = T.drift_bytes(1000 * 1000 * 1000, 3); d.tofile("/tmp/drift1g.bin"); d[:100 * 1000 * 1000].tofile("/tmp/drift100m.bin")
PY
for e in "X=1" "TRC_HOST_PART=2048" "TRC_HOST_PART=4096" "TRC_HOST_PART=8192" "TRC_HOST_NO_STRIPE=1"; do
  echo "== $e" >> $out
  env $e timeout 300 ./harness/trcbench -I 3 -e 46,56,1 --pin /tmp/drift1g.bin 2>&1 | grep -v "^file\|C Size" | sed 's/$/  [pin 1g]/' >> $out
done
for e in "X=1" "TRC_HOST_PART=2048" "TRC_HOST_PART=512"; do
  echo "== $e" >> $out
  env $e timeout 300 ./harness/trcbench -I 7 -e 46,56,1 --pin /tmp/drift100m.bin 2>&1 | grep -v "^file\|C Size" | sed 's/$/  [pin]/' >> $out
done
cat $out
