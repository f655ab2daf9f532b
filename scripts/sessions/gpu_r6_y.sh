#!/bin/bash
# round 6, session y: what the fan-out over a device list costs when all entries are ONE device (one link, one GPU: no gain possible)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
out=gpurun_out/r06y_devlist.txt; : > $out
python - <<'PY' >> $out 2>&1
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "turbo-range-coder_amd")
import trc_testlib as T
T.drift_bytes(100 * 1000 * 1000, 3).tofile("/tmp/drift100m.bin"); T.text_bytes(100 * 1000 * 1000, 7).tofile("/tmp/text100m.bin")
PY
for d in "" "0" "0,0" "0,0,0,0" "0,0,0,0,0,0,0,0"; do
  echo "== TRC_DEVICES=$d" >> $out
  TRC_DEVICES=$d timeout 300 ./harness/trcbench -I 5 -e 46,1 --pin /tmp/drift100m.bin 2>&1 | grep -v "^file\|C Size" | sed 's/$/  [pin]/' >> $out
  TRC_DEVICES=$d timeout 300 ./harness/trcbench -I 5 -e 65,45 --pin /tmp/text100m.bin 2>&1 | grep -v "^file\|C Size" | sed 's/$/  [pin]/' >> $out
  TRC_DEVICES=$d timeout 300 ./harness/trcbench -I 5 -e 46 /tmp/drift100m.bin 2>&1 | grep -v "^file\|C Size" >> $out
  TRC_DEVICES=$d timeout 300 ./harness/trcbench -I 5 -e 65 /tmp/text100m.bin 2>&1 | grep -v "^file\|C Size" >> $out
done
cat $out
