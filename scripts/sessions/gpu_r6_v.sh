#!/bin/bash
# round 6, session v: copy threads that look for the next piece before they sleep (TRC_COPY_SPIN pause instructions; 0 = sleep at once)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
out=gpurun_out/r06v_spin.txt; : > $out
python - <<'PY' >> $out 2>&1
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "turbo-range-coder_amd")
import trc_testlib as T
T.drift_bytes(100 * 1000 * 1000, 3).tofile("/tmp/drift100m.bin"); T.text_bytes(100 * 1000 * 1000, 7).tofile("/tmp/text100m.bin")
PY
for rep in 1 2 3; do for e in "TRC_COPY_SPIN=0" "TRC_COPY_SPIN=20000" "TRC_COPY_SPIN=100000" "TRC_COPY_SPIN=20000 TRC_HOST_PIECE=16777216" "TRC_COPY_SPIN=20000 TRC_HOST_PIECE=4194304"; do
  echo "== $e" >> $out
  env $e timeout 300 ./harness/trcbench -I 7 -e 46,56 /tmp/drift100m.bin 2>&1 | grep -v "^file\|C Size" >> $out
  env $e timeout 300 ./harness/trcbench -I 7 -e 65 /tmp/text100m.bin 2>&1 | grep -v "^file\|C Size" >> $out
done; done
cat $out
