#!/bin/bash
# round 6, session u: copy threads per direction for pageable callers (2 x 64-core host)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
out=gpurun_out/r06u_threads.txt; : > $out
python - <<'PY' >> $out 2>&1
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "turbo-range-coder_amd")
import trc_testlib as T
T.drift_bytes(100 * 1000 * 1000, 3).tofile("/tmp/drift100m.bin"); T.text_bytes(100 * 1000 * 1000, 7).tofile("/tmp/text100m.bin")
PY
for rep in 1 2; do for n in 6 10 16 24 32 48; do
  echo "== TRC_COPY_THREADS=$n" >> $out
  TRC_COPY_THREADS=$n timeout 300 ./harness/trcbench -I 7 -e 46,56,1 /tmp/drift100m.bin 2>&1 | grep -v "^file\|C Size" >> $out
  TRC_COPY_THREADS=$n timeout 300 ./harness/trcbench -I 7 -e 65,45 /tmp/text100m.bin 2>&1 | grep -v "^file\|C Size" >> $out
done; done
cat $out
