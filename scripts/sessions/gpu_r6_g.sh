#!/bin/bash
# round 6, session g: A/B on one box, pageable buffers: r6a = slices as the unit of copies, new = pieces + out-worker
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
out=gpurun_out/r06g_ab.txt; : > $out
python - <<'PY' >> $out 2>&1
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "turbo-range-coder_amd")
import trc_testlib as T
T.drift_bytes(100 * 1000 * 1000, 3).tofile("/tmp/drift100m.bin")
T.text_bytes(100 * 1000 * 1000, 7).tofile("/tmp/text100m.bin")
PY
nproc >> $out; lscpu | grep -i "model name\|numa node(s)\|socket" >> $out
run() {
  echo "== $*" >> $out
  env "$@" timeout 300 ./harness/trcbench -I 9 -e 46,1 /tmp/drift100m.bin 2>&1 | grep -v "^file\|C Size" >> $out
  env "$@" timeout 300 ./harness/trcbench -I 9 -e 65,42 /tmp/text100m.bin 2>&1 | grep -v "^file\|C Size" >> $out
}
for rep in 1 2 3; do
  run LD_LIBRARY_PATH=build_ab/r6a
  run X=1
done
run TRC_HOST_PIECE=33554432
run TRC_COPY_THREADS=3
run TRC_COPY_THREADS=12
cat $out
