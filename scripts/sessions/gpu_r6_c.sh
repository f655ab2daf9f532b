#!/bin/bash
# round 6, session c: is the pageable path slower than round 5's on the SAME box?  (build_ab/r5 = the library of commit 491f58f)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
out=gpurun_out/r06c_ab.txt; : > $out
python - <<'PY' >> $out 2>&1
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "turbo-range-coder_amd")
import trc_testlib as T
T.drift_bytes(100 * 1000 * 1000, 3).tofile("/tmp/drift100m.bin")
T.text_bytes(100 * 1000 * 1000, 7).tofile("/tmp/text100m.bin")
PY
run() {
  echo "== $*" >> $out
  env "$@" timeout 300 ./harness/trcbench -I 7 -e 46,1,79 /tmp/drift100m.bin 2>&1 | grep -v "^file\|C Size" >> $out
  env "$@" timeout 300 ./harness/trcbench -I 7 -e 46,1 --pin /tmp/drift100m.bin 2>&1 | grep -v "^file\|C Size" | sed 's/$/  [pin]/' >> $out
  env "$@" timeout 300 ./harness/trcbench -I 7 -e 65,42 /tmp/text100m.bin 2>&1 | grep -v "^file\|C Size" >> $out
  env "$@" timeout 300 ./harness/trcbench -I 7 -e 65,42 --pin /tmp/text100m.bin 2>&1 | grep -v "^file\|C Size" | sed 's/$/  [pin]/' >> $out
}
run LD_LIBRARY_PATH=build_ab/r5 TRC_CHUNK=512
run TRC_CHUNK=512
run TRC_CHUNK=512 TRC_HOST_STREAMS=1
run LD_LIBRARY_PATH=build_ab/r5 TRC_CHUNK=4096
run TRC_CHUNK=4096 TRC_HOST_STREAMS=1
run TRC_CHUNK=4096 TRC_HOST_STREAMS=2
cat $out
