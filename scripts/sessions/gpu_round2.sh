#!/bin/bash
# round-2 measurement session: everything profiles/r02_* is made from (run through gpurun; every step under its own timeout)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 bash scripts/gpu_profile_round.sh r02 > gpurun_out/r02_round.log 2>&1
timeout 200 python bench.py --workload mix100m --steps 10 2>/dev/null | grep "^{" | tail -1 > gpurun_out/r02_bench_mix100m.json
timeout 300 python bench.py --workload zipf1g --steps 5 --warmup 2 2>/dev/null | grep "^{" | tail -1 > gpurun_out/r02_bench_zipf1g.json
timeout 200 python bench.py --inflight 2 --no-cpu 2>/dev/null | grep "^{" | tail -1 > gpurun_out/r02_bench_inflight2.json
timeout 200 python bench.py --inflight 3 --chunk 1024 --no-cpu 2>/dev/null | grep "^{" | tail -1 > gpurun_out/r02_bench_inflight3_chunk1024.json
timeout 200 python bench.py --chunk 1024 --no-cpu 2>/dev/null | grep "^{" | tail -1 > gpurun_out/r02_bench_chunk1024.json
timeout 200 python bench.py --force-dist --steps 6 --no-cpu 2>/dev/null | grep "^{" | tail -1 > gpurun_out/r02_bench_forcedist.json
timeout 900 bash scripts/gpu_all_codecs.sh > /dev/null 2>&1
timeout 900 bash scripts/gpu_profile_codecs.sh "rccdfs2 rccdf anscdf rcs" > gpurun_out/r02_codecs.log 2>&1
{ TRC_CHUNK=512 timeout 200 ./harness/trcbench -I 5 -e 1,42,45,46,56,65,79 --text 100000000; timeout 200 ./harness/trcbench -I 5 -e 42,45,65,79 --text 100000000; } > gpurun_out/r02_host_pointer.txt 2>&1
timeout 120 ./harness/trcgather --gpus 1 --steps 5 2>&1 | tail -1 > gpurun_out/r02_trcgather.txt
(bash scripts/gpu_pmc_sq.sh sq_r02 "") > gpurun_out/r02_pmc_sq.txt 2>&1
ls gpurun_out | grep r02
