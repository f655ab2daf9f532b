#!/bin/bash
# round-3 measurement session: everything profiles/r03_* is made from (run through gpurun; every step under its own timeout)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 bash scripts/gpu_profile_round.sh r03 > gpurun_out/r03_round.log 2>&1
python scripts/pmc_traffic.py gpurun_out/r03_pmc_traffic.txt 512 "rocprofv3 --pmc passes of \`python bench.py --steps 3 --warmup 1 --no-cpu\`, TCC_EA0_RDREQ / WRREQ by request size, round 3 final build" gpurun_out/pmc_traffic.json >> gpurun_out/r03_round.log 2>&1
timeout 200 python bench.py --workload mix100m --steps 10 2>/dev/null | grep "^{" | tail -1 > gpurun_out/r03_bench_mix100m.json
timeout 300 python bench.py --workload zipf1g --steps 5 --warmup 2 2>/dev/null | grep "^{" | tail -1 > gpurun_out/r03_bench_zipf1g.json
timeout 200 python bench.py --inflight 2 --no-cpu 2>/dev/null | grep "^{" | tail -1 > gpurun_out/r03_bench_inflight2.json
timeout 200 python bench.py --chunk 1024 --no-cpu 2>/dev/null | grep "^{" | tail -1 > gpurun_out/r03_bench_chunk1024.json
timeout 200 python bench.py --chunk 4096 --no-cpu 2>/dev/null | grep "^{" | tail -1 > gpurun_out/r03_bench_chunk4096.json
timeout 200 python bench.py --force-dist --steps 6 --no-cpu 2>/dev/null | grep "^{" | tail -1 > gpurun_out/r03_bench_forcedist.json
timeout 200 python bench.py --steps 20 --warmup 5 --clock-warmup-ms 0 --no-cpu 2>/dev/null | grep "^{" | tail -1 > gpurun_out/r03_bench_contract_shape.json
timeout 1200 bash scripts/gpu_all_codecs.sh > /dev/null 2>&1; cp gpurun_out/all_codecs.txt gpurun_out/r03_all_codecs.txt 2>/dev/null
ls gpurun_out | grep r03_
