#!/bin/bash
# round 4: four-lanes-per-chunk rANS coding pass + kernel stats of the config 3 / 4 coders + chunk-policy test + schedule cost
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
echo "### parity (anscdf, anscdf1: the coders behind the coding pass)"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "anscdf and not bench_config and not gigabyte and not host_pointer" 2>&1 | tail -4
for q in 1 0; do
  echo "### TRC_ANSA_CODEQ=$q"
  TRC_ANSA_CODEQ=$q bash scripts/gpu_codec_sweep.sh "anscdf" "1536 512 4096"
done
echo "### kernel stats (rocprofv3), default chunk"
for c in anscdf rccdf rcs; do bash scripts/gpu_kstats.sh r4_$c --codec $c --no-beyond; done
echo "### total parity at the bench configurations (reference SHA-256)"
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bench_config" 2>&1 | tail -4
echo "### chunk policy"
timeout 1500 python -m pytest tests/test_gpu_chunk_policy.py -x -q -m gpu -s 2>&1 | tail -25
echo "### schedule cost on one rank: --force-dist --group G --lag L (headline coder)"
python bench.py --no-cpu --no-beyond 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('plain', r['value'], r['ms_per_step'])"
for gl in "1 1" "8 1" "8 4" "8 8"; do set -- $gl
  python bench.py --no-cpu --no-beyond --force-dist --group $1 --lag $2 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('force-dist group $1 lag $2', r['value'], r['ms_per_step'])"
done
} > gpurun_out/r04_codeq.log 2>&1
cat gpurun_out/r04_codeq.log
