#!/bin/bash
# round 4: coding pass again, histogram, the schedule's own cost (--force-dist --group G --lag L)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
echo "### parity anscdf / anscdf1"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "anscdf and not gigabyte and not host_pointer" 2>&1 | tail -3
bash scripts/gpu_codec_sweep.sh "anscdf" "1536 512 4096"
bash scripts/gpu_kstats.sh r4b_anscdf --codec anscdf --no-beyond
echo "### hist"
python scripts/probe/hist_time.py 2>&1 | tail -1
echo "### schedule cost on one rank: --force-dist --group G --lag L (headline coder)"
python bench.py --no-cpu --no-beyond 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('plain', r['value'], r['ms_per_step'])"
for gl in "1 1" "8 1" "8 4" "8 8"; do set -- $gl
  python bench.py --no-cpu --no-beyond --force-dist --group $1 --lag $2 > gpurun_out/fd.json 2> gpurun_out/fd.err
  tail -1 gpurun_out/fd.json | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('force-dist group $1 lag $2', r['value'], r['ms_per_step'])" || tail -15 gpurun_out/fd.err
done
} > gpurun_out/r04_misc.log 2>&1
cat gpurun_out/r04_misc.log
