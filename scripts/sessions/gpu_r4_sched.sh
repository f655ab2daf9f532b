#!/bin/bash
# round 4: the exchange schedule's own cost on one rank (--force-dist --group G --lag L: the 8-GPU schedule with nothing to send)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
one() { python bench.py --no-cpu --no-beyond "$@" 2>gpurun_out/fd.err | grep '^{' | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('%-44s %9.1f MB/s  %.4f ms/step' % ('$*', r['value'], r['ms_per_step']))" || tail -5 gpurun_out/fd.err; }
for rep in 1 2; do
one
one --force-dist --group 1 --lag 1
one --force-dist --group 8 --lag 1
one --force-dist --group 8 --lag 4
one --force-dist --group 8 --lag 8
done
echo "### hist"; python scripts/probe/hist_time.py 2>&1 | tail -1
} > gpurun_out/r04_sched.log 2>&1
cat gpurun_out/r04_sched.log
