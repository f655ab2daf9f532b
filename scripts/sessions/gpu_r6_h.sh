#!/bin/bash
# round 6, session h: N ranks on one GPU (gloo-staged exchange) + the schedule's own cost at G = 8 on one rank
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
out=gpurun_out/r06h_multirank.txt; : > $out
timeout 2400 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu 2>&1 | tail -30 >> $out
echo "== schedule cost on one rank: plain, --force-dist G=1 (root0), --force-dist --group 8 (rotate)" >> $out
for a in "" "--force-dist" "--force-dist --group 8" "--force-dist --group 8 --lag 4"; do
  env $( [ "$a" = "--force-dist" ] && echo TRC_BENCH_EXCHANGE=root0 || echo X=1 ) timeout 300 python bench.py --no-cpu --no-beyond --no-host --no-configs --no-cold $a 2>/dev/null | python -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-36s value %9.1f  ms/step %.4f  cold-clocks %.4f' % ('$a', j['value'], j['ms_per_step'], j['ms_per_step_cold_clocks']))" >> $out
done
cat $out
