#!/bin/bash
# round 6, session i: where the +24 us/step of the G = 8 schedule on one rank go: kernel times per step next to the step time
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
out=gpurun_out/r06i_schedule.txt; : > $out
for a in "" "--force-dist" "--force-dist --group 2" "--force-dist --group 8" "--force-dist --group 8 --lag 4"; do
  env $( [ "$a" = "--force-dist" ] && echo TRC_BENCH_EXCHANGE=root0 || echo X=1 ) timeout 300 python bench.py --steps 40 --no-cpu --no-beyond --no-host --no-configs --no-cold $a 2>/dev/null | python -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r = j['roofline']
print('%-32s ms/step %.4f  enc %.4f dec %.4f gather %.4f  sum %.4f  rest %.4f' % ('$a', j['ms_per_step'], r['enc_kernel_ms'], r['dec_kernel_ms'], r['enc_path']['gather_kernel_ms'], r['enc_kernel_ms'] + r['dec_kernel_ms'] + r['enc_path']['gather_kernel_ms'], j['ms_per_step'] - (r['enc_kernel_ms'] + r['dec_kernel_ms'] + r['enc_path']['gather_kernel_ms'])))" >> $out
done
cat $out
