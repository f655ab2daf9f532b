#!/bin/bash
# rocprofv3 kernel stats + bench line for every secondary coder at the bench chunk -> gpurun_out/r02_codec_*
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for cdc in ${1:-rccdfs rccdfsm rccdfs2 rccdf rccdfi anscdf anscdf1 rcs rccdf4 rccdf4i anscdf4}; do
  rm -rf gpurun_out/prof_$cdc
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$cdc -o $cdc -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --steps 5 --warmup 1 --codec $cdc > $GRAFT_REPO_ROOT/gpurun_out/prof_$cdc.log 2>&1)
  cp gpurun_out/prof_$cdc/${cdc}_kernel_stats.csv gpurun_out/r02_codec_${cdc}_kernel_stats.csv
  timeout 300 python bench.py --no-cpu --steps 10 --warmup 2 --codec $cdc 2>/dev/null | tail -1 > gpurun_out/r02_codec_${cdc}_bench.json
  python -c "
import json,csv
r=json.load(open('gpurun_out/r02_codec_${cdc}_bench.json')); print('$cdc', r['value'], 'enc', r['enc_MBps'], 'dec', r['dec_MBps'], r['config']['ratio'])
for x in list(csv.DictReader(open('gpurun_out/r02_codec_${cdc}_kernel_stats.csv')))[:4]: print('   ', x['Name'][:44], x['Calls'], x['AverageNs'])"
done
