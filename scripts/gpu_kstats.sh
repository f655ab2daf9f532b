#!/bin/bash
# rocprofv3 kernel stats of one bench configuration -> top kernels; usage: gpu_kstats.sh TAG "bench args"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=$1; shift
rm -rf gpurun_out/ks_$TAG
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ks_$TAG -o k -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --steps 5 --warmup 1 $@ > $GRAFT_REPO_ROOT/gpurun_out/ks_$TAG.log 2>&1)
tail -1 gpurun_out/ks_$TAG.log | cut -c1-200
python - <<PY
import csv,glob
f=glob.glob('gpurun_out/ks_$TAG/**/k_kernel_stats.csv',recursive=True)[0]
for x in list(csv.DictReader(open(f)))[:6]: print('   %-60s %4s %10.1f us' % (x['Name'][:60], x['Calls'], float(x['AverageNs'])/1e3))
PY
