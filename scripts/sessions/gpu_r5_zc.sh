#!/bin/bash
# round 5: full GPU suite on the tree with the rows decoder as default, then kernel stats of anscdf1 (where the encoder's 2.8 ms go)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5zc_gputest.log 2>&1; tail -3 gpurun_out/r5zc_gputest.log
bash scripts/gpu_kstats.sh zc_o1 --codec anscdf1 --no-beyond 2>&1 | tail -12
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/ks_zc_o1/**/k_kernel_stats.csv',recursive=True)[0]
for x in list(csv.DictReader(open(f)))[:14]: print('   %-70s %4s %10.1f us' % (x['Name'][:70], x['Calls'], float(x['AverageNs'])/1e3))
PY
