#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
echo "--- default"; bash scripts/gpu_kstats.sh scan_def "--no-beyond" 
echo "--- TRC_SCAN_MAX=0"; TRC_SCAN_MAX=0 bash scripts/gpu_kstats.sh scan_0 "--no-beyond"
done
