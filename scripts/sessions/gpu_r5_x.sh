# gather: flat loads (pointer from an integer in LDS) vs global loads
for rep in 1 2 3; do for v in main gflat; do
  if [ "$v" = "main" ]; then unset TRC_LIB; else export TRC_LIB=$GRAFT_REPO_ROOT/turbo-range-coder_amd/build/ab/lib$v.so; fi
  echo "== $v"; timeout 300 bash scripts/gpu_kstats.sh g$v "--no-beyond" 2>&1 | grep "gather\|ms_per_step" | cut -c1-120
done; done
