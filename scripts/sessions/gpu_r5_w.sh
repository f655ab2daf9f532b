for v in o1quad; do
  if [ "$v" = "main" ]; then unset TRC_LIB; else export TRC_LIB=$GRAFT_REPO_ROOT/turbo-range-coder_amd/build/ab/lib$v.so; fi
  echo "== $v"; timeout 300 bash scripts/gpu_kstats.sh o1$v "--codec anscdf1 --no-beyond --no-verify --no-cold" 2>&1 | grep "walk"
done
