# order-1 chains: where the time goes (ablation builds; results of the ablated builds are wrong by construction)
for v in o1skipb o1norec o1noent; do
  echo "$v"
  TRC_LIB=turbo-range-coder_amd/build/ab/lib$v.so python bench.py --codec anscdf1 --no-cpu --no-beyond --no-check --steps 5 --warmup 2 2>&1 | grep "^{" | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print(j['value'], j['ms_per_step'], r['enc_kernel_ms'], r['dec_kernel_ms'])"
done
