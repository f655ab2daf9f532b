# order-1 chains vs position-order model pass on the three workload kinds
for inp in drift text bwt; do for v in 0 1; do
  echo -n "$inp TRC_O1_CHAINS=$v: "
  TRC_O1_CHAINS=$v python bench.py --codec anscdf1 --input $inp --no-cpu --no-beyond --steps 5 --warmup 2 2>/dev/null | grep "^{" | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print(j['value'], j['ms_per_step'], r['enc_kernel_ms'], r['dec_kernel_ms'])"
done; done
