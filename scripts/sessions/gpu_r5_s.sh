# order-1 model pass by chains: parity + A/B
timeout 1500 python -m pytest tests -m gpu -x -q -k "anscdf1 or ANSO1 or o1 or order1" 2>&1 | tail -5
for v in 0 1; do
  echo "TRC_O1_CHAINS=$v"
  TRC_O1_CHAINS=$v python bench.py --codec anscdf1 --no-cpu --no-beyond --steps 5 --warmup 2 2>/dev/null | grep "^{" | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print(j['value'], j['ms_per_step'], r['enc_kernel_ms'], r['dec_kernel_ms'], j.get('payload_matches_reference_sha256'))"
done
