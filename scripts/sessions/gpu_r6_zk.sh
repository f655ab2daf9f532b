#!/bin/bash
# round 6, session zk: the order-1 decoder's two row forms at the chunk sizes below the default (explicit trc_set_chunk): is four lanes per chunk right there too?
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
out=gpurun_out/r06zk_o1_rows_chunks.txt; : > $out
for ch in 1024 2048 4096; do for r in 1 4; do
  echo -n "chunk $ch TRC_O1_ROWS=$r: " >> $out
  TRC_O1_ROWS=$r python bench.py --codec anscdf1 --chunk $ch --no-cpu --no-beyond --no-configs --no-host --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('enc %.3f ms dec %.3f ms  %.1f GB/s' % (r['enc_kernel_ms'], r['dec_kernel_ms'], d['value']/1e3))" >> $out
done; done
cat $out
