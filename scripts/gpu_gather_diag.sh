#!/bin/bash
# fresh-box diagnosis of harness/trcgather: real RCCL with one rank first (cold image), then the fake RCCL with peers
mkdir -p gpurun_out; L=gpurun_out/r03_gather.log; : > $L
{
echo "== cold: trcgather --gpus 1 (real RCCL)"; ( time timeout 900 harness/trcgather --gpus 1 --steps 2 --size 30000001 --chunk 512 --watchdog 200 ) 2>&1
echo "== again (warm)"; ( time timeout 300 harness/trcgather --gpus 1 --steps 2 --size 30000001 --chunk 512 ) 2>&1
echo "== small"; ( time timeout 300 harness/trcgather --gpus 1 --steps 2 --size 5000 --chunk 4096 --quiet ) 2>&1
echo "== page-in check: cat librccl"; ( time cat /opt/rocm/lib/librccl.so.1 > /dev/null ) 2>&1
export TRC_RCCL_LIB=$PWD/tests/libfake_rccl.so
for g in 1 2 3 4; do for b in 1 $g; do
echo "== fake: --gpus $g --batches $b"; ( time timeout 300 harness/trcgather --gpus $g --batches $b --steps 2 --size 30000001 --chunk 512 --quiet ) 2>&1
done; done
echo "== fake: ragged, 5 ranks, 3 batches, tiny"; ( timeout 300 harness/trcgather --gpus 5 --batches 3 --steps 1 --size 5000 --chunk 4096 --quiet ) 2>&1
echo "== fake: more ranks than chunks"; ( timeout 300 harness/trcgather --gpus 4 --batches 4 --steps 1 --size 700 --chunk 256 --quiet ) 2>&1
} >> $L 2>&1
tail -60 $L
