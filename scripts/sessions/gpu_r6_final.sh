#!/bin/bash
# round 6, evidence of the final tree: the whole -m gpu suite, the default bench line (with host_pointer / configs), rocprofv3 kernel
# stats of the same command, fabric-traffic counter passes (headline; configs 3 / 4 / -e45 / order-1), SQ + texture-address counters of
# the headline decoder, per-coder kernel stats, the all-coders table, the host-pointer table through the plain-C harness
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r06}
timeout 3000 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_gputest_final.log 2>&1; tail -3 gpurun_out/${TAG}_gputest_final.log
timeout 1200 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 300 gpurun_out/${TAG}_bench.json; echo
timeout 300 python bench.py --steps 20 --warmup 5 --clock-warmup-ms 0 --no-cpu --no-beyond > gpurun_out/${TAG}_bench_contract_shape.json 2>/dev/null
rm -rf gpurun_out/prof_$TAG
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o $TAG -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-beyond > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1)
cp gpurun_out/prof_$TAG/${TAG}_kernel_stats.csv gpurun_out/${TAG}_kernel_stats.csv 2>/dev/null; head -6 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-160
bash scripts/gpu_pmc.sh ${TAG}t "" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum" > gpurun_out/${TAG}_pmc_traffic.txt 2>&1
{
for c in rccdf anscdf rcs rccdfs2 anscdf1; do
  echo "#### $c"
  bash scripts/gpu_pmc.sh tr_$c "--codec $c" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum"
done
} > gpurun_out/${TAG}_pmc_traffic_cfg34.txt 2>&1
bash scripts/gpu_pmc.sh ${TAG}sq "" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TA_TA_BUSY_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum GRBM_GUI_ACTIVE" > gpurun_out/${TAG}_pmc_sq.txt 2>&1
for c in rccdf anscdf rcs anscdf1 ansb rccdfs2 rccdfs; do bash scripts/gpu_kstats.sh ${TAG}_$c --codec $c --no-beyond; done > gpurun_out/${TAG}_kernel_stats_cfg34.txt 2>&1
bash scripts/sessions/gpu_r6_o1.sh > /dev/null 2>&1
bash scripts/gpu_all_codecs.sh > /dev/null 2>&1; cp gpurun_out/all_codecs.txt gpurun_out/${TAG}_all_codecs.txt
python bench.py --no-cpu --workload zipf1g 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_zipf1g.json
python bench.py --no-cpu --no-beyond --force-dist --group 8 2>/dev/null | grep "^{" | tail -1 > gpurun_out/${TAG}_bench_forcedist_g8.json
# the host-pointer table through the plain-C harness: pageable and page-locked, 100 MB and 1 GB
python - <<'PY'
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "turbo-range-coder_amd")
import trc_testlib as T
d = T.drift_bytes(1000 * 1000 * 1000, 3); d.tofile("/tmp/drift1g.bin"); d[:100 * 1000 * 1000].tofile("/tmp/drift100m.bin")
t = T.text_bytes(1000 * 1000 * 1000, 7); t.tofile("/tmp/text1g.bin"); t[:100 * 1000 * 1000].tofile("/tmp/text100m.bin")
PY
{
for sz in 100m 1g; do for pin in "" "--pin"; do
  echo "== drift$sz $pin"; timeout 600 ./harness/trcbench -I 5 -e 1,46,47,56,64,66,79 $pin /tmp/drift$sz.bin 2>&1 | grep -v "^file"
  echo "== text$sz $pin"; timeout 600 ./harness/trcbench -I 5 -e 42,44,45,65 $pin /tmp/text$sz.bin 2>&1 | grep -v "^file"
done; done
echo "== reference harness linked against the library (oracle/_ref/turborc_hip), drift100m, -e46,56,1 and text100m -e65"
[ -x oracle/_ref/turborc_hip ] && { timeout 600 oracle/_ref/turborc_hip -I3 -J3 -e46,56,1 /tmp/drift100m.bin 2>&1 | tr '\b' ' ' | grep -v "^$" | tail -6; timeout 600 oracle/_ref/turborc_hip -I3 -J3 -e65 /tmp/text100m.bin 2>&1 | tr '\b' ' ' | grep -v "^$" | tail -3; }
} > gpurun_out/${TAG}_host_pointer.txt 2>&1
grep "rcs \|rccdf \|anscdf \|anscdf1 \|anscdf4s \|rccdfs2 \|rccdfs " gpurun_out/${TAG}_all_codecs.txt | head -16
cat gpurun_out/${TAG}_host_pointer.txt | cut -c1-140
