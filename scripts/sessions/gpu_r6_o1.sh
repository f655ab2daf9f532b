#!/bin/bash
# round 6: evidence of the order-1 coder after its round-6 steps: bench line, kernel stats, per-wave clocks of the walk kernel,
# decoder time against input size for the three row forms, SQ counters
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
python bench.py --codec anscdf1 --no-beyond --no-configs --no-host 2>/dev/null | tail -1 > gpurun_out/r06_bench_anscdf1.json; cut -c1-400 gpurun_out/r06_bench_anscdf1.json
bash scripts/gpu_kstats.sh r06o1 --codec anscdf1 --no-beyond --no-configs --no-host > gpurun_out/r06_o1_kstats.txt 2>&1
python - <<'PY' >> gpurun_out/r06_o1_kstats.txt
import csv,glob
f=glob.glob('gpurun_out/ks_r06o1/**/k_kernel_stats.csv',recursive=True)[0]
for x in list(csv.DictReader(open(f)))[:10]: print('   %-70s %4s %10.1f us' % (x['Name'][:70], x['Calls'], float(x['AverageNs'])/1e3))
PY
cat gpurun_out/r06_o1_kstats.txt
[ -f turbo-range-coder_amd/build/ab/libpw4.so ] && TRC_LIB=$PWD/turbo-range-coder_amd/build/ab/libpw4.so timeout 300 python scripts/probe/o1w_prof.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_o1_walk_prof.txt
{ for r in 1 4 2; do echo "== TRC_O1_ROWS=$r  (1: eight lanes per chunk, 4: four, 2: two)"; TRC_O1_ROWS=$r timeout 250 python scripts/probe/o1_dec_sizes.py 2>&1 | grep -v amdgpu.ids; done; } > gpurun_out/r06_o1_dec_sizes.txt
bash scripts/gpu_pmc.sh r06o1 "--codec anscdf1 --no-configs --no-host" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TA_TA_BUSY_sum TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum GRBM_GUI_ACTIVE" > gpurun_out/r06_pmc_o1.txt 2>&1
grep -A30 "o1_dec_rowsn" gpurun_out/r06_pmc_o1.txt | head -12
