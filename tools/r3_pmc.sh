#!/bin/bash
# PMC table of scan_range_kernel on C3 (separate --pmc passes, --kernel-trace only)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3pmc; rm -rf $O; mkdir -p $O
cd $R
python bench.py --no-cpu-baseline --cache /tmp/c3.seg --steps 20 > $O/bench_c3.json 2> $O/bench_c3.err
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --cache /tmp/c3.seg"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/pmc_sq1 -- $B > $O/pmc_sq1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_IFETCH --output-format csv -d $O/pmc_sq2 -- $B > $O/pmc_sq2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH --output-format csv -d $O/pmc_sq3 -- $B > $O/pmc_sq3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1
cd $R
python tools/pmc_summary.py scan_range_kernel sq1=$O/pmc_sq1 sq2=$O/pmc_sq2 sq3=$O/pmc_sq3 fetch=$O/pmc_fetch > $O/pmc_summary.csv 2> $O/pmc_summary.err
cat $O/pmc_summary.csv; tail -2 $O/pmc_sq3.log
rocprofv3 -L 2>/dev/null | grep -iE "ICACHE|IFETCH|INST_LEVEL|STALL" | head -30 > $O/counters.txt; head -30 $O/counters.txt
find $O -name "*.csv" -size +5M -delete
