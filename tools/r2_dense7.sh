#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/d7; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="python $R/tools/dense_check.py 10000000 100000 512 10 100 /tmp/z10.seg"
export DENSE_NO_OLD=1 DENSE_SAMPLE=1 DENSE_STEPS=2
$B > $O/plain.log 2>&1; grep "^{}" $O/plain.log
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq -- $B > $O/pmc_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --output-format csv -d $O/pmc_sq2 -- $B > $O/pmc_sq2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1
cd $R
python tools/pmc_summary.py scan_dense_kernel sq=$O/pmc_sq sq2=$O/pmc_sq2 fetch=$O/pmc_fetch > $O/pmc_summary.csv 2>&1; cat $O/pmc_summary.csv
tail -3 $O/pmc_sq2.log
find $O -name "*.csv" -size +2M -delete
