#!/bin/bash
# PMC passes for the dominant kernel (GPU box).  usage: r2_pmc.sh <outdir-name> [bench args]
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; shift; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --cache /tmp/c3.seg $*"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $B > $O/kt.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq -- $B > $O/pmc_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_WAVES --output-format csv -d $O/pmc_sq2 -- $B > $O/pmc_sq2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1
cd $R
python tools/pmc_summary.py scan_ sq=$O/pmc_sq sq2=$O/pmc_sq2 fetch=$O/pmc_fetch write=$O/pmc_write > $O/pmc_summary.csv
cat $O/pmc_summary.csv
find $O -name "*_kernel_stats.csv" | head -1 | xargs head -8
find $O -name "*.csv" -size +5M -delete
tail -2 $O/pmc_sq2.log
