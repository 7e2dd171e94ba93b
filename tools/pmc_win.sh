R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5b; mkdir -p $O; cd $R
B="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --extra-budget-s 0"
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/stats.log 2>&1
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/c3_kernel_stats.csv && head -6 $O/c3_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/pmc_sq1 -- $B > $O/pmc_sq1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_IFETCH --output-format csv -d $O/pmc_sq2 -- $B > $O/pmc_sq2.log 2>&1
cd $R
python tools/pmc_summary.py scan_win_kernel sq1=$O/pmc_sq1 sq2=$O/pmc_sq2 > $O/pmc_summary.csv 2> $O/pmc_summary.err; cat $O/pmc_summary.csv; tail -3 $O/pmc_summary.err
find $O -name "*.csv" -size +5M -delete
