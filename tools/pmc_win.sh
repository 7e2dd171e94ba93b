# counter passes on scan_win_kernel: bash tools/pmc_win.sh <out-subdir> [bench.py --tune value]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r5b}; mkdir -p $O; cd $R
T=""; [ -n "${2:-}" ] && T="--tune $2"
B="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-verify-sample --extra-budget-s 0 $T"
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/pmc_sq1 -- $B > $O/pmc_sq1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $O/pmc_sq2 -- $B > $O/pmc_sq2.log 2>&1
cd $R
python tools/pmc_summary.py scan_win_kernel sq1=$O/pmc_sq1 sq2=$O/pmc_sq2 > $O/pmc_summary.csv 2> $O/pmc_summary.err; cat $O/pmc_summary.csv; tail -3 $O/pmc_summary.err
find $O -name "*.csv" -size +5M -delete
