#!/bin/bash
# Round-3 measurement run: everything under profiles/r3_* comes from ONE call of this script on one build.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3final; rm -rf $O; mkdir -p $O
cd $R
git_rev=$(cat .git_rev 2>/dev/null || echo unknown)
echo "build: $(sha256sum vectorchord-bm25_amd/csrc/libvbm25.so | cut -c1-16)  $(date -u +%FT%TZ)" > $O/build.txt
lscpu > $O/lscpu.txt 2>&1
# ---- C3: the driver's line (+ --verify of all four batches), C2 latencies
timeout 900 python bench.py --verify --cache /tmp/c3.seg > $O/c3_bench.json 2> $O/c3_bench.err; tail -c 600 $O/c3_bench.json; echo
timeout 600 python bench.py --workload C2 > $O/c2_bench.json 2> $O/c2_bench.err; tail -c 700 $O/c2_bench.json; echo
# ---- C3: kernel trace + PMC passes (separate runs, --kernel-trace only)
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --cache /tmp/c3.seg"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3_stats -- $B > $O/c3_stats.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/c3_pmc1 -- $B > $O/c3_pmc1.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $O/c3_pmc2 -- $B > $O/c3_pmc2.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE FETCH_SIZE WRITE_SIZE --output-format csv -d $O/c3_pmc3 -- $B > $O/c3_pmc3.log 2>&1
cd $R
python tools/pmc_summary.py scan_range_kernel sq1=$O/c3_pmc1 sq2=$O/c3_pmc2 mem=$O/c3_pmc3 > $O/c3_pmc_scan_range_kernel.csv 2> $O/c3_pmc.err
cat $O/c3_pmc_scan_range_kernel.csv
find $O/c3_stats -name "*kernel_stats.csv" -exec cp {} $O/c3_kernel_stats.csv \;
head -8 $O/c3_kernel_stats.csv
# ---- C5
timeout 1200 python bench.py --workload C5 --cache /tmp/c5.seg --steps 8 --warmup 2 > $O/c5_bench.json 2> $O/c5_bench.err; tail -c 900 $O/c5_bench.json; echo
cd /tmp
B5="python $R/bench.py --workload C5 --steps 3 --warmup 1 --batches 2 --no-cpu-baseline --cache /tmp/c5.seg"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5_stats -- $B5 > $O/c5_stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/c5_pmc1 -- $B5 > $O/c5_pmc1.log 2>&1
cd $R
python tools/pmc_summary.py scan_dense_kernel mem=$O/c5_pmc1 > $O/c5_pmc_scan_dense_kernel.csv 2> $O/c5_pmc.err
cat $O/c5_pmc_scan_dense_kernel.csv
find $O/c5_stats -name "*kernel_stats.csv" -exec cp {} $O/c5_kernel_stats.csv \;
head -6 $O/c5_kernel_stats.csv
# big raw traces do not travel
find $O -name "*.csv" -size +3M -delete; find $O -name "*.db" -delete; du -sh $O
