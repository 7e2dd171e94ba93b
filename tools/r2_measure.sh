#!/bin/bash
# round-2 measurement (GPU box, one call): the driver's bench line (+ --verify), C2, C5, rocprofv3 kernel stats + PMC passes
# (every --pmc pass is a separate run with --kernel-trace only), build-time comparison of the host and device builders
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2m; rm -rf $O; mkdir -p $O
cd $R
python bench.py --verify --cache /tmp/c3.seg > $O/bench_c3.json 2> $O/bench_c3.err; tail -2 $O/bench_c3.err; cat $O/bench_c3.json
python bench.py --workload C2 --steps 200 > $O/bench_c2.json 2> $O/bench_c2.err; cat $O/bench_c2.json
python bench.py --workload C5 --steps 5 --warmup 1 --cache /tmp/c5.seg > $O/bench_c5.json 2> $O/bench_c5.err; tail -2 $O/bench_c5.err; cat $O/bench_c5.json
timeout 200 python tools/flush_timing.py 1000000 > $O/flush_timing.txt 2>&1; cat $O/flush_timing.txt
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --cache /tmp/c3.seg"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $B > $O/kt.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq -- $B > $O/pmc_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1
B5="python $R/bench.py --workload C5 --steps 3 --warmup 1 --no-cpu-baseline --cache /tmp/c5.seg"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt5 -- $B5 > $O/kt5.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch5 -- $B5 > $O/pmc_fetch5.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write5 -- $B5 > $O/pmc_write5.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq5 -- $B5 > $O/pmc_sq5.log 2>&1
VBM25_NE=0 timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch5_ne0 -- $B5 > $O/pmc_fetch5_ne0.log 2>&1
cd $R
python tools/pmc_summary.py scan_range_kernel sq=$O/pmc_sq fetch=$O/pmc_fetch write=$O/pmc_write > $O/pmc_summary.csv
python tools/pmc_summary.py scan_dense_kernel sq=$O/pmc_sq5 fetch=$O/pmc_fetch5 write=$O/pmc_write5 fetch_without_maxscore_split=$O/pmc_fetch5_ne0 > $O/pmc_summary_c5.csv
cat $O/pmc_summary.csv $O/pmc_summary_c5.csv
find $O/kt -name "*_kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; head -6 $O/kernel_stats.csv
find $O/kt5 -name "*_kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_c5.csv; head -5 $O/kernel_stats_c5.csv
find $O -name "*.csv" -size +5M -delete
lscpu | head -20 > $O/lscpu.txt
