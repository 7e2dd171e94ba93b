#!/bin/bash
# round-2 measurement (GPU box): the driver's bench line, C5, C2, rocprofv3 kernel stats + PMC passes for C3
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2m; mkdir -p $O
cd $R
python bench.py --verify --cache /tmp/c3.seg > $O/bench_c3.json 2> $O/bench_c3.err; tail -2 $O/bench_c3.err; cat $O/bench_c3.json
python bench.py --workload C2 --steps 200 > $O/bench_c2.json 2> $O/bench_c2.err; cat $O/bench_c2.json
python bench.py --workload C5 --steps 5 --warmup 1 > $O/bench_c5.json 2> $O/bench_c5.err; tail -2 $O/bench_c5.err; cat $O/bench_c5.json
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --cache /tmp/c3.seg"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $B > $O/kt.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq -- $B > $O/pmc_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1
cd $R
python tools/pmc_summary.py scan_range_kernel sq=$O/pmc_sq fetch=$O/pmc_fetch write=$O/pmc_write > $O/pmc_summary.csv
cat $O/pmc_summary.csv
find $O -name "*_kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; head -8 $O/kernel_stats.csv
find $O -name "*.csv" -size +5M -delete
lscpu | head -20 > $O/lscpu.txt
