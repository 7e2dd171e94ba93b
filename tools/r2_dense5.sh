#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/d5; rm -rf $O; mkdir -p $O
cd $R
timeout -s KILL 600 python bench.py --workload C5 --steps 5 --warmup 1 --no-cpu-baseline --cache /tmp/c5.seg > $O/bench_c5.json 2> $O/bench_c5.err; tail -3 $O/bench_c5.err; cat $O/bench_c5.json
timeout -s KILL 300 python tools/profile_dense.py 50000000 100000 1024 10 100 /tmp/c5.seg 2>&1 | head -16 > $O/prof_c5.log; cat $O/prof_c5.log
bash tools/r2_quick.sh
