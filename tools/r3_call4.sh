#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c4; rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_search.py -x -q -m gpu -k "not c5_full and not c3_full and not bench_distributed" > $O/pytest_search.log 2>&1; tail -25 $O/pytest_search.log
timeout 300 python bench.py --no-cpu-baseline --verify --cache /tmp/c3.seg --steps 50 > $O/bench_c3.json 2> $O/bench_c3.err; tail -5 $O/bench_c3.err; cat $O/bench_c3.json
