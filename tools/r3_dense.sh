#!/bin/bash
# dense kernel iteration: its tests, phase timers and the C5 bench line
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3dense; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_dense.py -x -q -m gpu > $O/pytest_dense.log 2>&1; tail -3 $O/pytest_dense.log
timeout 900 python tools/profile_dense.py 50000000 100000 1024 10 100 /tmp/c5.seg > $O/phases.txt 2>&1; cat $O/phases.txt
timeout 900 python bench.py --workload C5 --no-cpu-baseline --cache /tmp/c5.seg --steps 8 --warmup 2 > $O/bench_c5.json 2> $O/bench_c5.err; python -c "
import json; d=json.load(open('$O/bench_c5.json')); print(d['value'], d['ms_per_step'], d['roofline'])" || tail -5 $O/bench_c5.err
