#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/d8; rm -rf $O; mkdir -p $O
cd $R
timeout -s KILL 600 python -m pytest tests/test_gpu_dense.py -q > $O/pytest_dense.log 2>&1; tail -3 $O/pytest_dense.log
timeout -s KILL 600 python bench.py --workload C5 --steps 5 --warmup 1 --no-cpu-baseline --cache /tmp/c5.seg > $O/bench_c5.json 2> $O/bench_c5.err; tail -2 $O/bench_c5.err; python -c "
import json;d=json.loads(open('$O/bench_c5.json').read().strip().splitlines()[-1]);print('C5',d['value'],d['ms_per_step'],d['roofline'])"
timeout -s KILL 300 python tools/profile_dense.py 50000000 100000 1024 10 100 /tmp/c5.seg 2>&1 | head -14 > $O/prof_c5.log; cat $O/prof_c5.log
