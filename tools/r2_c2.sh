#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c2; rm -rf $O; mkdir -p $O
cd $R
timeout -s KILL 900 python -m pytest tests/test_gpu_search.py -q -x -k "c1 or c2 or tiny or edge or properties or mixed or correlated or codec or sqllogic or growing or routing" > $O/pytest.log 2>&1; tail -8 $O/pytest.log
timeout -s KILL 300 python bench.py --workload C2 --steps 200 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; tail -2 $O/bench_c2.err; python -c "
import json;d=json.loads(open('$O/bench_c2.json').read().strip().splitlines()[-1]);print('C2',d['value'],d['ms_per_step'],d['config'].get('latency'))"
VBM25_FUSED=0 timeout -s KILL 300 python bench.py --workload C2 --steps 200 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('C2 unfused',d['value'],d['ms_per_step'],d['config'].get('latency'))"
