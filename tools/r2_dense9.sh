#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/d9; rm -rf $O; mkdir -p $O
cd $R
for items in 4096 8192 16384 2048; do
  VBM25_DENSE_ITEMS=$items timeout -s KILL 600 python bench.py --workload C5 --steps 5 --warmup 1 --no-cpu-baseline --cache /tmp/c5.seg 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('C5 items $items',d['value'],d['ms_per_step'],d['roofline']['kernel_ms'])"
done
timeout -s KILL 900 python -m pytest tests/test_gpu_search.py -q -x -k "c5 or one_launch or mixed or c3" 2>&1 | tail -3
