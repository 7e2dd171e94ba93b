#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fz; rm -rf $O; mkdir -p $O
cd $R
timeout -s KILL 900 python -m pytest tests/test_gpu_search.py -q -x -k "not c5 and not c3_full" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python bench.py --no-cpu-baseline --steps 200 --verify --cache /tmp/c3.seg 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('C3 fused',d['value'],d['ms_per_step'],d['roofline']['kernel_ms'],d['roofline']['frac'],d['config'].get('verified_bit_exact_vs_oracle'))"
VBM25_FUSED=0 python bench.py --no-cpu-baseline --steps 200 --cache /tmp/c3.seg 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('C3 general',d['value'],d['ms_per_step'],d['roofline']['kernel_ms'],d['roofline']['frac'])"
python bench.py --workload C2 --steps 200 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('C2',d['value'],d['ms_per_step'],d['config'].get('latency'))"
