#!/bin/bash
# the whole GPU suite + stress of the dense kernel + bench lines (C3 verified, C2 latency)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3full; rm -rf $O; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -6 $O/pytest_gpu.log
DS_REPS=30 timeout 600 python tools/dense_stress.py > $O/stress.log 2>&1; echo "stress exit $?"; tail -3 $O/stress.log
timeout 600 python bench.py --no-cpu-baseline --verify --cache /tmp/c3.seg --steps 200 > $O/bench_c3.json 2> $O/bench_c3.err; python -c "
import json; d=json.load(open('$O/bench_c3.json')); print(d['value'], d['ms_per_step'], d['roofline'], d['config'].get('verified_bit_exact_vs_oracle'))" || tail -5 $O/bench_c3.err
timeout 600 python bench.py --no-cpu-baseline --cache /tmp/c3.seg --steps 200 --batches 1 > $O/bench_c3_b1.json 2> $O/bench_c3_b1.err; python -c "
import json; d=json.load(open('$O/bench_c3_b1.json')); print('one batch:', d['value'], d['ms_per_step'], d['roofline']['frac'])"
timeout 300 python bench.py --workload C2 --no-cpu-baseline --steps 200 > $O/bench_c2.json 2> $O/bench_c2.err; python -c "
import json; d=json.load(open('$O/bench_c2.json')); print(d['value'], d['ms_per_step'], d['config']['latency'])" || tail -5 $O/bench_c2.err
