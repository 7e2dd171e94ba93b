#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3q; rm -rf $O; mkdir -p $O
cd $R
timeout 300 python tools/profile_range.py C3 /tmp/c3.seg > $O/phases.txt 2>&1; cat $O/phases.txt
timeout 300 python bench.py --no-cpu-baseline --verify --cache /tmp/c3.seg --steps 50 > $O/bench_c3.json 2> $O/bench_c3.err; python -c "
import json; d=json.load(open('$O/bench_c3.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['config'].get('verified_bit_exact_vs_oracle'))" || tail -5 $O/bench_c3.err
