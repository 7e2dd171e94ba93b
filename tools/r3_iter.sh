#!/bin/bash
# one development iteration on the GPU: assertion build on the stress shape, the search tests, phase timers, bench line
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3iter; rm -rf $O; mkdir -p $O
cd $R
VBM25_LIBRARY=$R/vectorchord-bm25_amd/csrc/libvbm25_chk.so RD_REPS=10 timeout 300 python tools/range_debug.py > $O/dbg_chk.log 2>&1
echo "chk exit $?"; grep -E "RESULT|assert|entry|rep .* q" $O/dbg_chk.log | tail -4 | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_search.py -x -q -m gpu -k "not c5_full and not bench_distributed" > $O/pytest_search.log 2>&1; tail -4 $O/pytest_search.log
timeout 300 python tools/profile_range.py C3 /tmp/c3.seg > $O/phases.txt 2>&1; cat $O/phases.txt
timeout 300 python bench.py --no-cpu-baseline --verify --cache /tmp/c3.seg --steps 50 > $O/bench_c3.json 2> $O/bench_c3.err; python -c "
import json; d=json.load(open('$O/bench_c3.json')); print(d['value'], d['ms_per_step'], d['roofline'], d['config'].get('verified_bit_exact_vs_oracle'))" || tail -5 $O/bench_c3.err
