#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c10; rm -rf $O; mkdir -p $O
cd $R
VBM25_LIBRARY=$R/vectorchord-bm25_amd/csrc/libvbm25_chk.so RD_REPS=30 timeout 400 python tools/range_debug.py > $O/dbg_chk.log 2>&1
echo "exit $?"; grep -E "RESULT|assert|entry|rep .* q" $O/dbg_chk.log | tail -6 | cut -c1-300
RD_REPS=30 timeout 400 python tools/range_debug.py > $O/dbg_head.log 2>&1
echo "exit $?"; grep -E "RESULT|assert|entry|rep .* q" $O/dbg_head.log | tail -6 | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_search.py -x -q -m gpu -k "not c5_full and not bench_distributed" > $O/pytest_search.log 2>&1; tail -12 $O/pytest_search.log
timeout 300 python bench.py --no-cpu-baseline --verify --cache /tmp/c3.seg --steps 50 > $O/bench_c3.json 2> $O/bench_c3.err; tail -3 $O/bench_c3.err; cat $O/bench_c3.json
