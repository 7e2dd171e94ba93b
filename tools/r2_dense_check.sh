#!/bin/bash
# scan_dense_kernel on the GPU box: its parity tests, step time of the variants on a 10 M-document Zipf corpus, phase timers
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/d4; rm -rf $O; mkdir -p $O
cd $R
timeout -s KILL 240 python -m pytest tests/test_gpu_dense.py -q -x -k "every_query_declared_dense" > $O/pytest_small.log 2>&1
rc=$?; tail -15 $O/pytest_small.log; echo "small rc=$rc"
if [ $rc -eq 137 ]; then echo "HANG in the small test; stopping"; exit 1; fi
timeout -s KILL 600 python -m pytest tests/test_gpu_dense.py -q > $O/pytest_dense.log 2>&1
rc=$?; tail -30 $O/pytest_dense.log; echo "dense rc=$rc"
if [ $rc -eq 137 ]; then echo "HANG in the dense tests; stopping"; exit 1; fi
DENSE_SAMPLE=16 timeout -s KILL 400 python tools/dense_check.py 10000000 100000 512 10 100 /tmp/z10.seg > $O/dense_5m.log 2>&1
tail -12 $O/dense_5m.log
timeout -s KILL 300 python tools/profile_dense.py 10000000 100000 512 10 100 /tmp/z10.seg 2>&1 | grep -v "^--\|waves" | head -30 > $O/prof_5m.log; cat $O/prof_5m.log
