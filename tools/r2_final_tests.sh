#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; cd $R
timeout -s KILL 900 python -m pytest tests/test_gpu_dense.py tests/test_gpu_flush.py tests/test_cpp_host_mirror.py -m gpu -q 2>&1 | tail -4
timeout -s KILL 900 python -m pytest tests/test_gpu_search.py -m gpu -q -k "not c5_full and not c3_full" 2>&1 | tail -4
for i in 1 2 3; do DBG_K=128 python -u tools/dense_debug.py 2>&1 | cut -c1-120 | head -2; done
