#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/d2; rm -rf $O; mkdir -p $O
cd $R
timeout -s KILL 300 python tools/profile_dense.py 5000000 100000 256 10 100 /tmp/z5.seg > $O/prof_5m.log 2>&1; cat $O/prof_5m.log
VBM25_NE=0 timeout -s KILL 300 python tools/profile_dense.py 5000000 100000 256 10 100 /tmp/z5.seg > $O/prof_5m_ne0.log 2>&1; cat $O/prof_5m_ne0.log
