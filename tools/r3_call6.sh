#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c6; rm -rf $O; mkdir -p $O
cd $R
VBM25_LIBRARY=$R/vectorchord-bm25_amd/csrc/libvbm25_lb2.so RD_REPS=30 timeout 400 python tools/range_debug.py > $O/dbg_lb2.log 2>&1; grep -E "differ|RESULT|rep .* q" $O/dbg_lb2.log | grep -v " 0 of" | tail -12
RD_REPS=30 timeout 400 python tools/range_debug.py > $O/dbg_head.log 2>&1; grep -E "differ|RESULT|rep .* q" $O/dbg_head.log | grep -v " 0 of" | tail -12
