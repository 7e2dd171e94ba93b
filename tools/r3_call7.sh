#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c7; rm -rf $O; mkdir -p $O
cd $R
VBM25_LIBRARY=$R/vectorchord-bm25_amd/csrc/libvbm25_chk.so RD_REPS=12 timeout 400 python tools/range_debug.py > $O/dbg_chk.log 2>&1; grep -E "differ|RESULT|rep .* q|assert|got" $O/dbg_chk.log | grep -v " 0 of" | tail -24
