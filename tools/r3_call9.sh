#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c9; rm -rf $O; mkdir -p $O
cd $R
VBM25_LIBRARY=$R/vectorchord-bm25_amd/csrc/libvbm25_chk.so RD_REPS=40 timeout 400 python tools/range_debug.py > $O/dbg_chk.log 2>&1
echo "exit $?"; grep -E "RESULT|assert|entry" $O/dbg_chk.log | tail -12 | cut -c1-400
