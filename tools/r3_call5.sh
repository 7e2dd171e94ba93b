#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c5; rm -rf $O; mkdir -p $O
cd $R
RD_REPS=8 timeout 300 python tools/range_debug.py > $O/dbg_2m.log 2>&1; tail -30 $O/dbg_2m.log
RD_REPS=6 RD_NQ=64 timeout 300 python tools/range_debug.py > $O/dbg_2m_64.log 2>&1; tail -12 $O/dbg_2m_64.log
