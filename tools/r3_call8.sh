#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c8; rm -rf $O; mkdir -p $O
cd $R
for v in nodma chk ""; do
  lib=$R/vectorchord-bm25_amd/csrc/libvbm25${v:+_$v}.so
  VBM25_LIBRARY=$lib RD_REPS=12 timeout 300 python tools/range_debug.py > $O/dbg_$v.log 2>&1
  echo "== [$v] exit $?"; grep -E "differ|RESULT|rep .* q|assert|got|fault" $O/dbg_$v.log | grep -v " 0 of" | tail -8 | cut -c1-330
done
