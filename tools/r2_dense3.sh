#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/d3; rm -rf $O; mkdir -p $O
cd $R
for so in libvbm25_exp_NOATOMIC.so libvbm25_exp_INTATOMIC.so; do
  for ne in 0 1; do
    echo "== $so NE=$ne"
    VBM25_SO=$so VBM25_NE=$ne timeout -s KILL 300 python tools/profile_dense.py 5000000 100000 256 10 100 /tmp/z5.seg 2>&1 | grep -v "waves 0" | head -14
  done
done > $O/exp.log 2>&1
cat $O/exp.log
