#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/d6; rm -rf $O; mkdir -p $O
cd $R
for so in libvbm25.so libvbm25_un6.so; do
  echo "== $so"
  VBM25_SO=$so DENSE_NO_OLD=1 DENSE_SAMPLE=4 DENSE_VARIANTS="VBM25_DENSE_ITEMS=2048;VBM25_DENSE_ITEMS=8192;VBM25_DENSE_ITEMS=16384" timeout -s KILL 400 python tools/dense_check.py 10000000 100000 512 10 100 /tmp/z10.seg 2>&1 | grep -v "same records\|oracle brute"
done > $O/sweep.log 2>&1
cat $O/sweep.log
