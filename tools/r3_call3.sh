#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c3; rm -rf $O; mkdir -p $O
cd $R
V=$R/build/variants/oexp/vectorchord-bm25_amd/csrc
for name in nosv lb2 base; do
  VBM25_LIBRARY=$V/libvbm25_$name.so DS_REPS=10 DS_ITEMS=1024,4096 DS_K=200,256 timeout 200 python tools/dense_stress.py > $O/stress_o_$name.log 2>&1
  echo "== old+$name exit $?"; grep -E "^items|RESULT|threshold|DIFFERS|q[0-9]:|field" $O/stress_o_$name.log | head -14
done
CS=$R/vectorchord-bm25_amd/csrc
for name in d256; do
  VBM25_LIBRARY=$CS/libvbm25_$name.so DS_REPS=30 DS_ITEMS=256,1024,4096,16384 DS_K=10,200,256 timeout 300 python tools/dense_stress.py > $O/stress_$name.log 2>&1
  echo "== HEAD $name exit $?"; grep -E "^items|RESULT|threshold|assert|DIFFERS|q[0-9]:|field" $O/stress_$name.log | head -30
done
timeout 120 tools/ubench/valu_rates > $O/valu_rates.txt 2>&1; grep -E "SIMD 4" $O/valu_rates.txt
