#!/bin/bash
# dense kernel A/B of library builds on C5: VARIANTS = paths relative to the repository
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3dense_ab; rm -rf $O; mkdir -p $O
cd $R
python -c "
import sys; sys.path.insert(0,'.')
import vectorchord_bm25_amd as vb
from bench import WORKLOADS
n_docs, vocab, mean_len, len_mode, zipf_s, nq, nterms, k = WORKLOADS['C5']
vb.Segment.synth(n_docs, vocab, mean_len=mean_len, len_mode=len_mode, zipf_s=zipf_s, seed=20260925, threads=64).save('/tmp/c5.seg')"
for v in $VARIANTS; do
  n=$(basename $v .so)
  VBM25_LIBRARY=$R/$v timeout 600 python bench.py --workload C5 --no-cpu-baseline --cache /tmp/c5.seg --steps 8 --warmup 2 --batches 2 > $O/$n.json 2> $O/$n.err
  python -c "
import json; d=json.load(open('$O/$n.json')); print('$n', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])" || tail -5 $O/$n.err
done
