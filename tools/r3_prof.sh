#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3prof; rm -rf $O; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest_gpu.log
python -c "
import sys; sys.path.insert(0,'.')
import vectorchord_bm25_amd as vb
from bench import WORKLOADS
n_docs, vocab, mean_len, len_mode, zipf_s, nq, nterms, k = WORKLOADS['C3']
vb.Segment.synth(n_docs, vocab, mean_len=mean_len, len_mode=len_mode, zipf_s=zipf_s, seed=20260925, threads=64).save('/tmp/c3.seg')"
timeout 300 python tools/profile_range.py C3 /tmp/c3.seg > $O/phases.txt 2>&1; cat $O/phases.txt
