#!/bin/bash
# the whole GPU suite on the final build + the in-kernel phase timers of both scan kernels (profile build)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3last; rm -rf $O; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest_gpu.log
python -c "
import sys; sys.path.insert(0,'.')
import vectorchord_bm25_amd as vb
from bench import WORKLOADS
n_docs, vocab, mean_len, len_mode, zipf_s, nq, nterms, k = WORKLOADS['C3']
vb.Segment.synth(n_docs, vocab, mean_len=mean_len, len_mode=len_mode, zipf_s=zipf_s, seed=20260925, threads=64).save('/tmp/c3.seg')"
timeout 300 python tools/profile_range.py C3 /tmp/c3.seg > $O/c3_phase_timers.txt 2>&1; cat $O/c3_phase_timers.txt
DS_REPS=20 timeout 600 python tools/dense_stress.py > $O/dense_stress.log 2>&1; tail -2 $O/dense_stress.log
python __graft_entry__.py smoke 2>&1 | tail -1
