#!/bin/bash
# one development iteration on the GPU: the search tests, phase timers, verified bench line
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3iter; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_search.py -x -q -m gpu -k "not c5_full and not bench_distributed" > $O/pytest_search.log 2>&1; tail -4 $O/pytest_search.log
python -c "
import sys; sys.path.insert(0,'.')
import vectorchord_bm25_amd as vb
from bench import WORKLOADS
n_docs, vocab, mean_len, len_mode, zipf_s, nq, nterms, k = WORKLOADS['C3']
vb.Segment.synth(n_docs, vocab, mean_len=mean_len, len_mode=len_mode, zipf_s=zipf_s, seed=20260925, threads=64).save('/tmp/c3.seg')"
timeout 300 python tools/profile_range.py C3 /tmp/c3.seg > $O/phases.txt 2>&1; cat $O/phases.txt
timeout 300 python bench.py --no-cpu-baseline --verify --cache /tmp/c3.seg --steps 100 > $O/bench_c3.json 2> $O/bench_c3.err; python -c "
import json; d=json.load(open('$O/bench_c3.json')); print(d['value'], d['ms_per_step'], d['roofline'], d['config'].get('verified_bit_exact_vs_oracle'))" || tail -5 $O/bench_c3.err
