#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3d2; rm -rf $O; mkdir -p $O
cd $R
VBM25_DEBUG=1 timeout 300 python - > $O/fail.log 2>&1 <<'PY'
import os, sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import vectorchord_bm25_amd as vb
from bench import WORKLOADS, make_queries
n_docs, vocab, mean_len, len_mode, zipf_s, nq, nterms, k = WORKLOADS["C3"]
seg = vb.Segment.synth(n_docs, vocab, mean_len=mean_len, len_mode=len_mode, zipf_s=zipf_s, seed=20260925, threads=16)
gix = vb.GpuIndex(seg)
terms, off = make_queries(seg, vocab, nq, nterms, seed=1, zipf_s=zipf_s)
b = vb.Batch(gix, nq, len(terms), k)
b.set_queries(terms, off)
import time
for rep in range(6):
    t0 = time.perf_counter(); b.run(); h, n = b.fetch(); dt = time.perf_counter() - t0
    print("rep", rep, "ms", round(dt * 1e3, 3), "items/failed", b.debug_counts(), flush=True)
PY
tail -30 $O/fail.log
