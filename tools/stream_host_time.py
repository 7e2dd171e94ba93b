#!/usr/bin/env python3
"""Where a pipelined step's time goes on the host: wall time inside vbm25_stream_submit and vbm25_stream_collect per batch against
the step itself (C3's shape, depth 3).   usage: stream_host_time.py [depth] [batches]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vectorchord_bm25_amd as vb
from bench import WORKLOADS, make_queries

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
n_docs, vocab, mean_len, len_mode, zipf_s, nq, nterms, k = WORKLOADS["C3"]
seg = vb.DeviceSegment.synth(n_docs, vocab, mean_len=mean_len, len_mode=len_mode, zipf_s=zipf_s, seed=20260925, device=0)
gix = vb.GpuIndex(seg)
shards = [make_queries(seg, vocab, nq, nterms, seed=1 + i, zipf_s=zipf_s) for i in range(4)]
st = vb.Stream(gix, depth, nq, max(len(t) for t, _ in shards), k)
outs = [(np.zeros((nq, k), dtype=vb.HIT_DTYPE), np.zeros(nq, dtype=np.uint32)) for _ in range(depth)]
for rep in range(2):
    ts = tc = 0.0
    t0 = time.perf_counter()
    for i in range(n):
        if st.in_flight == depth:
            a = time.perf_counter()
            st.collect(outs[i % depth])
            tc += time.perf_counter() - a
        a = time.perf_counter()
        st.submit(*shards[i % 4])
        ts += time.perf_counter() - a
    while st.in_flight:
        st.collect(outs[0])
    wall = time.perf_counter() - t0
    print(f"depth {depth}, {n} batches: step {1e6 * wall / n:.1f} us = {n * nq / wall / 1e6:.3f} M q/s; inside submit {1e6 * ts / n:.1f} us, inside collect "
          f"{1e6 * tc / n:.1f} us per batch (collect includes waiting for the device)")
