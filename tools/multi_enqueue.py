#!/usr/bin/env python3
"""Host enqueue cost of the C-ABI multi-GPU route (vbm25_multi_*), measured on ONE GPU with N replicas of the C3 index on device 0:
how long does the single host thread need to enqueue a step on N devices (vbm25_multi_batch_run: per device hipSetDevice, the scan's
launches, the download of the records), against the step itself?  On an N-GPU node the devices run concurrently, so the host's
enqueue time per step is the floor of the step; the prediction printed at the end is max(enqueue, one device's step)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vectorchord_bm25_amd as vb
from bench import make_queries

n_rep = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_docs = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
seg = vb.DeviceSegment.synth(n_docs, 30_000, mean_len=100, len_mode=1, seed=20260925, device=0).download()
multi = vb.MultiIndex(seg, [0] * n_rep)
nq = 1024 * n_rep
terms, off = make_queries(seg, 30_000, nq, 5, seed=1, zipf_s=0.0)
mb = vb.MultiBatch(multi, nq, len(terms), 10)
mb.set_queries(terms, off)
for _ in range(3):
    mb.run()
    mb.fetch()
steps = 30
t_run, t_all, t_set = [], [], []
for _ in range(steps):
    t0 = time.perf_counter()
    mb.set_queries(terms, off)
    t1 = time.perf_counter()
    mb.run()
    t2 = time.perf_counter()
    mb.fetch()
    t3 = time.perf_counter()
    t_set.append(t1 - t0)
    t_run.append(t2 - t1)
    t_all.append(t3 - t0)
# one replica alone: the step a device needs for its shard
one = vb.MultiIndex(seg, [0])
mb1 = vb.MultiBatch(one, 1024, int(off[1024]), 10)
mb1.set_queries(terms[:off[1024]], off[:1025])
for _ in range(3):
    mb1.run()
    mb1.fetch()
t_one = []
for _ in range(steps):
    t0 = time.perf_counter()
    mb1.run()
    mb1.fetch()
    t_one.append(time.perf_counter() - t0)
med = lambda v: 1e3 * float(np.median(v))
print(f"{n_rep} replicas on device 0, {nq} queries per step ({n_docs} documents, C3's query shape)")
print(f"  vbm25_multi_batch_set_queries (host: shards staged in pinned memory, uploads enqueued)  {med(t_set):8.3f} ms = {1e3 * med(t_set) / n_rep:6.1f} us per device")
print(f"  vbm25_multi_batch_run (host: enqueue of every device's scan and download)                {med(t_run):8.3f} ms = {1e3 * med(t_run) / n_rep:6.1f} us per device")
print(f"  whole step here (the replicas share ONE GPU: their scans run one after the other)        {med(t_all):8.3f} ms")
print(f"  one replica alone, run + fetch of its 1024 queries                                       {med(t_one):8.3f} ms")
floor = med(t_set) + med(t_run)
print(f"  predicted step on {n_rep} GPUs: max(host enqueue {floor:.3f} ms, one device {med(t_one):.3f} ms) = {max(floor, med(t_one)):.3f} ms "
      f"-> {nq / max(floor, med(t_one)) / 1e3:.2f} M queries/s; the host's share of it {100 * floor / max(floor, med(t_one)):.0f} %")
