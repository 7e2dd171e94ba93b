#!/usr/bin/env python3
"""Phase timers of scan_range_kernel (csrc/libvbm25_prof.so, built with -DVBM25_PROFILE): cycles per tile and wave.
usage: tools/profile_range.py [workload] [segment cache file]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("VBM25_LIBRARY", os.path.join(ROOT, "vectorchord-bm25_amd", "csrc", "libvbm25_prof.so"))
import vectorchord_bm25_amd as vb
from bench import WORKLOADS, make_queries

wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
cache = sys.argv[2] if len(sys.argv) > 2 else ""
n_docs, vocab, mean_len, len_mode, zipf_s, nq, nterms, k = WORKLOADS[wl]
if cache and os.path.exists(cache):
    seg = vb.Segment.load(cache)
else:
    seg = vb.Segment.synth(n_docs, vocab, mean_len=mean_len, len_mode=len_mode, zipf_s=zipf_s, seed=20260925, threads=16)
    if cache:
        seg.save(cache)
gix = vb.GpuIndex(seg)
terms, off = make_queries(seg, vocab, nq, nterms, seed=1, zipf_s=zipf_s)
b = vb.Batch(gix, nq, len(terms), k)
b.set_queries(terms, off)
b.run()
b.fetch()
L = vb.lib()
GRID, NW = 512, 8
nrec = (16 * GRID * NW + 32) // 33
out = np.zeros(33 * nrec, dtype=np.uint64)
L.vbm25_batch_profile.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
b.run()
assert L.vbm25_batch_profile(b.h, out.ctypes.data_as(C.c_void_p), nrec) == 0
p = out[: 16 * GRID * NW].reshape(GRID, NW, 16).astype(np.float64)
tiles = p[:, 1, 0].sum()  # tiles processed (counted by worker 1 of each workgroup)
items = p[:, 0, 12].sum()
print(f"{wl}: {int(items)} items, {tiles / items:.1f} tiles per item, rows per tile {p[:, 0, 10].sum() / tiles:.1f}, hits scored per tile "
      f"{p[:, :, 13].sum() / tiles:.1f}, cold blocks per tile {p[:, :, 11].sum() / tiles:.2f}, pool shrinks per item {p[:, 0, 14].sum() / items:.2f}")
print(f"wave lifetime cycles mean {p[:, :, 15].mean():.0f} max {p[:, :, 15].max():.0f}; tile loops {p[:, :, 9].mean():.0f}; item setup {p[:, 0, 8].sum() / items:.0f} per item")
print(f"first tile's tail (bootstrap) {p[:, 1:, 8].sum() / (7 * items):.0f} cycles per item and worker; tails with barriers {p[:, 1, 10].sum() / items:.2f} per item")
names = [(1, "S1a decode (registers)"), (4, "wait at barrier B (previous tile)"), (5, "previous tile's tail: pool, cold pass"),
         (6, "S1b stage, mark, events"), (7, "control wave: poll + plan + hits"), (2, "wait at barrier A"), (3, "S2 rows x terms (+ S3 on the control wave)")]
for label, sel in (("workers (mean of 7)", p[:, 1:, :]), ("control wave", p[:, 0:1, :])):
    print(f"-- {label}: cycles per tile")
    tot = 0.0
    for i, n in names:
        v = sel[:, :, i].sum() / (tiles * sel.shape[1])
        tot += v
        print(f"   {n:36s} {v:8.0f}")
    print(f"   {'sum':36s} {tot:8.0f}   (tile loop per tile: {sel[:, :, 9].sum() / (tiles * sel.shape[1]):.0f})")
