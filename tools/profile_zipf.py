#!/usr/bin/env python3
"""Counters of the MaxScore split in scan_range_kernel (libvbm25_prof.so) on a Zipf workload.
usage: profile_zipf.py <n_docs> <vocab> <nq> <nterms> <k>"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vectorchord_bm25_amd as vb
from vectorchord_bm25_amd import _lib

_lib._SO = os.path.join(ROOT, "vectorchord-bm25_amd", "csrc", "libvbm25_prof.so")
_lib._lib = None
from bench import make_queries

n_docs, vocab, nq, nterms, k = (int(x) for x in sys.argv[1:6])
seg = vb.Segment.synth(n_docs, vocab, mean_len=100, len_mode=1, zipf_s=1.0, seed=20260925, threads=16)
gix = vb.GpuIndex(seg)
terms, off = make_queries(seg, vocab, nq, nterms, seed=1, zipf_s=1.0)
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
w = p[:, 1:, :]
tiles = p[:, 1, 0].sum()
print(f"tiles {tiles:.0f}; rows/tile {p[:, 1, 10].sum() / tiles:.1f}; cold blocks {w[:, :, 11].sum():.0f}")
print(f"candidates entering the completion {p[:, :, 1].sum():.0f}; after the block bounds {p[:, :, 2].sum():.0f}; block lookups (decodes) {p[:, :, 14].sum():.0f}")
print(f"cycles per wave: lifetime {p[:, :, 15].mean():.0f}; completion pass 1 {p[:, :, 3].mean():.0f}; pass 2 {p[:, :, 4].mean():.0f}; S2 {p[:, :, 5].mean():.0f}; S3 rows {p[:, :, 7].mean():.0f}; cold incl. completion {p[:, :, 13].mean():.0f}")
