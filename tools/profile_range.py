#!/usr/bin/env python3
"""Phase timers of scan_range_kernel (libvbm25_prof.so, built with -DVBM25_PROFILE)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vectorchord_bm25_amd as vb
from vectorchord_bm25_amd import _lib

_lib._SO = os.environ.get("VBM25_LIBRARY") or os.path.join(ROOT, "vectorchord-bm25_amd", "csrc", "libvbm25_prof.so")
_lib._lib = None
from bench import WORKLOADS, make_queries

wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
cache = sys.argv[2] if len(sys.argv) > 2 else ""
n_docs, vocab, mean_len, len_mode, zipf_s, nq, nterms, k = WORKLOADS[wl]
if cache and os.path.exists(cache):
    seg = vb.Segment.load(cache)
else:
    seg = vb.DeviceSegment.synth(n_docs, vocab, mean_len=mean_len, len_mode=len_mode, zipf_s=zipf_s, seed=20260925, device=0)
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
tiles = p[:, 1:, 0]
print(f"tiles per wave mean {tiles.mean():.1f}; items per WG {p[:, 0, 12].mean():.2f}; rows/tile {p[:, 1, 10].sum() / max(1, tiles[:, 0].sum()):.1f}; "
      f"cold blocks/tile {p[:, :, 11].sum() / max(1, tiles[:, 0].sum()):.2f}")
print(f"wave lifetime cycles mean {p[:, :, 15].mean():.0f} max {p[:, :, 15].max():.0f}; in tile loops {p[:, :, 9].mean():.0f}; setup/item {p[:, :, 8].sum() / max(1, p[:, :, 12].sum()):.0f}")
names = {1: "S1 (workers) / plan (wave 0)", 2: "barrier A", 5: "S2 (wipe, rows x terms)", 6: "barrier B", 7: "S3 rows", 13: "S3 cold"}
for w in ("workers", 0, 1):
    sel = p[:, 1:, :] if w == "workers" else p[:, w:w + 1, :]
    t = p[:, 1:2, 0].sum() * sel.shape[1]
    print(f"-- waves {w}: cycles per tile")
    for i, n in names.items():
        print(f"   {n:32s} {sel[:, :, i].sum() / t:8.0f}")
