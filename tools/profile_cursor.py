#!/usr/bin/env python3
"""Phase timers of scan_cursor_kernel (libvbm25_prof.so, built with -DVBM25_PROFILE)."""
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
from bench import WORKLOADS, make_queries

wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
n_docs, vocab, mean_len, len_mode, zipf_s, nq, nterms, k = WORKLOADS[wl]
seg = vb.Segment.synth(n_docs, vocab, mean_len=mean_len, len_mode=len_mode, zipf_s=zipf_s, threads=16)
gix = vb.GpuIndex(seg)
terms, off = make_queries(seg, vocab, nq, nterms, seed=1, zipf_s=zipf_s)
b = vb.Batch(gix, nq, len(terms), k)
b.set_queries(terms, off)
b.run()
b.fetch()
L = vb.lib()
GRID = 256 * 24
nrec = (16 * GRID + 32) // 33
out = np.zeros(33 * nrec, dtype=np.uint64)
L.vbm25_batch_profile.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
b.run()
assert L.vbm25_batch_profile(b.h, out.ctypes.data_as(C.c_void_p), nrec) == 0
p = out[: 16 * GRID].reshape(GRID, 16).astype(np.float64)
live = p[:, 6] > 0
steps = p[:, 0].sum()
items = p[:, 6].sum()
print(f"waves with work {int(live.sum())} of {GRID}; items {int(items)}; blocks {int(steps)}; blocks/item {steps / items:.1f}")
print(f"wave lifetime cycles: mean {p[live, 7].mean():.0f} max {p[live, 7].max():.0f} min {p[live, 7].min():.0f}")
print(f"pending appended/block {p[:, 1].sum() / steps:.3f}; resolve calls/block {p[:, 2].sum() / steps:.3f}; resolved docs/block {p[:, 3].sum() / steps:.3f}")
print(f"cold passes/block {p[:, 5].sum() / steps:.4f}; blocks staged while not hot {p[:, 13].sum() / steps:.4f}")
names = {8: "setup (per item)", 9: "select+fields+advance", 10: "evict (resolve, cold)", 11: "ids+stage+mark", 12: "pending append", 14: "poll+wipe"}
tot = 0.0
for i, n in names.items():
    d = items if i == 8 else steps
    print(f"  {n:30s} {p[:, i].sum() / d:9.0f} cycles/{'item' if i == 8 else 'block'}")
    if i != 8:
        tot += p[:, i].sum() / steps
print(f"  sum per block {tot:.0f}; resolve cycles/block {p[:, 4].sum() / steps:.0f} (per call {p[:, 4].sum() / max(1, p[:, 2].sum()):.0f}); cold cycles/block {p[:, 15].sum() / steps:.0f} (per pass {p[:, 15].sum() / max(1, p[:, 5].sum()):.0f})")
print(f"  busy cycles per wave (sum of phases) {(p[:, 8:15].sum()) / live.sum():.0f}")
