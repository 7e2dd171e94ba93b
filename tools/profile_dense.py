#!/usr/bin/env python3
"""Phase timers of scan_dense_kernel (libvbm25_prof.so, built with -DVBM25_PROFILE).
usage: profile_dense.py C5 | <n_docs> <vocab> <nq> <nterms> <k> [cache]"""
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
from bench import WORKLOADS, make_queries, usable_cpus

if sys.argv[1] in WORKLOADS:
    n_docs, vocab, _, _, _, nq, nterms, k = WORKLOADS[sys.argv[1]]
    cache = "device"
else:
    n_docs, vocab, nq, nterms, k = (int(x) for x in sys.argv[1:6])
    cache = sys.argv[6] if len(sys.argv) > 6 else ""
if cache == "device":
    seg = vb.DeviceSegment.synth(n_docs, vocab, mean_len=100, len_mode=1, zipf_s=1.0, seed=20260925, device=0)
elif cache and os.path.exists(cache):
    seg = vb.Segment.load(cache)
else:
    seg = vb.Segment.synth(n_docs, vocab, mean_len=100, len_mode=1, zipf_s=1.0, seed=20260925, threads=usable_cpus())
    if cache:
        seg.save(cache)
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
b.set_timing(True)
b.run()
assert L.vbm25_batch_profile(b.h, out.ctypes.data_as(C.c_void_p), nrec) == 0
print("scan kernels ms", b.kernel_ms())
p = out[: 16 * GRID * NW].reshape(GRID, NW, 16).astype(np.float64)
w0 = p[:, 0, :]
win = w0[:, 0].sum()
print(f"workgroups with work {int((w0[:, 12] > 0).sum())}; items {int(w0[:, 12].sum())}; windows {int(win)} ({win / max(1, w0[:, 12].sum()):.1f} per item); "
      f"candidates buffered {int(p[:, :, 10].sum())} ({p[:, :, 10].sum() / max(1, win):.2f} per window), re-scored exactly {int(p[:, :, 11].sum())}")
print(f"tasks fetched {int(p[:, :, 13].sum())} skipped {int(p[:, :, 14].sum())}; per window {p[:, :, 13].sum() / max(1, win):.1f} + {p[:, :, 14].sum() / max(1, win):.1f}")
print(f"wave lifetime cycles mean {p[:, :, 15].mean():.0f} max {p[:, :, 15].max():.0f} min {p[:, :, 15].min():.0f}; in window loops mean {p[:, :, 9].mean():.0f}; item setup {p[:, :, 8].sum() / max(1, p[:, :, 12].sum()):.0f} per item")
if os.environ.get("PD_SUB"):
    print(f"run_group per window per wave: entry read {p[:, :, 10].sum() / win / 8:.0f}  fetch wait {p[:, :, 11].sum() / win / 8:.0f}  accumulate {p[:, :, 14].sum() / win / 8:.0f}")
names = {1: "P0 enumerate + task list (1 barrier)", 2: "P1 essential tasks", 3: "P2 non-essential phases", 4: "barrier before P3", 5: "P3 scan, wipe (no barrier)",
         6: "flush: filter", 7: "flush: exact re-scoring"}
for w in ("all", 0, 7):
    sel = p if w == "all" else p[:, w:w + 1, :]
    t = win * sel.shape[1]
    print(f"-- waves {w}: cycles per window (share of the window loops)")
    tot = sel[:, :, 9].sum()
    for i, n in names.items():
        print(f"   {n:40s} {sel[:, :, i].sum() / t:9.0f}  {100 * sel[:, :, i].sum() / tot:5.1f} %")
