#!/usr/bin/env python3
"""Phase timers of scan_win_kernel (libvbm25_prof.so, built with -DVBM25_PROFILE): cycles per window and wave."""
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
for kv in (sys.argv[2].split(",") if len(sys.argv) > 2 and sys.argv[2] else []):
    n, v = kv.split("=")
    vb.set_tuning(n, int(v))
n_docs, vocab, mean_len, len_mode, zipf_s, nq, nterms, k = WORKLOADS[wl]
seg = vb.DeviceSegment.synth(n_docs, vocab, mean_len=mean_len, len_mode=len_mode, zipf_s=zipf_s, seed=20260925, device=0)
gix = vb.GpuIndex(seg)
terms, off = make_queries(seg, vocab, nq, nterms, seed=1, zipf_s=zipf_s)
b = vb.Batch(gix, nq, len(terms), k)
b.set_queries(terms, off)
assert b.debug_route() == 3, b.debug_route()
b.run()
b.fetch()
L = vb.lib()
WAVES = 768 * 4
out = np.zeros(16 * WAVES, dtype=np.uint64)
L.vbm25_batch_profile.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
b.run()
assert L.vbm25_batch_profile(b.h, out.ctypes.data_as(C.c_void_p), 384) == 0  # (384 workgroups of 8 waves = 3072 waves)
p = out.reshape(WAVES, 16).astype(np.float64)
win = p[:, 0].sum()
print(f"windows per wave {p[:, 0].mean():.1f}; items per wave {p[:, 12].mean():.2f}; second arrivals per window {p[:, 10].sum() / win:.2f}")
print(f"wave lifetime cycles mean {p[:, 15].mean():.0f} max {p[:, 15].max():.0f} min {p[:, 15].min():.0f}; in window loops {p[:, 9].mean():.0f}; "
      f"setup per item {p[:, 8].sum() / p[:, 12].sum():.0f}; item end (drain, cold pass, result) {p[:, 13].sum() / p[:, 12].sum():.0f}")
names = {1: "wait for the window's runs", 2: "marks phase 1 (stage, the group's atomics issued)", 3: "marks phase 2 (returned words -> second arrivals)",
         4: "wipe", 5: "wait for the last window's tf/fn words", 6: "C2 of the last window's open passes", 7: "C1 of this window (+ passes beyond two)",
         11: "threshold wait, loads issued"}
for i, n in names.items():
    print(f"   {n:44s} {p[:, i].sum() / win:8.0f}")
# where the spread of the waves' lifetimes comes from: between workgroups (CUs) or inside them; work (second arrivals) or place
W = 12
wg = p[: (WAVES // W) * W].reshape(-1, W, 16)
life = wg[:, :, 15]
print(f"lifetime: workgroup means min {life.mean(1).min():.0f} max {life.mean(1).max():.0f}; spread inside a workgroup (max - min) mean {np.mean(life.max(1) - life.min(1)):.0f}")
print(f"correlation of a wave's lifetime with its second arrivals {np.corrcoef(p[:, 15], p[:, 10])[0, 1]:.2f}, with its time in the window loops {np.corrcoef(p[:, 15], p[:, 9])[0, 1]:.2f}")
slot = life.mean(0)
print("mean lifetime by wave slot of the workgroup:", " ".join(f"{x:.0f}" for x in slot))
xcd = life.mean(1).reshape(-1, 8).mean(0) if life.shape[0] % 8 == 0 else None
print("mean lifetime by workgroup number mod 8 (XCD):", None if xcd is None else " ".join(f"{x:.0f}" for x in xcd))
