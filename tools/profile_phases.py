#!/usr/bin/env python3
"""Phase timers of scan_kernel, the tile kernel (libvbm25_prof.so, built with -DVBM25_PROFILE); run with
VBM25_NO_CURSOR=1, otherwise the cursor kernel serves these queries (tools/profile_cursor.py).
Prints average cycles per tile of one worker wave and of the planner wave."""
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
sys.path.insert(0, ROOT)
from bench import WORKLOADS, make_queries

wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
cache = sys.argv[2] if len(sys.argv) > 2 else ""
n_docs, vocab, mean_len, len_mode, zipf_s, nq, nterms, k = WORKLOADS[wl]
if cache and os.path.exists(cache):
    seg = vb.Segment.load(cache)
else:
    seg = vb.Segment.synth(n_docs, vocab, mean_len=mean_len, len_mode=len_mode, zipf_s=zipf_s, threads=16)
    if cache:
        seg.save(cache)
gix = vb.GpuIndex(seg)
terms, off = make_queries(seg, vocab, nq, nterms, seed=1, zipf_s=zipf_s)
b = vb.Batch(gix, nq, len(terms), k)
b.set_queries(terms, off)
b.run()
b.fetch()
L = vb.lib()
print("occupancy API: workgroups per CU =", L.vbm25_scan_occupancy())
NWG = 1536
out = np.zeros((NWG, 33), dtype=np.uint64)
L.vbm25_batch_profile.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
b.run()
assert L.vbm25_batch_profile(b.h, out.ctypes.data_as(C.c_void_p), NWG) == 0
w, p, tot = out[:, :16].astype(np.float64), out[:, 16:32].astype(np.float64), out[:, 32].astype(np.float64)
tiles = p[:, 7].sum()
print(f"workgroups {NWG}, tiles {int(tiles)}, tiles/wg {tiles / NWG:.1f}, total cycles/wg {tot.mean():.0f}")
print(f"entries per tile (worker wave 0's view of nent): {w[:, 7].sum() / tiles:.2f}")
names_w = ["A decode+mark", "wait X", "issue loads", "B fast/slow", "wait Y", "C heads+wipe", "wait Z(+W)"]
for i, n in enumerate(names_w):
    print(f"worker  {n:12s} {w[:, i].sum() / tiles:9.0f} cycles/tile")
names_p = ["setup+plan0 (per item)", "plan_start", "wait X", "plan_finish", "wait Y+Z", "merge(+W)", "  slow postings (count, not cycles)"]
for i, n in enumerate(names_p):
    d = NWG if i == 0 else tiles
    print(f"planner {n:22s} {p[:, i].sum() / d:9.0f} cycles/{'item' if i == 0 else 'tile'}")
for i, n in [(8, "join: load list"), (9, "join: all-pairs"), (10, "join: >=3 addends"), (11, "join total (Y -> cand merge)"), (12, "cand merge"), (13, "publish"), (14, "candidates (count)")]:
    print(f"planner   {n:30s} {p[:, i].sum() / tiles:9.1f} /tile")
print("failed items:", int(p[:, 6].sum()), "| hot tiles:", int(p[:, 13].sum()), "of", int(tiles))
print("slow list: max", int(p[:, 8].max()), "mean/tile", p[:, 9].sum() / tiles, "tiles > 64:", int(p[:, 10].sum()), "| cand mean/tile", p[:, 11].sum() / tiles, "max", int(p[:, 12].max()))
print(f"cycles per tile overall: {tot.sum() / tiles:.0f}")
