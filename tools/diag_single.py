import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vectorchord_bm25_amd as vb
seg = vb.Segment.synth(2_000_000, 30000, mean_len=100, len_mode=1, seed=5)
gix = vb.GpuIndex(seg)
rng = np.random.default_rng(2)
nq = 512
toks = np.stack([rng.choice(30000, 5, replace=False) for _ in range(nq)]).astype(np.uint32)
t = np.sort(seg.token_terms(toks.reshape(-1)).reshape(nq, 5), axis=1).reshape(-1)
off = (np.arange(nq + 1) * 5).astype(np.uint32)
order = sys.argv[1] if len(sys.argv) > 1 else "batch_first"
if order == "batch_first":
    h1, n1 = vb.search_batch(gix, t, off, 10)
for rep in range(3):
    for q in (3, 77, 500):
        hq, nq1 = vb.search_batch(gix, t[off[q]:off[q + 1]], np.array([0, 5], dtype=np.uint32), 10)
        print(order, rep, q, "n", nq1, "first", hq[0][:2], flush=True)
if order != "batch_first":
    h1, n1 = vb.search_batch(gix, t, off, 10)
print("batch q3", h1[3][:2], n1[3])
