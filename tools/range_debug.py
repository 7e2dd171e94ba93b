#!/usr/bin/env python3
"""Repeat one batch through the range kernel and report every query whose hits differ from the oracle's brute force:
which documents are missing / extra, how many of the query's lists hold them.  RD_DOCS, RD_NQ, RD_K, RD_REPS, RD_TERMS."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc, vectorchord_bm25_amd as vb
n_docs = int(os.environ.get("RD_DOCS", "2000000")); nq = int(os.environ.get("RD_NQ", "512")); k = int(os.environ.get("RD_K", "10"))
reps = int(os.environ.get("RD_REPS", "10")); nt = int(os.environ.get("RD_TERMS", "5")); vocab = int(os.environ.get("RD_VOCAB", "30000"))
seg = vb.Segment.synth(n_docs, vocab, mean_len=100, len_mode=1, seed=5)
gix = vb.GpuIndex(seg)
oix = orc.OracleIndex.from_arrays(seg.meta(), seg.arrays())
rng = np.random.default_rng(2)
toks = np.stack([rng.choice(vocab, nt, replace=False) for _ in range(nq)]).astype(np.uint32)
t = np.sort(seg.token_terms(toks.reshape(-1)).reshape(nq, nt), axis=1).reshape(-1)
off = (np.arange(nq + 1) * nt).astype(np.uint32)
ref, nref, _ = oix.search_batch(t, off, k, mode="brute", threads=8)
b = vb.Batch(gix, nq, len(t), k)
b.set_queries(t, off)
bad_runs = 0
for rep in range(reps):
    b.run()
    hits, nh = b.fetch()
    items, failed = b.debug_counts()
    chk = b.debug_check()
    if chk and chk[0]:
        import struct
        def dbl(lo, hi): return struct.unpack("<d", struct.pack("<II", lo, hi))[0]
        print(f" rep {rep}: device assertion code {chk[0]} value {chk[1]} (0x{chk[1]:x}) item {chk[2]} [3]={chk[3]} rest {[hex(x) for x in chk[4:]]}", flush=True)
        if chk[0] == 36:
            print(f"   entry {chk[1]} of {chk[3]}: score {dbl(chk[4], chk[5])} doc {chk[6]} tile {chk[7]}; before: {dbl(chk[8], chk[9])} doc {chk[10]}; after: {dbl(chk[11], chk[12])} doc {chk[13]}; hscale {dbl(chk[14], chk[15])}", flush=True)
    th = b.debug_theta(nq)
    nbad = 0
    for q in range(nq):
        same = nh[q] == nref[q] and np.array_equal(hits["doc_id"][q, :nh[q]], ref["doc_id"][q, :nref[q]]) and \
            np.array_equal(hits["score"][q, :nh[q]].view(np.uint64), ref["score"][q, :nref[q]].view(np.uint64))
        if same:
            continue
        nbad += 1
        if nbad <= 3:
            got = set(int(x) for x in hits["doc_id"][q, :nh[q]]); want = set(int(x) for x in ref["doc_id"][q, :nref[q]])
            miss = sorted(want - got); extra = sorted(got - want)
            sc = {int(r["doc_id"]): float(r["score"]) for r in ref[q, :nref[q]]}
            print("   got:", [(int(h["doc_id"]), round(float(h["score"]), 4)) for h in hits[q, :nh[q]]])
            print(f" rep {rep} q{q}: n_hits {nh[q]}/{nref[q]} missing {[(d, round(sc[d], 4), int(np.where(ref['doc_id'][q]==d)[0][0])) for d in miss[:4]]} extra {extra[:4]} "
                  f"theta {th[q]:.6g} oracle k-th {float(ref['score'][q, nref[q]-1]):.6g}", flush=True)
    bad_runs += nbad != 0
    print(f"rep {rep}: {nbad} of {nq} queries differ; items {items} failed {failed}", flush=True)
print("RESULT", "clean" if bad_runs == 0 else f"{bad_runs} bad runs")
