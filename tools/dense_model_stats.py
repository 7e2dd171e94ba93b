"""Pruning statistics of the dense-window scheme (oracle/dense_model.inc) on a Zipf corpus, on the CPU:
how many blocks the MaxScore split + block-level skip avoid and how many candidates get re-scored.
usage: python tools/dense_model_stats.py [n_docs] [vocab] [nq] [nterms] [k] [wmax] [phases]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc
import vectorchord_bm25_amd as vb
from bench import make_queries

n_docs, vocab, nq, nterms, k, wmax, phases = (int(a) for a in (sys.argv[1:] + [2_000_000, 100_000, 16, 10, 100, 16384, 1][len(sys.argv) - 1:]))
seg = vb.Segment.synth(n_docs, vocab, mean_len=100, len_mode=1, zipf_s=1.0, seed=3)
oix = orc.OracleIndex.from_arrays(seg.meta(), seg.arrays())
terms, off = make_queries(seg, vocab, nq, nterms, seed=1, zipf_s=1.0)
tot = {}
for q in range(nq):
    t = terms[off[q]:off[q + 1]]
    ref = oix.search_brute(t, k)
    t0 = time.time()
    got, st = oix.dense_model(t, k, wmax=wmax, phases=phases)
    dt = time.time() - t0
    ok = got.tobytes() == ref.tobytes()
    for key, v in st.items():
        tot[key] = tot.get(key, 0) + v
    print(f"q{q}: {'OK ' if ok else 'DIFF'} {st} theta={got['score'][-1]:.3f} top={got['score'][0]:.3f} {dt:.1f}s", flush=True)
    assert ok
print("total", tot)
print(f"blocks fetched / enumerated: {(tot['fetched_untested'] + tot['tested'] - tot['skipped']) / tot['blocks']:.3f}; "
      f"candidates per window {tot['candidates'] / tot['windows']:.3f}, re-scored exactly {tot['rescored'] / max(1, tot['candidates']):.3f} of them; "
      f"phases per window {tot['phases'] / tot['windows']:.2f}")
