#!/usr/bin/env python3
"""How much does the reference's own traversal skip?  Blocks the Block-WAND restatement (oracle/, search.rs:149-280) decompresses,
as a fraction of the blocks of the query's terms, for the query shapes of BASELINE.json and of the reference's differential
fuzz (tests/fuzz:217-303: ~100-term queries, top-100).  CPU only.  A shape on which the reference decodes (almost) every block
has nothing for a GPU kernel to prune either: exhaustive evaluation of it is not a missing optimisation.
usage: tools/wand_pruning.py [n_docs]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402
import vectorchord_bm25_amd as vb  # noqa: E402

n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
L = orc.lib()
L.orc_wand_blocks_decoded.restype = C.c_ulonglong
L.orc_wand_blocks_decoded.argtypes = [C.c_int]


def run(name, seg, vocab, nq, nterms, k, zipf):
    oix = orc.OracleIndex.from_arrays(seg.meta(), seg.arrays())
    first = seg.arrays()["term_first_block"].astype(np.int64)
    rng = np.random.default_rng(5)
    if zipf > 0:
        p = 1.0 / np.arange(1, vocab + 1) ** zipf
        p /= p.sum()
    total = dec = 0
    for _ in range(nq):
        if zipf > 0:
            draws = rng.choice(vocab, nterms * 4, p=p)
            _, fi = np.unique(draws, return_index=True)
            toks = draws[np.sort(fi)[:nterms]]
        else:
            toks = rng.choice(vocab, nterms, replace=False)
        ids = seg.token_terms(toks.astype(np.uint32))
        ids = np.sort(ids[ids != 0xffffffff])
        L.orc_wand_blocks_decoded(1)
        oix.search_wand(ids, k)
        dec += L.orc_wand_blocks_decoded(1)
        total += int((first[ids + 1] - first[ids]).sum())
    print(f"{name:58s} blocks of the queries' terms {total:10d}   decompressed by Block-WAND {dec:10d} = {dec / total:.3f}", flush=True)


uni = vb.Segment.synth(n_docs, 30_000, mean_len=100, len_mode=1, zipf_s=0.0, seed=20260925)
run(f"C3 shape: {n_docs} docs / 30k uniform vocab, 5 terms, top-10", uni, 30_000, 64, 5, 10, 0.0)
run("fuzz shape on it: 100 terms, top-100 (tests/fuzz:217-303)", uni, 30_000, 16, 100, 100, 0.0)
run("  ... 32 terms, top-100", uni, 30_000, 16, 32, 100, 0.0)
zipf = vb.Segment.synth(n_docs, 100_000, mean_len=100, len_mode=1, zipf_s=1.0, seed=20260925)
run(f"C5 shape: {n_docs} docs / 100k Zipf(1) vocab, 10 terms, top-100", zipf, 100_000, 32, 10, 100, 1.0)
run("fuzz shape on it: 100 terms, top-100", zipf, 100_000, 8, 100, 100, 1.0)
run("  ... 32 terms, top-100", zipf, 100_000, 8, 32, 100, 1.0)
