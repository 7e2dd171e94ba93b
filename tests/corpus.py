"""Small synthetic corpora for tests (numpy, doc-major; mirrors tests/fuzz:168-205 of the
reference: each document is L i.i.d. token draws, tf = multiplicity, length = sum tf)."""
import numpy as np


def token_keys(vocab):
    """intern() short path (vector.rs:21-24): ASCII decimal, zero padded to 16 bytes.
    Returns (keys sorted bytewise [V,16], rank_of_token[V])."""
    raw = np.zeros((vocab, 16), dtype=np.uint8)
    for t in range(vocab):
        s = str(t).encode()
        raw[t, :len(s)] = np.frombuffer(s, dtype=np.uint8)
    order = np.lexsort(raw.T[::-1])  # bytewise lexicographic
    rank = np.empty(vocab, dtype=np.int64)
    rank[order] = np.arange(vocab)
    return raw[order], rank


def make_corpus(n_docs, vocab, seed=0, length="fixed", mean_len=100, zipf=None, max_len=2000):
    """Returns dict with doc_len, doc_payload, term_key (only terms that occur),
    term_start, post_doc, post_tf, and token->rank map (-1 if absent)."""
    rng = np.random.default_rng(seed)
    if length == "fixed":
        lens = np.full(n_docs, mean_len, dtype=np.int64)
    elif length == "lognormal":
        lens = np.clip(np.rint(rng.lognormal(np.log(mean_len * 0.8), 0.6, n_docs)), 8,
                       max_len).astype(np.int64)
    elif length == "mixed":  # a few very short / very long docs, exercises fieldnorm range
        lens = np.clip(np.rint(rng.lognormal(np.log(mean_len * 0.8), 1.0, n_docs)), 1,
                       max_len).astype(np.int64)
    else:
        raise ValueError(length)
    total = int(lens.sum())
    if zipf is None:
        toks = rng.integers(0, vocab, total)
    else:
        p = 1.0 / np.arange(1, vocab + 1) ** zipf
        p /= p.sum()
        toks = rng.choice(vocab, total, p=p)
    docs = np.repeat(np.arange(n_docs), lens)
    keys_sorted, rank = token_keys(vocab)
    r = rank[toks]
    # (term rank, doc) pairs -> unique with counts
    code = r * n_docs + docs
    uniq, cnt = np.unique(code, return_counts=True)
    post_rank = uniq // n_docs
    post_doc = (uniq % n_docs).astype(np.uint32)
    post_tf = cnt.astype(np.uint32)
    present = np.unique(post_rank)
    dense = np.full(vocab, -1, dtype=np.int64)
    dense[present] = np.arange(len(present))
    dr = dense[post_rank]
    term_start = np.zeros(len(present) + 1, dtype=np.uint64)
    np.add.at(term_start, dr + 1, 1)
    term_start = np.cumsum(term_start).astype(np.uint64)
    doc_payload = np.stack([(np.arange(n_docs) // 64) >> 16, (np.arange(n_docs) // 64) & 0xffff,
                            np.arange(n_docs) % 64 + 1], axis=1).astype(np.uint16)
    token_to_term = np.full(vocab, -1, dtype=np.int64)
    token_to_term[:] = dense[rank]
    return dict(
        n_docs=n_docs, doc_len=lens.astype(np.uint32), doc_payload=doc_payload,
        term_key=keys_sorted[present], term_start=term_start, post_doc=post_doc, post_tf=post_tf,
        token_to_term=token_to_term, vocab=vocab)


def make_queries(corpus, nq, n_terms, seed=1, zipf=None):
    """Distinct tokens per query drawn like document tokens; returned as ascending term ranks
    (CSR).  Tokens absent from the index are kept as rank >= n_terms (ignored by search)."""
    rng = np.random.default_rng(seed)
    vocab = corpus["vocab"]
    n_idx_terms = len(corpus["term_key"])
    terms, off = [], [0]
    if zipf is not None:
        p = 1.0 / np.arange(1, vocab + 1) ** zipf
        p /= p.sum()
    for _ in range(nq):
        if zipf is None:
            toks = rng.choice(vocab, size=min(n_terms, vocab), replace=False)
        else:
            toks = np.unique(rng.choice(vocab, size=n_terms * 3, p=p))[:n_terms]
        t = corpus["token_to_term"][toks]
        t = np.where(t < 0, n_idx_terms + toks, t)  # unknown token -> out-of-range rank
        t = np.unique(t)
        terms.extend(t.tolist())
        off.append(len(terms))
    return np.array(terms, dtype=np.uint32), np.array(off, dtype=np.uint32)
