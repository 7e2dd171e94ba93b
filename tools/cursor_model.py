"""Scalar model of the cursor scan (scan_cursor_kernel) used to check its invariants on the CPU.

Not part of the product or of the test suite's oracle: it models only the control logic of one
wave (block order, hashed marks, the two staged blocks per term, the pending list, the done bits,
the cold pass) over already decoded posting lists, with small parameters so that corner cases
(clears, overflow, chunk edges) happen often.  Scores are exact python floats (IEEE f64).
"""
import random
import sys

NONE = 1 << 32


def run_chunk(lists, tfs, score_fn, ub, k, clo, chi, BS, W, TCLR, theta0=0.0, trace=None):
    """lists[t]: sorted doc ids of term t; tfs[t][i]; score_fn(t, i) -> partial score.
    Returns list of (score, doc) offered by this chunk (deduped top-k applied by caller)."""
    m = len(lists)
    nblk = [(len(l) + BS - 1) // BS for l in lists]
    blk = lambda t, b: (b * BS, min(len(lists[t]), (b + 1) * BS))

    def first_block(t):
        for b in range(nblk[t]):
            s, e = blk(t, b)
            if lists[t][e - 1] >= clo:
                return b
        return nblk[t]

    nb = [first_block(t) for t in range(m)]

    def pos(t):
        if nb[t] >= nblk[t]:
            return NONE
        s, _ = blk(t, nb[t])
        return lists[t][s] if lists[t][s] < chi else NONE

    bm = [set(), set()]
    h1 = lambda d: d % W
    h2 = lambda d: ((d // W) * 97 + d) % W
    slots = [[None, None] for _ in range(m)]  # each: dict(b, docs, idx0, done:set, cold)
    cur = [0] * m
    pending = []
    out = []
    theta = theta0
    top = []  # local list of (score, doc)

    def offer(sc, d):
        nonlocal theta
        out.append((sc, d))
        top.append((sc, d))
        top.sort(key=lambda x: (-x[0], x[1]))
        del top[k:]
        if len(top) == k:
            theta = max(theta, top[-1][0])

    def resolve(L):
        nonlocal pending
        keep = [d for d in pending if d >= L]
        todo = [d for d in pending if d < L]
        pending = keep
        for d in todo:
            found = []
            for t in range(m):
                for s in (0, 1):
                    sl = slots[t][s]
                    if sl is None:
                        continue
                    if sl["docs"][0] <= d <= sl["docs"][-1] and d in sl["docs"]:
                        found.append((t, s, sl["docs"].index(d)))
            if not found:
                raise AssertionError("pending doc %d has no staged posting" % d)
            t0, s0, i0 = found[0]
            already = i0 in slots[t0][s0]["done"]
            for t, s, i in found:
                slots[t][s]["done"].add(i)
            if already:
                continue
            sc = 0.0
            for t, s, i in found:
                sc += score_fn(t, slots[t][s]["idx0"] + i)
            # completeness check against the full lists
            full = sum(1 for t in range(m) if d in set_lists[t])
            if full != len(found):
                raise AssertionError("doc %d: %d postings staged of %d" % (d, len(found), full))
            if sc >= theta:
                offer(sc, d)

    def cold_pass(t, sl):
        for i, d in enumerate(sl["docs"]):
            if i in sl["done"] or d < clo or d >= chi:
                continue
            full = sum(1 for u in range(m) if d in set_lists[u])
            if full != 1:
                raise AssertionError("cold pass saw doc %d with %d postings (not done)" % (d, full))
            sc = score_fn(t, sl["idx0"] + i)
            if sc >= theta:
                offer(sc, d)

    set_lists = [set(l) for l in lists]
    steps = 0
    while True:
        P = [pos(t) for t in range(m)]
        L = min(P)
        if L == NONE:
            break
        sel = P.index(L)
        old = slots[sel][cur[sel] ^ 1]
        if old is not None:
            if pending and min(pending) <= old["docs"][-1]:
                resolve(L)
            if old["cold"]:
                cold_pass(sel, old)
        s, e = blk(sel, nb[sel])
        docs = lists[sel][s:e]
        hot = theta > ub
        cur[sel] ^= 1
        slots[sel][cur[sel]] = dict(docs=docs, idx0=s, done=set(), cold=not hot)
        for d in docs:
            if d < clo or d >= chi:
                continue
            a, b = h1(d) in bm[0], h2(d) in bm[1]
            bm[0].add(h1(d))
            bm[1].add(h2(d))
            if a and b:
                pending.append(d)
        nb[sel] += 1
        steps += 1
        if steps % TCLR == 0:
            bm = [set(), set()]
            for t in range(m):
                sl = slots[t][cur[t]]
                if sl is None:
                    continue
                for d in sl["docs"]:
                    if clo <= d < chi:
                        bm[0].add(h1(d))
                        bm[1].add(h2(d))
    resolve(NONE)
    for t in range(m):
        for s in (0, 1):
            sl = slots[t][s]
            if sl is not None and sl["cold"]:
                cold_pass(t, sl)
    return out


def trial(seed):
    rng = random.Random(seed)
    N = rng.choice([200, 1000, 5000])
    m = rng.randint(1, 6)
    BS = rng.choice([4, 8, 16])
    W = rng.choice([64, 256, 1024])
    TCLR = rng.choice([1, 3, 8])
    k = rng.choice([1, 3, 10])
    lists, tfs = [], []
    base = sorted(rng.sample(range(N), rng.randint(1, N // 2)))
    for t in range(m):
        mode = rng.random()
        if mode < 0.3:   # correlated with base
            l = sorted(set(d for d in base if rng.random() < 0.8) | set(rng.sample(range(N), rng.randint(1, 20))))
        elif mode < 0.5:  # sparse
            l = sorted(rng.sample(range(N), rng.randint(1, 10)))
        else:
            l = sorted(rng.sample(range(N), rng.randint(1, N // 3)))
        lists.append(l)
        tfs.append([rng.randint(1, 4) for _ in l])
    s0 = [rng.uniform(1, 5) for _ in range(m)]
    fn = [rng.uniform(0.3, 2.0) for _ in range(N)]
    score_fn = lambda t, i: (tfs[t][i] * s0[t]) / (tfs[t][i] + fn[lists[t][i]])
    ub = max(max(score_fn(t, i) for i in range(len(lists[t]))) for t in range(m))
    # brute force
    acc = {}
    for t in range(m):
        for i, d in enumerate(lists[t]):
            acc[d] = acc.get(d, 0.0) + score_fn(t, i)
    want = sorted(((s, d) for d, s in acc.items()), key=lambda x: (-x[0], x[1]))[:k]
    nchunks = rng.randint(1, 4)
    got = []
    theta = 0.0
    for c in range(nchunks):
        clo, chi = N * c // nchunks, N * (c + 1) // nchunks
        o = run_chunk(lists, tfs, score_fn, ub, k, clo, chi, BS, W, TCLR, theta0=theta)
        got += o
        part = sorted(got, key=lambda x: (-x[0], x[1]))[:k]
        if len(part) == k:
            theta = part[-1][0]
    docs = [d for _, d in got]
    assert len(docs) == len(set(docs)), "document offered twice"
    got = sorted(got, key=lambda x: (-x[0], x[1]))[:k]
    assert got == want, (seed, got, want)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    for seed in range(n):
        trial(seed)
    print("ok", n)
