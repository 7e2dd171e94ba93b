"""Ranking comparison rules (SURVEY section 7 "Hard part 1").

The reference leaves the order of equal-score hits to Rust's BinaryHeap and sums f64
partial scores in a traversal-dependent order, so two correct implementations may differ
(a) by ~1 ulp in a score and (b) in the order / membership of hits that tie.  The rule:
  * same number of hits, per-rank scores within `tol` (1e-5, BASELINE.json north_star);
  * hits are grouped into maximal runs of (nearly) equal reference score; between runs the
    order must be identical; inside a run the doc-id SETS must be equal;
  * the last run may be cut by k: there every returned doc must belong to the tie group of
    an extended (k + slack) reference ranking.
`assert_bit_exact` is the strict rule used for HIP path == canonical oracle."""
import numpy as np

TIE_REL = 1e-12


def _groups(scores):
    groups, start = [], 0
    for i in range(1, len(scores) + 1):
        if i == len(scores) or abs(scores[i] - scores[start]) > TIE_REL * max(1.0, abs(scores[start])):
            groups.append((start, i))
            start = i
    return groups


def assert_same_ranking(ref, got, ref_ext=None, tol=1e-5, what=""):
    assert len(ref) == len(got), f"{what}: {len(ref)} vs {len(got)} hits"
    if len(ref) == 0:
        return
    rs, gs = np.asarray(ref["score"]), np.asarray(got["score"])
    assert np.all(np.abs(rs - gs) <= tol), f"{what}: score mismatch {np.abs(rs - gs).max()}"
    assert np.all(np.diff(gs) <= 1e-12 * np.maximum(1.0, np.abs(gs[:-1]))), f"{what}: not sorted"
    groups = _groups(rs)
    for gi, (s, e) in enumerate(groups):
        rset = set(int(x) for x in ref["doc_id"][s:e])
        gset = set(int(x) for x in got["doc_id"][s:e])
        if rset == gset:
            continue
        last = gi == len(groups) - 1
        assert last and ref_ext is not None, f"{what}: docs differ in ranks [{s},{e}): {rset} vs {gset}"
        es = np.asarray(ref_ext["score"])
        tie = np.abs(es - rs[s]) <= TIE_REL * max(1.0, abs(rs[s]))
        allowed = set(int(x) for x in ref_ext["doc_id"][tie])
        assert gset <= allowed, f"{what}: boundary tie group {gset - allowed} not tied at k"
    # payload must be the doc's payload in both
    for r in (ref, got):
        assert r["payload"].shape == (len(r), 3)


def assert_bit_exact(ref, got, what=""):
    assert len(ref) == len(got), f"{what}: {len(ref)} vs {len(got)} hits"
    assert np.array_equal(ref["doc_id"], got["doc_id"]), f"{what}: doc ids differ\n{ref['doc_id']}\n{got['doc_id']}"
    assert np.array_equal(ref["score"].view(np.uint64), got["score"].view(np.uint64)), f"{what}: score bits differ"
    assert np.array_equal(ref["payload"], got["payload"]), f"{what}: payload differs"


def edit_distance(a, b):
    """Levenshtein distance between two id lists (the reference fuzz's tolerance metric)."""
    a, b = list(a), list(b)
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]
