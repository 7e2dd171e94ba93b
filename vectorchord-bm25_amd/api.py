"""Python host mirror of the reference's interface for the query path.

Names follow the reference (crates/bm25): `intern` (vector.rs:19-35), `Query`
(vector.rs:96-134), `search(index, k, query)` (search.rs:28-36).  Everything here is a thin
wrapper over the C ABI in include/vbm25.h; no computation happens in Python.
"""
import ctypes as C

import numpy as np

from ._lib import IndexDesc, SynthParams, Vbm25Error, check, lib

HIT_DTYPE = np.dtype({"names": ["score", "doc_id", "payload"],
                      "formats": ["<f8", "<u4", ("<u2", (3,))],
                      "offsets": [0, 8, 12], "itemsize": 24})

WIDTH = 16  # crates/bm25/src/lib.rs:37

_DESC_ARRAYS = [
    ("term_key", np.uint8), ("term_df", np.uint32), ("term_wand_fn", np.uint8),
    ("term_wand_tf", np.uint32), ("term_first_block", np.uint32), ("blk_min_doc", np.uint32),
    ("blk_max_doc", np.uint32), ("blk_n", np.uint8), ("blk_wand_fn", np.uint8),
    ("blk_wand_tf", np.uint32), ("blk_meta_doc", np.uint8), ("blk_meta_tf", np.uint8),
    ("blk_off8", np.uint32), ("blob", np.uint8), ("doc_fieldnorm", np.uint8),
    ("doc_payload", np.uint16),
]


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


def intern(string: bytes, seed: bytes = None) -> bytes:
    """vector.rs:19-35: strings shorter than 16 bytes without NUL are zero padded; the others are the first
    16 bytes of blake3::keyed_hash(seed, string) with the last byte forced non-zero.  `seed` = MetaTuple.seed
    of the index (pages_seed())."""
    string = bytes(string)
    out = (C.c_uint8 * WIDTH)()
    if seed is not None and len(seed) != 32:
        raise ValueError("the seed is 32 bytes")
    check(lib().vbm25_intern(seed, string, len(string), out))
    return bytes(out)


class Query:
    """Sorted, de-duplicated token keys (vector.rs:96-134)."""

    def __init__(self, keys):
        keys = [bytes(k) for k in keys]
        if any(len(k) != WIDTH for k in keys) or any(a >= b for a, b in zip(keys, keys[1:])):
            raise ValueError("invalid data")  # Query::new -> expect("invalid data")
        self.keys = keys

    @classmethod
    def from_tokens(cls, tokens, seed=None):
        """cast_tsvector_to_query (src/datatype/tsvector.rs:96-105): intern, sort, dedup."""
        return cls(sorted({intern(t, seed) for t in tokens}))


class Segment:
    """Host-side sealed segment (flattened arrays), built by the library."""

    def __init__(self, handle):
        self.h = C.c_void_p(handle) if not isinstance(handle, C.c_void_p) else handle
        self.desc = IndexDesc()
        check(lib().vbm25_segment_desc(self.h, C.byref(self.desc)))

    @classmethod
    def build(cls, k1, b, doc_len, doc_payload, term_key, term_start, post_doc, post_tf, threads=0):
        doc_len = np.ascontiguousarray(doc_len, dtype=np.uint32)
        doc_payload = np.ascontiguousarray(doc_payload, dtype=np.uint16)
        term_key = np.ascontiguousarray(term_key, dtype=np.uint8)
        term_start = np.ascontiguousarray(term_start, dtype=np.uint64)
        post_doc = np.ascontiguousarray(post_doc, dtype=np.uint32)
        post_tf = np.ascontiguousarray(post_tf, dtype=np.uint32)
        out = C.c_void_p()
        check(lib().vbm25_segment_build(k1, b, len(doc_len), _p(doc_len), _p(doc_payload),
                                        len(term_start) - 1, _p(term_key), _p(term_start),
                                        _p(post_doc), _p(post_tf), threads, C.byref(out)))
        return cls(out)

    @classmethod
    def build_device(cls, k1, b, doc_len, doc_payload, term_key, term_start, post_doc, post_tf, device=0):
        """vbm25_segment_build_device: the same segment, encoded by the GPU (csrc/flush.hip)."""
        doc_len = np.ascontiguousarray(doc_len, dtype=np.uint32)
        doc_payload = np.ascontiguousarray(doc_payload, dtype=np.uint16)
        term_key = np.ascontiguousarray(term_key, dtype=np.uint8)
        term_start = np.ascontiguousarray(term_start, dtype=np.uint64)
        post_doc = np.ascontiguousarray(post_doc, dtype=np.uint32)
        post_tf = np.ascontiguousarray(post_tf, dtype=np.uint32)
        out = C.c_void_p()
        check(lib().vbm25_segment_build_device(device, k1, b, len(doc_len), _p(doc_len), _p(doc_payload),
                                               len(term_start) - 1, _p(term_key), _p(term_start),
                                               _p(post_doc), _p(post_tf), C.byref(out)))
        return cls(out)

    @classmethod
    def build_device_unsorted(cls, k1, b, doc_len, doc_payload, term_key, map_term, map_doc, map_tf, device=0):
        """vbm25_segment_build_device_unsorted: (token rank, document, tf) triples in any order; the device sorts and encodes."""
        doc_len = np.ascontiguousarray(doc_len, dtype=np.uint32)
        doc_payload = np.ascontiguousarray(doc_payload, dtype=np.uint16)
        term_key = np.ascontiguousarray(term_key, dtype=np.uint8)
        map_term = np.ascontiguousarray(map_term, dtype=np.uint32)
        map_doc = np.ascontiguousarray(map_doc, dtype=np.uint32)
        map_tf = np.ascontiguousarray(map_tf, dtype=np.uint32)
        out = C.c_void_p()
        check(lib().vbm25_segment_build_device_unsorted(device, k1, b, len(doc_len), _p(doc_len), _p(doc_payload),
                                                        len(term_key), _p(term_key), len(map_term), _p(map_term),
                                                        _p(map_doc), _p(map_tf), C.byref(out)))
        return cls(out)

    @classmethod
    def synth(cls, n_docs, vocab, mean_len=100, len_mode=1, zipf_s=0.0, k1=1.2, b=0.75,
              seed=20260925, threads=0):
        p = SynthParams(n_docs, vocab, mean_len, len_mode, zipf_s, k1, b, seed, threads, 0)
        out = C.c_void_p()
        check(lib().vbm25_segment_synth(C.byref(p), C.byref(out)))
        return cls(out)

    @classmethod
    def load(cls, path):
        out = C.c_void_p()
        check(lib().vbm25_segment_load(path.encode(), C.byref(out)))
        return cls(out)

    def save(self, path):
        check(lib().vbm25_segment_save(self.h, path.encode()))

    def __del__(self):
        try:
            lib().vbm25_segment_free(self.h)
        except Exception:
            pass

    @property
    def n_docs(self):
        return self.desc.n_docs

    @property
    def n_terms(self):
        return self.desc.n_terms

    @property
    def n_blocks(self):
        return self.desc.n_blocks

    def arrays(self):
        """numpy views (no copy) of the flattened arrays; valid while self is alive."""
        d = self.desc
        sizes = {"term_key": 16 * d.n_terms, "term_df": d.n_terms, "term_wand_fn": d.n_terms,
                 "term_wand_tf": d.n_terms, "term_first_block": d.n_terms + 1,
                 "blk_min_doc": d.n_blocks, "blk_max_doc": d.n_blocks, "blk_n": d.n_blocks,
                 "blk_wand_fn": d.n_blocks, "blk_wand_tf": d.n_blocks, "blk_meta_doc": d.n_blocks,
                 "blk_meta_tf": d.n_blocks, "blk_off8": d.n_blocks + 1, "blob": d.blob_bytes,
                 "doc_fieldnorm": d.n_docs, "doc_payload": 3 * d.n_docs}
        out = {}
        for name, dt in _DESC_ARRAYS:
            n = sizes[name]
            ptr = getattr(d, name)
            if n == 0 or not ptr:
                out[name] = np.zeros(0, dtype=dt)
                continue
            buf = (C.c_uint8 * (n * np.dtype(dt).itemsize)).from_address(ptr)
            out[name] = np.frombuffer(buf, dtype=dt)
        out["term_key"] = out["term_key"].reshape(-1, 16)
        out["doc_payload"] = out["doc_payload"].reshape(-1, 3)
        return out

    def meta(self):
        d = self.desc
        return dict(n_docs=d.n_docs, n_terms=d.n_terms, n_blocks=d.n_blocks, sum_len=d.sum_len,
                    k1=d.k1, b=d.b)

    def token_terms(self, tokens):
        tokens = np.ascontiguousarray(tokens, dtype=np.uint32)
        out = np.zeros(len(tokens), dtype=np.uint32)
        check(lib().vbm25_segment_synth_token_terms(self.h, _p(tokens), len(tokens), _p(out)))
        return out

    def query_bytes(self, term_ids, k):
        term_ids = np.ascontiguousarray(term_ids, dtype=np.uint32)
        return int(lib().vbm25_query_bytes(C.byref(self.desc), _p(term_ids), len(term_ids), k))


class DeviceSegment:
    """A sealed segment that lives in HBM (vbm25_device_segment): built or generated on the device, never downloaded unless
    asked (download() -> Segment).  GpuIndex(device_segment) makes the index of it without a round trip through the host."""

    def __init__(self, handle):
        self.h = handle
        nd, nt, nb, npost = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint64()
        check(lib().vbm25_device_segment_info(self.h, C.byref(nd), C.byref(nt), C.byref(nb), C.byref(npost)))
        self.n_docs, self.n_terms, self.n_blocks, self.n_postings = nd.value, nt.value, nb.value, npost.value

    @classmethod
    def synth(cls, n_docs, vocab, mean_len=100, len_mode=1, zipf_s=0.0, k1=1.2, b=0.75, seed=20260925, device=0):
        p = SynthParams(n_docs, vocab, mean_len, len_mode, zipf_s, k1, b, seed, 0, 0)
        out = C.c_void_p()
        check(lib().vbm25_device_segment_synth(C.byref(p), device, C.byref(out)))
        return cls(out)

    @classmethod
    def build(cls, k1, b, doc_len, doc_payload, term_key, term_start, post_doc, post_tf, device=0):
        doc_len = np.ascontiguousarray(doc_len, dtype=np.uint32)
        doc_payload = np.ascontiguousarray(doc_payload, dtype=np.uint16)
        term_key = np.ascontiguousarray(term_key, dtype=np.uint8)
        term_start = np.ascontiguousarray(term_start, dtype=np.uint64)
        post_doc = np.ascontiguousarray(post_doc, dtype=np.uint32)
        post_tf = np.ascontiguousarray(post_tf, dtype=np.uint32)
        out = C.c_void_p()
        check(lib().vbm25_device_segment_build(device, k1, b, len(doc_len), _p(doc_len), _p(doc_payload), len(term_start) - 1,
                                               _p(term_key), _p(term_start), _p(post_doc), _p(post_tf), C.byref(out)))
        return cls(out)

    def download(self):
        out = C.c_void_p()
        check(lib().vbm25_device_segment_download(self.h, C.byref(out)))
        return Segment(out)

    def token_terms(self, tokens):
        tokens = np.ascontiguousarray(tokens, dtype=np.uint32)
        out = np.zeros(len(tokens), dtype=np.uint32)
        check(lib().vbm25_device_segment_token_terms(self.h, _p(tokens), len(tokens), _p(out)))
        return out

    def query_bytes(self, term_ids, k):
        term_ids = np.ascontiguousarray(term_ids, dtype=np.uint32)
        return int(lib().vbm25_device_segment_query_bytes(self.h, _p(term_ids), len(term_ids), k))

    def __del__(self):
        try:
            if self.h:
                lib().vbm25_device_segment_free(self.h)
        except Exception:
            pass


def desc_from_arrays(meta, arrays):
    """IndexDesc over caller-owned numpy arrays (returns (desc, keepalive))."""
    keep = {}
    d = IndexDesc()
    d.n_docs, d.n_terms, d.n_blocks = meta["n_docs"], meta["n_terms"], meta["n_blocks"]
    d.sum_len, d.k1, d.b = meta["sum_len"], meta["k1"], meta["b"]
    for name, dt in _DESC_ARRAYS:
        a = np.ascontiguousarray(arrays[name], dtype=dt)
        keep[name] = a
        setattr(d, name, a.ctypes.data if a.size else None)
    d.blob_bytes = keep["blob"].size
    return d, keep


class GpuIndex:
    """HBM-resident sealed segment (vbm25_index)."""

    def __init__(self, segment_or_desc, device=0, keepalive=None):
        self.h = C.c_void_p()
        if isinstance(segment_or_desc, DeviceSegment):  # already in HBM: vbm25_index_create_from_device (on the segment's device)
            self.n_terms, self.n_docs = segment_or_desc.n_terms, segment_or_desc.n_docs
            check(lib().vbm25_index_create_from_device(segment_or_desc.h, C.byref(self.h)))
            return
        desc = segment_or_desc.desc if isinstance(segment_or_desc, Segment) else segment_or_desc
        self.n_terms = desc.n_terms
        self.n_docs = desc.n_docs
        check(lib().vbm25_index_create(C.byref(desc), device, C.byref(self.h)))

    def __del__(self):
        try:
            if self.h:
                lib().vbm25_index_destroy(self.h)
        except Exception:
            pass

    @property
    def device_bytes(self):
        return int(lib().vbm25_index_device_bytes(self.h))

    def lookup_terms(self, keys):
        """address_tokens::read for a list of 16-byte keys -> term ids (0xffffffff = absent)."""
        buf = np.frombuffer(b"".join(keys), dtype=np.uint8) if keys else np.zeros(0, np.uint8)
        out = np.zeros(len(keys), dtype=np.uint32)
        check(lib().vbm25_lookup_terms(self.h, _p(buf), len(keys), _p(out)))
        return out


class Batch:
    """Device-resident query batch (vbm25_batch)."""

    def __init__(self, index, max_queries, max_total_terms, k):
        self.index, self.k, self.nq = index, k, 0
        self.h = C.c_void_p()
        check(lib().vbm25_batch_create(index.h, max_queries, max(1, max_total_terms), k,
                                       C.byref(self.h)))

    def __del__(self):
        try:
            if self.h:
                lib().vbm25_batch_destroy(self.h)
        except Exception:
            pass

    def set_queries(self, term_ids, q_off):
        term_ids = np.ascontiguousarray(term_ids, dtype=np.uint32)
        q_off = np.ascontiguousarray(q_off, dtype=np.uint32)
        check(lib().vbm25_batch_set_queries(self.h, _p(term_ids), q_off.ctypes.data_as(C.c_void_p),
                                            len(q_off) - 1))
        self.nq = len(q_off) - 1

    def run(self, stream=None):
        check(lib().vbm25_batch_run(self.h, C.c_void_p(stream) if stream else None))

    def fetch(self):
        hits = np.zeros((self.nq, self.k), dtype=HIT_DTYPE)
        n_hits = np.zeros(self.nq, dtype=np.uint32)
        check(lib().vbm25_batch_fetch(self.h, _p(hits) if self.nq else None,
                                      _p(n_hits) if self.nq else None))
        return hits, n_hits

    def set_timing(self, enabled=True):
        check(lib().vbm25_batch_set_timing(self.h, int(enabled)))

    def kernel_ms(self):
        ms, n = C.c_double(), C.c_uint32()
        check(lib().vbm25_batch_kernel_ms(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def debug_counts(self):
        """tuning / test aid (not in include/vbm25.h): (work items of the last run, items the first-choice
        kernel handed to scan_many_kernel)"""
        f = lib().vbm25_batch_debug_counts
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        ni, nf = C.c_uint32(), C.c_uint32()
        check(f(self.h, C.byref(ni), C.byref(nf)))
        return ni.value, nf.value

    def debug_route(self):
        """test aid: 0 general route (plan_kernel), 1 one launch, 2 plan-free scan_range_kernel, 3 scan_win_kernel, 4 exhaustive"""
        f = lib().vbm25_batch_debug_route
        f.restype = C.c_int
        f.argtypes = [C.c_void_p]
        return int(f(self.h))

    def debug_routes(self):
        """bench / test aid: (queries classed sparse, dense, many-term by the host; work items of the last run on the general route
        that went to the sparse, the dense and the many-term kernel)"""
        f = lib().vbm25_batch_debug_routes
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_void_p]
        out = (C.c_uint32 * 6)()
        check(f(self.h, out))
        return tuple(int(x) for x in out)

    def debug_win_launches(self):
        """test aid: 1 = the last run's scan_win_kernel merged in the kernel (one launch), 3 = scan_win_kernel + scan_many_kernel +
        merge_kernel (also the re-run after a one-launch run that gave an item up), 0 = another route"""
        f = lib().vbm25_batch_debug_win_launches
        f.restype = C.c_int
        f.argtypes = [C.c_void_p]
        return int(f(self.h))

    def debug_theta(self, nq):
        """test aid: the thresholds (as float64 scores) the last run ended with; None if the library lacks the entry"""
        f = getattr(lib(), "vbm25_batch_debug_theta", None)
        if f is None:
            return None
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_void_p]
        out = np.zeros(nq, dtype=np.uint64)
        check(f(self.h, out.ctypes.data))
        return out.view(np.float64)

    def debug_check(self):
        """-DVBM25_CHECK builds (tools/dense_stress.py): (code, value, item, thread) of the first violated device
        assertion since the last call; code 0 = none.  A library without the entry point reports None."""
        f = getattr(lib(), "vbm25_batch_debug_check", None)
        if f is None:
            return None
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_void_p]
        out = (C.c_uint32 * 16)()
        check(f(self.h, out))
        return tuple(int(x) for x in out)


class Stream:
    """The pipelined host-buffer boundary (vbm25_stream_*): up to `depth` batches in flight, first in first out."""

    def __init__(self, index, depth, max_queries, max_total_terms, k):
        self.index, self.k = index, k
        self.h = C.c_void_p()
        L = lib()
        check(L.vbm25_stream_create(index.h, depth, max_queries, max(1, max_total_terms), k, C.byref(self.h)))
        self._nq = []

    def __del__(self):
        try:
            if self.h:
                lib().vbm25_stream_destroy(self.h)
        except Exception:
            pass

    def submit(self, term_ids, q_off):
        term_ids = np.ascontiguousarray(term_ids, dtype=np.uint32)
        q_off = np.ascontiguousarray(q_off, dtype=np.uint32)
        check(lib().vbm25_stream_submit(self.h, term_ids.ctypes.data if len(term_ids) else None, q_off.ctypes.data, len(q_off) - 1))
        self._nq.append(len(q_off) - 1)  # (only a batch the library accepted is in the ring)

    def collect_raw(self):
        """collect without this object's bookkeeping (tests: the library's own error on an empty ring)"""
        got = C.c_uint32()
        check(lib().vbm25_stream_collect(self.h, None, None, C.byref(got)))

    def collect(self, out=None):
        """The oldest batch in flight: (hits [nq, k], n_hits [nq]).  `out` = (hits, n_hits) arrays to write into."""
        if not self._nq:
            self.collect_raw()  # (the library's own error for an empty ring)
        nq = self._nq[0]
        hits, n_hits = out if out is not None else (np.zeros((nq, self.k), dtype=HIT_DTYPE), np.zeros(nq, dtype=np.uint32))
        if hits.size < nq * self.k or n_hits.size < nq:
            raise ValueError(f"collect: the output arrays hold {hits.size} records / {n_hits.size} counts, the batch needs {nq * self.k} / {nq}")
        got = C.c_uint32()
        check(lib().vbm25_stream_collect(self.h, hits.ctypes.data, n_hits.ctypes.data, C.byref(got)))
        self._nq.pop(0)  # (popped only after the library handed the batch over: an error leaves the bookkeeping in step with the ring)
        assert got.value == nq
        return hits, n_hits

    @property
    def in_flight(self):
        return int(lib().vbm25_stream_in_flight(self.h))


def search_batch_filtered(index, term_ids, q_off, k, keep, overfetch=2, return_truncated=False):
    """`prefilter = on` (default.rs:120-128, fetcher.rs:180-216: a candidate enters Results only if filter(payload) holds -- a heap
    visibility check the GPU cannot make) as the shim runs it: OVER-FETCH and filter on the host.  The GPU returns overfetch * k
    hits per query; the host keeps those `keep(hits) -> bool array` accepts; a query left with fewer than k accepted hits although
    the GPU delivered a full list is asked again, four times deeper, until k survive or its matches are exhausted (bm25.limit's
    maximum, 65535, bounds the depth as it bounds the reference's k).  Exact: the accepted hits are the first k accepted of the
    unfiltered ranking, which is what the reference's filtered search returns (ties aside).  Returns (hits[nq, k], n_hits[nq],
    rounds); with `return_truncated=True` a fourth value: per query, True when the depth reached bm25.limit's maximum with fewer than
    k accepted hits although the GPU's list was full -- the answer may then miss accepted documents beyond rank 65535 (the
    reference's k is bounded the same way)."""
    term_ids = np.ascontiguousarray(term_ids, dtype=np.uint32)
    q_off = np.ascontiguousarray(q_off, dtype=np.uint32)
    nq = len(q_off) - 1
    out = np.zeros((nq, max(k, 1)), dtype=HIT_DTYPE)
    n_out = np.zeros(nq, dtype=np.uint32)
    truncated = np.zeros(nq, dtype=bool)
    todo = np.arange(nq)
    depth = min(65535, max(k, int(k * overfetch)))
    rounds = 0
    while len(todo):
        rounds += 1
        sub_terms = np.concatenate([term_ids[q_off[q]:q_off[q + 1]] for q in todo]) if len(todo) else term_ids[:0]
        sub_off = np.concatenate([[0], np.cumsum([q_off[q + 1] - q_off[q] for q in todo])]).astype(np.uint32)
        hits, n_hits = search_batch(index, sub_terms, sub_off, depth)
        again = []
        for i, q in enumerate(todo):
            h = hits[i, :n_hits[i]]
            ok = h[np.asarray(keep(h), dtype=bool)]
            if len(ok) >= k or n_hits[i] < depth or depth == 65535:  # enough, or the query has no more matches to offer
                n_out[q] = min(k, len(ok))
                out[q, :n_out[q]] = ok[:k]
                truncated[q] = len(ok) < k and n_hits[i] == depth and depth == 65535
            else:
                again.append(q)
        todo = np.array(again, dtype=np.int64)
        depth = min(65535, depth * 4)
    if return_truncated:
        return out, n_out, rounds, truncated
    return out, n_out, rounds


class MultiIndex:
    """vbm25_multi_create: the sealed segment on several GPUs of one node -- uploaded once, replicated GPU to GPU.
    `devices` may list a device more than once (two replicas on device 0: the single-GPU test of the N-GPU path)."""

    def __init__(self, segment, devices):
        self.segment = segment
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        self.h = C.c_void_p()
        check(lib().vbm25_multi_create(C.byref(segment.desc), devs, len(devices), C.byref(self.h)))
        self.n_devices = lib().vbm25_multi_device_count(self.h)

    def __del__(self):
        try:
            if self.h:
                lib().vbm25_multi_destroy(self.h)
        except Exception:
            pass

    def search_batch(self, term_ids, q_off, k):
        """vbm25_multi_search_batch: as search_batch(), the batch cut into contiguous shards over the replicas."""
        term_ids = np.ascontiguousarray(term_ids, dtype=np.uint32)
        q_off = np.ascontiguousarray(q_off, dtype=np.uint32)
        nq = len(q_off) - 1
        hits = np.zeros((nq, max(k, 1)), dtype=HIT_DTYPE)
        n_hits = np.zeros(nq, dtype=np.uint32)
        check(lib().vbm25_multi_search_batch(self.h, _p(term_ids), q_off.ctypes.data_as(C.c_void_p), nq, k,
                                             hits.ctypes.data_as(C.c_void_p), n_hits.ctypes.data_as(C.c_void_p)))
        return hits, n_hits


class MultiBatch:
    """vbm25_multi_batch_*: the shards resident on their devices; run() is asynchronous on every device's stream
    (scan + download of the records), fetch() waits for all of them."""

    def __init__(self, multi, max_queries, max_total_terms, k):
        self.multi, self.k, self.nq = multi, k, 0
        self.h = C.c_void_p()
        check(lib().vbm25_multi_batch_create(multi.h, max_queries, max(1, max_total_terms), k, C.byref(self.h)))

    def __del__(self):
        try:
            if self.h:
                lib().vbm25_multi_batch_destroy(self.h)
        except Exception:
            pass

    def set_queries(self, term_ids, q_off):
        term_ids = np.ascontiguousarray(term_ids, dtype=np.uint32)
        q_off = np.ascontiguousarray(q_off, dtype=np.uint32)
        check(lib().vbm25_multi_batch_set_queries(self.h, _p(term_ids), q_off.ctypes.data_as(C.c_void_p), len(q_off) - 1))
        self.nq = len(q_off) - 1

    def run(self):
        check(lib().vbm25_multi_batch_run(self.h))

    def fetch(self):
        hits = np.zeros((self.nq, self.k), dtype=HIT_DTYPE)
        n_hits = np.zeros(self.nq, dtype=np.uint32)
        check(lib().vbm25_multi_batch_fetch(self.h, _p(hits) if self.nq else None, _p(n_hits) if self.nq else None))
        return hits, n_hits


def set_tuning(name, value):
    """test / tuning aid (vbm25_tuning_set, not in include/vbm25.h): process-wide switch read when a Batch / GpuIndex
    scratch batch is created.  Names: dense_x1000, dense, ne, fused, ne_ratio, dense_items, range_items,
    range_min_chunk, range_grid, dense_grid, fused_items, arith, win, win_force, win_items, win_grid, win_skew, win_guided,
    win_planes, rel16_plane and id16_plane (read at index creation).  No switch of the product library changes results (`dbg`, the
    timing experiments of scan_win_kernel, exists only in the development build libvbm25_dev.so)."""
    f = lib().vbm25_tuning_set
    f.restype = C.c_int
    f.argtypes = [C.c_char_p, C.c_longlong]
    check(f(name.encode(), int(value)))


def reset_tuning():
    f = lib().vbm25_tuning_reset
    f.restype = None
    f.argtypes = []
    f()


def search_batch(index, term_ids, q_off, k):
    """vbm25_search_batch: nq queries (CSR of ascending term ids) -> (hits[nq,k], n_hits[nq])."""
    term_ids = np.ascontiguousarray(term_ids, dtype=np.uint32)
    q_off = np.ascontiguousarray(q_off, dtype=np.uint32)
    nq = len(q_off) - 1
    hits = np.zeros((nq, max(k, 1)), dtype=HIT_DTYPE)
    n_hits = np.zeros(nq, dtype=np.uint32)
    check(lib().vbm25_search_batch(index.h, _p(term_ids), q_off.ctypes.data_as(C.c_void_p), nq, k,
                                   hits.ctypes.data_as(C.c_void_p),
                                   n_hits.ctypes.data_as(C.c_void_p)))
    return hits, n_hits


def search(index, k, query):
    """bm25::search(&index, k, &query, |_| true) (search.rs:28-36) for one Query:
    best-first list of (score, payload) with the doc id alongside."""
    ids = index.lookup_terms(query.keys)
    ids = np.sort(ids[ids != 0xffffffff])  # unknown tokens are ignored (search.rs:59-61)
    hits, n = search_batch(index, ids, np.array([0, len(ids)], dtype=np.uint32), k)
    return hits[0, :n[0]]


def growing_search(segment_or_desc, query, k, g_start, g_key, g_tf, g_fieldnorm, g_payload, g_deleted=None):
    """Host side of the shim for unsealed documents (search.rs:83-135): `query` is a Query, the
    documents are a CSR over their elements (16-byte keys + term frequencies)."""
    desc = segment_or_desc.desc if isinstance(segment_or_desc, Segment) else segment_or_desc
    keys = np.frombuffer(b"".join(query.keys), dtype=np.uint8) if query.keys else np.zeros(0, np.uint8)
    g_start = np.ascontiguousarray(g_start, dtype=np.uint64)
    g_key = np.ascontiguousarray(g_key, dtype=np.uint8)
    g_tf = np.ascontiguousarray(g_tf, dtype=np.uint32)
    g_fieldnorm = np.ascontiguousarray(g_fieldnorm, dtype=np.uint8)
    g_payload = np.ascontiguousarray(g_payload, dtype=np.uint16)
    g_deleted = None if g_deleted is None else np.ascontiguousarray(g_deleted, dtype=np.uint8)
    hits = np.zeros(max(k, 1), dtype=HIT_DTYPE)
    n = C.c_uint32()
    check(lib().vbm25_growing_search(C.byref(desc), _p(keys), len(query.keys), k, len(g_start) - 1,
                                     _p(g_start), _p(g_key), _p(g_tf), _p(g_fieldnorm), _p(g_payload),
                                     _p(g_deleted), _p(hits), C.byref(n)))
    return hits[:n.value]


def merge_hits(sealed, grow, k):
    """Top-k of the union of two best-first hit lists (the last step of the shim)."""
    sealed = np.ascontiguousarray(sealed, dtype=HIT_DTYPE)
    grow = np.ascontiguousarray(grow, dtype=HIT_DTYPE)
    out = np.zeros(max(k, 1), dtype=HIT_DTYPE)
    n = C.c_uint32()
    check(lib().vbm25_merge_hits(_p(sealed), len(sealed), _p(grow), len(grow), k, _p(out), C.byref(n)))
    return out[:n.value]


READ_PAGE_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_uint32)


class GrowingDesc(C.Structure):
    _fields_ = [("n_docs", C.c_uint32), ("_pad", C.c_uint32), ("n_elements", C.c_uint64),
                ("start", C.c_void_p), ("key", C.c_void_p), ("tf", C.c_void_p),
                ("fieldnorm", C.c_void_p), ("payload", C.c_void_p), ("deleted", C.c_void_p)]


def _page_reader(pages):
    """`pages`: a sequence of 8192-byte page images (bytes / numpy uint8), or a callable
    page_id -> address.  Returns (callback object, keepalive)."""
    if callable(pages):
        cb = READ_PAGE_FN(lambda ctx, i: pages(i))
        return cb, pages
    bufs = [np.frombuffer(bytes(p), dtype=np.uint8) if not isinstance(p, np.ndarray) else p for p in pages]
    for b in bufs:
        if b.size != 8192:
            raise ValueError("a page image is 8192 bytes")
    cb = READ_PAGE_FN(lambda ctx, i: bufs[i].ctypes.data if i < len(bufs) else None)
    return cb, bufs


def segment_from_pages(pages):
    """Flatten a bm25 index relation in the reference's on-disk format (vbm25_segment_from_pages)."""
    cb, keep = _page_reader(pages)
    out = C.c_void_p()
    check(lib().vbm25_segment_from_pages(C.cast(cb, C.c_void_p), None, C.byref(out)))
    return Segment(out)


def pages_fingerprint(pages) -> bytes:
    """vbm25_pages_fingerprint: cache key of the relation's sealed segment (changes on VACUUM / REINDEX)."""
    cb, keep = _page_reader(pages)
    out = (C.c_uint8 * 32)()
    check(lib().vbm25_pages_fingerprint(C.cast(cb, C.c_void_p), None, out))
    return bytes(out)


def pages_seed(pages) -> bytes:
    """MetaTuple.seed of the relation (the key of intern's hash)."""
    cb, keep = _page_reader(pages)
    out = (C.c_uint8 * 32)()
    check(lib().vbm25_pages_seed(C.cast(cb, C.c_void_p), None, out))
    return bytes(out)


def growing_from_pages(pages):
    """The unsealed documents of the relation as the arrays growing_search takes
    (dict: g_start, g_key, g_tf, g_fieldnorm, g_payload, g_deleted; copies)."""
    cb, keep = _page_reader(pages)
    h = C.c_void_p()
    check(lib().vbm25_growing_from_pages(C.cast(cb, C.c_void_p), None, C.byref(h)))
    try:
        d = GrowingDesc()
        check(lib().vbm25_growing_get_desc(h, C.byref(d)))

        def arr(ptr, n, dt):
            if not n or not ptr:
                return np.zeros(0, dtype=dt)
            buf = (C.c_uint8 * (n * np.dtype(dt).itemsize)).from_address(ptr)
            return np.frombuffer(buf, dtype=dt).copy()
        return dict(g_start=arr(d.start, d.n_docs + 1, np.uint64), g_key=arr(d.key, 16 * d.n_elements, np.uint8),
                    g_tf=arr(d.tf, d.n_elements, np.uint32), g_fieldnorm=arr(d.fieldnorm, d.n_docs, np.uint8),
                    g_payload=arr(d.payload, 3 * d.n_docs, np.uint16).reshape(-1, 3),
                    g_deleted=arr(d.deleted, d.n_docs, np.uint8))
    finally:
        lib().vbm25_growing_free(h)


def evaluate(segment_or_desc, doc_keys, doc_tfs, query):
    """bm25::evaluate (evaluate.rs:22-74): the `<&>` operator as a plain function; the SQL operator
    returns the negation of this value (operators.rs:54)."""
    desc = segment_or_desc.desc if isinstance(segment_or_desc, Segment) else segment_or_desc
    dk = np.frombuffer(b"".join(doc_keys), dtype=np.uint8) if len(doc_keys) else np.zeros(0, np.uint8)
    dt = np.ascontiguousarray(doc_tfs, dtype=np.uint32)
    qk = np.frombuffer(b"".join(query.keys), dtype=np.uint8) if query.keys else np.zeros(0, np.uint8)
    out = C.c_double()
    check(lib().vbm25_evaluate(C.byref(desc), _p(dk), _p(dt), len(dt), _p(qk), len(query.keys), C.byref(out)))
    return out.value


def evaluate_batch(index, q_terms, doc_start, doc_term, doc_tf):
    """vbm25_evaluate_batch: bm25::evaluate for many documents against one query on the device (term-id space)."""
    q_terms = np.ascontiguousarray(q_terms, dtype=np.uint32)
    doc_start = np.ascontiguousarray(doc_start, dtype=np.uint64)
    doc_term = np.ascontiguousarray(doc_term, dtype=np.uint32)
    doc_tf = np.ascontiguousarray(doc_tf, dtype=np.uint32)
    out = np.zeros(len(doc_start) - 1, dtype=np.float64)
    check(lib().vbm25_evaluate_batch(index.h, _p(q_terms), len(q_terms), len(doc_start) - 1,
                                     doc_start.ctypes.data_as(C.c_void_p), _p(doc_term), _p(doc_tf), _p(out)))
    return out
