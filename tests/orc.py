"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "liboracle.so")


class Hit(C.Structure):
    _fields_ = [("score", C.c_double), ("doc_id", C.c_uint32), ("payload", C.c_uint16 * 3),
                ("_pad", C.c_uint16)]


HIT_DTYPE = np.dtype({"names": ["score", "doc_id", "payload"],
                      "formats": ["<f8", "<u4", ("<u2", (3,))],
                      "offsets": [0, 8, 12], "itemsize": 24})
assert HIT_DTYPE.itemsize == C.sizeof(Hit) == 24

_u8p = C.POINTER(C.c_uint8)
_u16p = C.POINTER(C.c_uint16)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)


class IndexView(C.Structure):
    _fields_ = [
        ("n_docs", C.c_uint32), ("n_terms", C.c_uint32), ("n_blocks", C.c_uint32),
        ("_pad", C.c_uint32), ("sum_len", C.c_uint64), ("blob_bytes", C.c_uint64),
        ("k1", C.c_double), ("b", C.c_double),
        ("term_key", _u8p), ("term_df", _u32p), ("term_wand_fn", _u8p), ("term_wand_tf", _u32p),
        ("term_first_block", _u32p), ("blk_min_doc", _u32p), ("blk_max_doc", _u32p),
        ("blk_n", _u8p), ("blk_wand_fn", _u8p), ("blk_wand_tf", _u32p), ("blk_meta_doc", _u8p),
        ("blk_meta_tf", _u8p), ("blk_off8", _u32p), ("blob", _u8p), ("doc_fieldnorm", _u8p),
        ("doc_payload", _u16p),
    ]


def build_oracle():
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("oracle.cpp", "dense_model.inc", "pages.cpp", "oracle.h")]
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build_oracle())
    vp = C.c_void_p
    L.orc_fieldnorm_to_length.restype = C.c_uint32
    L.orc_fieldnorm_to_length.argtypes = [C.c_uint8]
    L.orc_length_to_fieldnorm.restype = C.c_uint8
    L.orc_length_to_fieldnorm.argtypes = [C.c_uint32]
    L.orc_idf.restype = C.c_double
    L.orc_idf.argtypes = [C.c_uint32, C.c_uint32]
    L.orc_tf.restype = C.c_double
    L.orc_tf.argtypes = [C.c_uint8, C.c_uint32, C.c_double, C.c_double, C.c_double]
    L.orc_cache_evaluate.restype = C.c_double
    L.orc_cache_evaluate.argtypes = [C.c_uint32, C.c_uint32, C.c_double, C.c_double, C.c_double,
                                     C.c_uint8, C.c_uint32]
    L.orc_score_from_f64.restype = C.c_int64
    L.orc_score_from_f64.argtypes = [C.c_double]
    L.orc_score_to_f64.restype = C.c_double
    L.orc_score_to_f64.argtypes = [C.c_int64]
    L.orc_compress_document_ids.restype = C.c_uint8
    L.orc_compress_document_ids.argtypes = [C.c_uint32, vp, C.c_uint32, vp, vp]
    L.orc_compress_term_frequencies.restype = C.c_uint8
    L.orc_compress_term_frequencies.argtypes = [vp, C.c_uint32, vp, vp]
    L.orc_decompress_document_ids.restype = C.c_uint32
    L.orc_decompress_document_ids.argtypes = [C.c_uint32, C.c_uint8, vp, C.c_uint32, vp]
    L.orc_decompress_term_frequencies.restype = C.c_uint32
    L.orc_decompress_term_frequencies.argtypes = [C.c_uint8, vp, C.c_uint32, vp]
    L.orc_heap_script.restype = C.c_uint32
    L.orc_heap_script.argtypes = [vp, vp, C.c_uint32, vp, vp, vp]
    L.orc_index_build.restype = vp
    L.orc_index_build.argtypes = [C.c_double, C.c_double, C.c_uint32, vp, vp, C.c_uint32, vp, vp,
                                  vp, vp]
    L.orc_index_from_view.restype = vp
    L.orc_index_from_view.argtypes = [vp]
    L.orc_index_free.restype = None
    L.orc_index_free.argtypes = [vp]
    L.orc_index_get_view.restype = None
    L.orc_index_get_view.argtypes = [vp, vp]
    L.orc_search_wand.restype = C.c_uint32
    L.orc_search_wand.argtypes = [vp, vp, C.c_uint32, C.c_uint32, vp]
    L.orc_search_brute.restype = C.c_uint32
    L.orc_search_brute.argtypes = [vp, vp, C.c_uint32, C.c_uint32, vp]
    L.orc_search_batch.restype = C.c_double
    L.orc_search_batch.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_int, C.c_int, vp, vp]
    L.orc_search_wand_growing.restype = C.c_uint32
    L.orc_search_wand_growing.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp,
                                          vp, vp, vp, vp]
    L.orc_evaluate.restype = C.c_int64
    L.orc_evaluate.argtypes = [vp, vp, vp, C.c_uint32, vp, C.c_uint32]
    L.orc_query_bytes.restype = C.c_uint64
    L.orc_query_bytes.argtypes = [vp, vp, C.c_uint32, C.c_uint32]
    L.orc_dense_model.restype = C.c_uint32
    L.orc_dense_model.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                  C.c_int, vp, vp]
    L.orc_pages_build.restype = vp
    L.orc_pages_build.argtypes = [vp, vp]
    L.orc_pages_insert.restype = None
    L.orc_pages_insert.argtypes = [vp, vp, C.c_uint32, vp, vp]
    L.orc_pages_mark_deleted_growing.restype = None
    L.orc_pages_mark_deleted_growing.argtypes = [vp, C.c_uint32]
    L.orc_pages_count.restype = C.c_uint32
    L.orc_pages_count.argtypes = [vp]
    L.orc_pages_get.restype = vp
    L.orc_pages_get.argtypes = [vp, C.c_uint32]
    L.orc_pages_get_mut.restype = vp
    L.orc_pages_get_mut.argtypes = [vp, C.c_uint32]
    L.orc_pages_free.restype = None
    L.orc_pages_free.argtypes = [vp]
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _arr(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).view(dtype).copy()


class OracleIndex:
    """Flattened index held by the oracle; arrays exposed as numpy copies."""

    def __init__(self, handle):
        self.h = handle
        v = IndexView()
        lib().orc_index_get_view(self.h, C.byref(v))
        self.n_docs, self.n_terms, self.n_blocks = v.n_docs, v.n_terms, v.n_blocks
        self.sum_len, self.k1, self.b = v.sum_len, v.k1, v.b
        nt, nb, nd = v.n_terms, v.n_blocks, v.n_docs
        self.arrays = {
            "term_key": _arr(v.term_key, 16 * nt, np.uint8).reshape(nt, 16),
            "term_df": _arr(v.term_df, nt, np.uint32),
            "term_wand_fn": _arr(v.term_wand_fn, nt, np.uint8),
            "term_wand_tf": _arr(v.term_wand_tf, nt, np.uint32),
            "term_first_block": _arr(v.term_first_block, nt + 1, np.uint32),
            "blk_min_doc": _arr(v.blk_min_doc, nb, np.uint32),
            "blk_max_doc": _arr(v.blk_max_doc, nb, np.uint32),
            "blk_n": _arr(v.blk_n, nb, np.uint8),
            "blk_wand_fn": _arr(v.blk_wand_fn, nb, np.uint8),
            "blk_wand_tf": _arr(v.blk_wand_tf, nb, np.uint32),
            "blk_meta_doc": _arr(v.blk_meta_doc, nb, np.uint8),
            "blk_meta_tf": _arr(v.blk_meta_tf, nb, np.uint8),
            "blk_off8": _arr(v.blk_off8, nb + 1, np.uint32),
            "blob": _arr(v.blob, v.blob_bytes, np.uint8),
            "doc_fieldnorm": _arr(v.doc_fieldnorm, nd, np.uint8),
            "doc_payload": _arr(v.doc_payload, 3 * nd, np.uint16).reshape(nd, 3),
        }

    @classmethod
    def build(cls, k1, b, doc_len, doc_payload, term_key, term_start, post_doc, post_tf):
        doc_len = np.ascontiguousarray(doc_len, dtype=np.uint32)
        doc_payload = np.ascontiguousarray(doc_payload, dtype=np.uint16)
        term_key = np.ascontiguousarray(term_key, dtype=np.uint8)
        term_start = np.ascontiguousarray(term_start, dtype=np.uint64)
        post_doc = np.ascontiguousarray(post_doc, dtype=np.uint32)
        post_tf = np.ascontiguousarray(post_tf, dtype=np.uint32)
        h = lib().orc_index_build(k1, b, len(doc_len), _p(doc_len), _p(doc_payload),
                                  len(term_start) - 1, _p(term_key), _p(term_start),
                                  _p(post_doc), _p(post_tf))
        return cls(h)

    @classmethod
    def from_arrays(cls, meta, arrays):
        """meta: dict(n_docs,n_terms,n_blocks,sum_len,k1,b); arrays as in self.arrays."""
        keep = {k: np.ascontiguousarray(v) for k, v in arrays.items()}
        v = IndexView()
        v.n_docs, v.n_terms, v.n_blocks = meta["n_docs"], meta["n_terms"], meta["n_blocks"]
        v.sum_len, v.k1, v.b = meta["sum_len"], meta["k1"], meta["b"]
        v.blob_bytes = keep["blob"].size
        for name, ptype in IndexView._fields_:
            if name in keep:
                setattr(v, name, keep[name].ctypes.data_as(ptype))
        return cls(lib().orc_index_from_view(C.byref(v)))

    def __del__(self):
        try:
            lib().orc_index_free(self.h)
        except Exception:
            pass

    def _search(self, fn, terms, k):
        terms = np.ascontiguousarray(terms, dtype=np.uint32)
        out = np.zeros(max(k, 1), dtype=HIT_DTYPE)
        n = fn(self.h, _p(terms), len(terms), k, _p(out))
        return out[:n]

    def search_wand(self, terms, k):
        return self._search(lib().orc_search_wand, terms, k)

    def search_brute(self, terms, k):
        return self._search(lib().orc_search_brute, terms, k)

    def dense_model(self, terms, k, wmax=16384, w0=256, lo=0, hi=None, phases=1):
        """model of the device's dense-window kernel (oracle/dense_model.inc): hits + statistics"""
        terms = np.ascontiguousarray(terms, dtype=np.uint32)
        out = np.zeros(max(k, 1), dtype=HIT_DTYPE)
        st = np.zeros(8, dtype=np.uint64)
        hi = self.n_docs if hi is None else hi
        n = lib().orc_dense_model(self.h, _p(terms), len(terms), k, wmax, w0, lo, hi, int(phases), _p(out), _p(st))
        assert n != 0xffffffff, "a 16-bit accumulator overflowed"
        names = ("windows", "blocks", "fetched_untested", "tested", "skipped", "candidates", "rescored", "phases")
        return out[:n], dict(zip(names, (int(x) for x in st)))

    def search_batch(self, terms, q_off, k, mode="wand", threads=1):
        terms = np.ascontiguousarray(terms, dtype=np.uint32)
        q_off = np.ascontiguousarray(q_off, dtype=np.uint32)
        nq = len(q_off) - 1
        out = np.zeros((nq, k), dtype=HIT_DTYPE)
        nh = np.zeros(nq, dtype=np.uint32)
        secs = lib().orc_search_batch(self.h, _p(terms), _p(q_off), nq, k,
                                      0 if mode == "wand" else 1, threads, _p(out), _p(nh))
        return out, nh, secs

    def search_wand_growing(self, terms, k, g_start, g_term, g_tf, g_fieldnorm, g_payload,
                            g_deleted):
        terms = np.ascontiguousarray(terms, dtype=np.uint32)
        g_start = np.ascontiguousarray(g_start, dtype=np.uint64)
        g_term = np.ascontiguousarray(g_term, dtype=np.uint32)
        g_tf = np.ascontiguousarray(g_tf, dtype=np.uint32)
        g_fieldnorm = np.ascontiguousarray(g_fieldnorm, dtype=np.uint8)
        g_payload = np.ascontiguousarray(g_payload, dtype=np.uint16)
        g_deleted = np.ascontiguousarray(g_deleted, dtype=np.uint8)
        out = np.zeros(max(k, 1), dtype=HIT_DTYPE)
        n = lib().orc_search_wand_growing(self.h, _p(terms), len(terms), k, len(g_start) - 1,
                                          _p(g_start), _p(g_term), _p(g_tf), _p(g_fieldnorm),
                                          _p(g_payload), _p(g_deleted), _p(out))
        return out[:n]

    def evaluate(self, doc_terms, doc_tfs, terms):
        doc_terms = np.ascontiguousarray(doc_terms, dtype=np.uint32)
        doc_tfs = np.ascontiguousarray(doc_tfs, dtype=np.uint32)
        terms = np.ascontiguousarray(terms, dtype=np.uint32)
        return lib().orc_evaluate(self.h, _p(doc_terms), _p(doc_tfs), len(doc_terms), _p(terms),
                                  len(terms))

    def query_bytes(self, terms, k):
        terms = np.ascontiguousarray(terms, dtype=np.uint32)
        return lib().orc_query_bytes(self.h, _p(terms), len(terms), k)


def compress_doc_ids(min_doc, ids):
    ids = np.ascontiguousarray(ids, dtype=np.uint32)
    out = np.zeros(512, dtype=np.uint8)
    ln = C.c_uint32(0)
    meta = lib().orc_compress_document_ids(min_doc, _p(ids), len(ids), _p(out), C.byref(ln))
    return meta, out[:ln.value].copy()


def compress_tfs(tfs):
    tfs = np.ascontiguousarray(tfs, dtype=np.uint32)
    out = np.zeros(512, dtype=np.uint8)
    ln = C.c_uint32(0)
    meta = lib().orc_compress_term_frequencies(_p(tfs), len(tfs), _p(out), C.byref(ln))
    return meta, out[:ln.value].copy()


def decompress_doc_ids(min_doc, meta, payload, stale=None):
    payload = np.ascontiguousarray(payload, dtype=np.uint8)
    out = np.zeros(128, dtype=np.uint32) if stale is None else np.array(stale, dtype=np.uint32)
    n = lib().orc_decompress_document_ids(min_doc, meta, _p(payload), len(payload), _p(out))
    return out[:n]


def decompress_tfs(meta, payload):
    payload = np.ascontiguousarray(payload, dtype=np.uint8)
    out = np.zeros(128, dtype=np.uint32)
    n = lib().orc_decompress_term_frequencies(meta, _p(payload), len(payload), _p(out))
    return out[:n]


def heap_script(keys, ops):
    keys = np.ascontiguousarray(keys, dtype=np.int64)
    ops = np.ascontiguousarray(ops, dtype=np.int32)
    popped = np.zeros(len(ops), dtype=np.int32)
    sorted_ = np.zeros(len(ops), dtype=np.int32)
    ns = C.c_uint32(0)
    npop = lib().orc_heap_script(_p(keys), _p(ops), len(ops), _p(popped), _p(sorted_),
                                 C.byref(ns))
    return popped[:npop], sorted_[:ns.value]


class Pages:
    """A relation in the reference's on-disk layout (oracle/pages.cpp): build.rs + flush.rs pages
    of an OracleIndex, plus insert.rs for unsealed documents."""

    def __init__(self, index, seed=None):
        seed = np.frombuffer(bytes(seed), np.uint8) if seed is not None else None
        self.h = C.c_void_p(lib().orc_pages_build(index.h, _p(seed) if seed is not None else None))

    def __del__(self):
        try:
            if self.h:
                lib().orc_pages_free(self.h)
        except Exception:
            pass

    def insert(self, payload, keys, tfs):
        payload = np.ascontiguousarray(payload, dtype=np.uint16)
        keys = np.frombuffer(b"".join(keys), np.uint8) if len(keys) else np.zeros(0, np.uint8)
        tfs = np.ascontiguousarray(tfs, dtype=np.uint32)
        lib().orc_pages_insert(self.h, _p(payload), len(tfs), _p(keys) if len(tfs) else None,
                               _p(tfs) if len(tfs) else None)

    def mark_deleted_growing(self, nth):
        lib().orc_pages_mark_deleted_growing(self.h, nth)

    def __len__(self):
        return int(lib().orc_pages_count(self.h))

    def address(self, i):
        """Address of page i's 8192 bytes (stable until the next insert)."""
        return lib().orc_pages_get(self.h, i) if i < len(self) else None

    def page(self, i, writable=False):
        ptr = lib().orc_pages_get_mut(self.h, i)
        buf = (C.c_uint8 * 8192).from_address(ptr)
        a = np.frombuffer(buf, dtype=np.uint8)
        return a if writable else a.copy()
