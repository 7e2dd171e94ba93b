"""Loads (and, on request, builds) csrc/libvbm25.so and declares its C ABI for ctypes."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
# VBM25_LIBRARY (tools/ and tuning only): another build of the library, e.g. csrc/libvbm25_chk.so
_SO = os.environ.get("VBM25_LIBRARY") or os.path.join(_CSRC, "libvbm25.so")


class Vbm25Error(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"vbm25 error {code}: {msg}")
        self.code = code


class IndexDesc(C.Structure):
    _fields_ = [
        ("n_docs", C.c_uint32), ("n_terms", C.c_uint32), ("n_blocks", C.c_uint32),
        ("_pad", C.c_uint32), ("sum_len", C.c_uint64), ("blob_bytes", C.c_uint64),
        ("k1", C.c_double), ("b", C.c_double),
        ("term_key", C.c_void_p), ("term_df", C.c_void_p), ("term_wand_fn", C.c_void_p),
        ("term_wand_tf", C.c_void_p), ("term_first_block", C.c_void_p),
        ("blk_min_doc", C.c_void_p), ("blk_max_doc", C.c_void_p), ("blk_n", C.c_void_p),
        ("blk_wand_fn", C.c_void_p), ("blk_wand_tf", C.c_void_p), ("blk_meta_doc", C.c_void_p),
        ("blk_meta_tf", C.c_void_p), ("blk_off8", C.c_void_p), ("blob", C.c_void_p),
        ("doc_fieldnorm", C.c_void_p), ("doc_payload", C.c_void_p),
    ]


class SynthParams(C.Structure):
    _fields_ = [
        ("n_docs", C.c_uint32), ("vocab", C.c_uint32), ("mean_len", C.c_uint32),
        ("len_mode", C.c_uint32), ("zipf_s", C.c_double), ("k1", C.c_double), ("b", C.c_double),
        ("seed", C.c_uint64), ("threads", C.c_int), ("_pad", C.c_int),
    ]


# name -> (restype, argtypes); every symbol include/vbm25.h declares
vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
ABI = {
    "vbm25_last_error": (C.c_char_p, []),
    "vbm25_version": (C.c_char_p, []),
    "vbm25_segment_build": (i32, [C.c_double, C.c_double, u32, vp, vp, u32, vp, vp, vp, vp, i32, vp]),
    "vbm25_segment_build_device": (i32, [i32, C.c_double, C.c_double, u32, vp, vp, u32, vp, vp, vp, vp, vp]),
    "vbm25_segment_build_device_unsorted": (i32, [i32, C.c_double, C.c_double, u32, vp, vp, u32, vp, C.c_uint64, vp, vp, vp, vp]),
    "vbm25_segment_synth": (i32, [vp, vp]),
    "vbm25_segment_synth_token_terms": (i32, [vp, vp, u32, vp]),
    "vbm25_segment_desc": (i32, [vp, vp]),
    "vbm25_segment_free": (None, [vp]),
    "vbm25_segment_save": (i32, [vp, C.c_char_p]),
    "vbm25_segment_load": (i32, [C.c_char_p, vp]),
    "vbm25_query_bytes": (u64, [vp, vp, u32, u32]),
    "vbm25_fieldnorm_table": (i32, [vp]),
    "vbm25_cache_s1": (i32, [u32, u64, C.c_double, C.c_double, vp]),
    "vbm25_growing_search": (i32, [vp, vp, u32, u32, u32, vp, vp, vp, vp, vp, vp, vp, vp]),
    "vbm25_merge_hits": (i32, [vp, u32, vp, u32, u32, vp, vp]),
    "vbm25_evaluate": (i32, [vp, vp, vp, u32, vp, u32, vp]),
    "vbm25_segment_from_pages": (i32, [vp, vp, vp]),
    "vbm25_growing_from_pages": (i32, [vp, vp, vp]),
    "vbm25_growing_get_desc": (i32, [vp, vp]),
    "vbm25_growing_free": (None, [vp]),
    "vbm25_pages_fingerprint": (i32, [vp, vp, vp]),
    "vbm25_pages_seed": (i32, [vp, vp, vp]),
    "vbm25_intern": (i32, [vp, vp, C.c_size_t, vp]),
    "vbm25_index_create": (i32, [vp, i32, vp]),
    "vbm25_index_destroy": (None, [vp]),
    "vbm25_index_device_bytes": (u64, [vp]),
    "vbm25_lookup_terms": (i32, [vp, vp, u32, vp]),
    "vbm25_search_batch": (i32, [vp, vp, vp, u32, u32, vp, vp]),
    "vbm25_batch_create": (i32, [vp, u32, u32, u32, vp]),
    "vbm25_batch_destroy": (None, [vp]),
    "vbm25_batch_set_queries": (i32, [vp, vp, vp, u32]),
    "vbm25_batch_run": (i32, [vp, vp]),
    "vbm25_batch_fetch": (i32, [vp, vp, vp]),
    "vbm25_batch_device_results": (i32, [vp, vp, vp]),
    "vbm25_batch_set_timing": (i32, [vp, i32]),
    "vbm25_batch_kernel_ms": (i32, [vp, vp, vp]),
    "vbm25_evaluate_batch": (i32, [vp, vp, u32, u32, vp, vp, vp, vp]),
    "vbm25_device_segment_build": (i32, [i32, C.c_double, C.c_double, u32, vp, vp, u32, vp, vp, vp, vp, vp]),
    "vbm25_device_segment_synth": (i32, [vp, i32, vp]),
    "vbm25_device_segment_download": (i32, [vp, vp]),
    "vbm25_device_segment_token_terms": (i32, [vp, vp, u32, vp]),
    "vbm25_device_segment_query_bytes": (u64, [vp, vp, u32, u32]),
    "vbm25_device_segment_info": (i32, [vp, vp, vp, vp, vp]),
    "vbm25_device_segment_free": (None, [vp]),
    "vbm25_index_create_from_device": (i32, [vp, vp]),
    "vbm25_multi_create": (i32, [vp, vp, i32, vp]),
    "vbm25_multi_destroy": (None, [vp]),
    "vbm25_multi_device_count": (i32, [vp]),
    "vbm25_multi_index": (i32, [vp, i32, vp]),
    "vbm25_multi_search_batch": (i32, [vp, vp, vp, u32, u32, vp, vp]),
    "vbm25_multi_batch_create": (i32, [vp, u32, u32, u32, vp]),
    "vbm25_multi_batch_destroy": (None, [vp]),
    "vbm25_multi_batch_set_queries": (i32, [vp, vp, vp, u32]),
    "vbm25_multi_batch_run": (i32, [vp]),
    "vbm25_multi_batch_fetch": (i32, [vp, vp, vp]),
    "vbm25_stream_create": (i32, [vp, u32, u32, u32, u32, vp]),
    "vbm25_stream_destroy": (None, [vp]),
    "vbm25_stream_submit": (i32, [vp, vp, vp, u32]),
    "vbm25_stream_collect": (i32, [vp, vp, vp, vp]),
    "vbm25_stream_in_flight": (i32, [vp]),
}


def library_path():
    return _SO


def build(force=False):
    """Compile csrc/ for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-s", "-C", _CSRC, "clean"])
    subprocess.check_call(["make", "-s", "-C", _CSRC])
    return _SO


_lib = None


def lib():
    """The loaded library.  There is no fallback: a missing libvbm25.so is an error."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise Vbm25Error(-3, f"{_SO} is missing: run `make -C {_CSRC}` (or "
                             "__graft_entry__.build()); the GPU path has no CPU fallback")
        L = C.CDLL(_SO)
        for name, (res, args) in ABI.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise Vbm25Error(rc, lib().vbm25_last_error().decode(errors="replace"))
