"""vectorchord-bm25_amd: MI355X-native BM25 top-k scorer behind VectorChord-bm25's
`bm25::search` boundary.

The product is the C-ABI library `csrc/libvbm25.so` (HIP kernels for gfx950 + host-side
segment builder, declared in include/vbm25.h).  This package is the thin Python host mirror
used by the tests and bench.py; it never touches oracle/.
"""
from ._lib import build, lib, library_path, Vbm25Error  # noqa: F401
from .api import (  # noqa: F401
    HIT_DTYPE, Segment, DeviceSegment, GpuIndex, Batch, Query, intern, search, search_batch, search_batch_filtered, growing_search, merge_hits,
    segment_from_pages, growing_from_pages, evaluate, evaluate_batch, set_tuning, reset_tuning, MultiIndex, MultiBatch, Stream)
from . import api, sharded  # noqa: F401,E402
