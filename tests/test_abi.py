"""The C-ABI library loads and exports every symbol include/vbm25.h declares (no GPU use)."""
import ctypes
import os
import re

import numpy as np
import pytest

import vectorchord_bm25_amd as vb
from vectorchord_bm25_amd._lib import ABI

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "vbm25.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vbm25_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    names = declared_functions()
    assert len(names) >= 20
    L = ctypes.CDLL(vb.library_path())
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/vbm25.h but not exported"
        assert n in ABI, f"{n} has no ctypes binding"
    assert sorted(ABI) == names


def test_hit_layout_and_version():
    assert vb.HIT_DTYPE.itemsize == 24
    assert b"gfx950" in vb.lib().vbm25_version()


def test_error_reporting_without_gpu_use():
    # argument errors are reported through the status + thread-local message, never a crash
    with pytest.raises(vb.Vbm25Error) as e:
        vb.Segment.synth(0, 10)
    assert e.value.code == -1 and "positive" in str(e.value)
    with pytest.raises(vb.Vbm25Error):
        vb.Segment.load("/nonexistent/file")


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    seg = vb.Segment.synth(1000, 50, mean_len=20, threads=1)
    with pytest.raises(vb.Vbm25Error) as e:
        vb.GpuIndex(seg)
    assert e.value.code == -3  # VBM25_ERR_DEVICE: fails loudly, no CPU path
    # the device builders likewise (the host builder is a different entry point, not a fallback)
    a = seg.arrays()
    keys, doc_len = a["term_key"], np.ones(1000, dtype=np.uint32)
    payload = np.zeros((1000, 3), dtype=np.uint16)
    one = np.zeros(1, dtype=np.uint32)
    for build in (lambda: vb.Segment.build_device(1.2, 0.75, doc_len, payload, keys[:1], np.array([0, 1], dtype=np.uint64), one, one + 1),
                  lambda: vb.Segment.build_device_unsorted(1.2, 0.75, doc_len, payload, keys[:1], one, one, one + 1)):
        with pytest.raises(vb.Vbm25Error) as e:
            build()
        assert e.value.code == -3


def test_intern_and_query():
    assert vb.intern(b"12345") == b"12345" + b"\0" * 11
    with pytest.raises(vb.Vbm25Error):
        vb.intern(b"x" * 16)  # needs the index's seed
    seed = bytes(range(32))
    long_key = vb.intern(b"x" * 16, seed)
    assert len(long_key) == 16 and long_key[15] != 0 and long_key != vb.intern(b"x" * 17, seed)
    assert vb.intern(b"a\0b", seed) != b"a\0b" + b"\0" * 13  # a NUL inside takes the hash path too
    assert vb.intern(b"short", seed) == b"short" + b"\0" * 11
    q = vb.Query.from_tokens([b"9", b"10", b"9"])
    assert q.keys == [vb.intern(b"10"), vb.intern(b"9")]  # bytewise order: "10" < "9"
    with pytest.raises(ValueError):
        vb.Query([vb.intern(b"9"), vb.intern(b"10")])


def test_tuning_switches_of_the_product_library():
    """vbm25_tuning_set (exported, not in the header) knows the routing / geometry switches and refuses everything else --
    in particular `dbg`, the development switch that turns parts of scan_win_kernel off (wrong results by design): it exists
    only in libvbm25_dev.so.  Host-only: no device is touched."""
    import vectorchord_bm25_amd as vb

    try:
        for name in ("win", "win_items", "win_skew", "win_planes", "rel16_plane", "id16_plane", "arith", "fused", "dense_x1000"):
            vb.set_tuning(name, 1)
        for name in ("dbg", "team", "team_dbg", "no_such_switch"):
            with pytest.raises(RuntimeError, match="unknown tuning switch"):
                vb.set_tuning(name, 1)
    finally:
        vb.reset_tuning()
