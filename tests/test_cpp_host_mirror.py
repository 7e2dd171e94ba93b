"""include/vbm25.hpp (C++ mirror of the reference's host interface) compiles against the C ABI
and behaves like the Python mirror."""
import os
import subprocess

import numpy as np
import pytest

import vectorchord_bm25_amd as vb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vectorchord-bm25_amd", "csrc")


def build_example(tmp_path):
    exe = str(tmp_path / "search_example")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "search_example.cpp"), "-L", CSRC, "-lvbm25",
                           f"-Wl,-rpath,{CSRC}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


def test_cpp_mirror_host_side(tmp_path):
    out = subprocess.check_output([build_example(tmp_path), "host"], text=True)
    assert "query has 3 keys; first = 10" in out  # sorted bytewise, de-duplicated
    assert "intern long lexeme without a seed: error -1" in out
    assert "intern long lexeme with a seed: last byte non-zero: 1" in out
    assert "segment: 50000 docs, 200 terms" in out
    assert "growing: 1 hit(s), merged 1, payload (1,2,3), score > 0: 1" in out
    assert "from_pages without pages: error -2" in out
    import torch
    if not torch.cuda.is_available():
        assert "index create without GPU: error -3" in out


@pytest.mark.gpu
def test_cpp_mirror_search_matches_python(tmp_path):
    out = subprocess.check_output([build_example(tmp_path)], text=True)
    seg = vb.Segment.synth(50000, 200, mean_len=40, len_mode=1, seed=7, threads=2)
    hits = vb.search(vb.GpuIndex(seg), 5, vb.Query.from_tokens([b"9", b"10", b"123", b"9"]))
    lines = [l for l in out.splitlines() if l.startswith("doc ")]
    assert len(lines) == len(hits) == 5
    for l, h in zip(lines, hits):
        f = l.split()
        assert int(f[1]) == h["doc_id"] and float(f[3]) == h["score"]
