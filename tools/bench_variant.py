#!/usr/bin/env python3
"""Run bench.py against another build of the library: tools/bench_variant.py <path/to/lib.so> [bench args]."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vectorchord_bm25_amd  # noqa: F401  (registers the package)
from vectorchord_bm25_amd import _lib

_lib._SO = os.path.abspath(sys.argv[1])
_lib._lib = None
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
