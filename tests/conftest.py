import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

# The library sends batches of fewer than 64 work items to the tile kernel (single-query latency).  The
# parity tests are small by design, so they switch that rule off: every query of <= 8 terms, k <= 256 goes
# through the cursor kernel, the one that serves the benchmark.  test_tiny_batches_take_the_tile_kernel
# covers the rule itself.
os.environ.setdefault("VBM25_CUR_MIN_ITEMS", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    # GPU tests never run implicitly on a box without a GPU.
    try:
        import torch

        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
