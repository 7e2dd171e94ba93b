import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))



@pytest.fixture
def tuning():
    """Test-only switches of the library (vbm25_tuning_set: read when a batch object is created), reset afterwards:
    tuning(dense_x1000=0) declares every query dense, tuning(dense=0) sends dense queries to the exhaustive
    scan_many_kernel, tuning(ne=0) switches the MaxScore split off, tuning(fused=0) the one-launch route."""
    import vectorchord_bm25_amd as vb

    def set_(**kw):
        for name, value in kw.items():
            vb.set_tuning(name, value)

    yield set_
    vb.reset_tuning()


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
