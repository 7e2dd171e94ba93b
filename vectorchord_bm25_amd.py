"""Import shim: the package directory is `vectorchord-bm25_amd/` (a hyphen cannot appear in a
Python module name), so `import vectorchord_bm25_amd` loads it from there."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vectorchord-bm25_amd")
_spec = importlib.util.spec_from_file_location(
    "vectorchord_bm25_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["vectorchord_bm25_amd"] = _mod
_spec.loader.exec_module(_mod)
