"""Import alias: the package directory is `mc-cnn_amd/` (not a valid Python
identifier), so `import mc_cnn_amd` loads it from there."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mc-cnn_amd")
_spec = importlib.util.spec_from_file_location(
    "mc_cnn_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["mc_cnn_amd"] = _mod
_spec.loader.exec_module(_mod)
