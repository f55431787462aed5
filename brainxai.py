"""Import shim: exposes the package directory ``multimodal-brain-pattern-identification_xai_amd/`` as ``brainxai``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "multimodal-brain-pattern-identification_xai_amd")
_spec = importlib.util.spec_from_file_location("brainxai", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["brainxai"] = _mod
_spec.loader.exec_module(_mod)
