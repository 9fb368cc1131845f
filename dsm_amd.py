"""Import shim: the package directory is named `delayed-streams-modeling_amd` (hyphenated, after the
upstream repository), which is not a valid Python identifier.  `import dsm_amd` loads it."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "delayed-streams-modeling_amd")
_spec = importlib.util.spec_from_file_location(
    "dsm_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["dsm_amd"] = _mod
_spec.loader.exec_module(_mod)
