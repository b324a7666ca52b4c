"""Import alias: the package directory is named `llm.f90_amd/` (after the reference repo), which
is not a valid Python identifier, so `import llm_f90_amd` loads it from that directory."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "llm.f90_amd")
_spec = importlib.util.spec_from_file_location(
    "llm_f90_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["llm_f90_amd"] = _mod
_spec.loader.exec_module(_mod)
