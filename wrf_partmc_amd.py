"""Import alias: the package directory is `wrf-partmc_amd/` (not a valid Python
identifier), so `import wrf_partmc_amd` loads it from there."""
import importlib.util
import os
import sys

_d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "wrf-partmc_amd")
_spec = importlib.util.spec_from_file_location(
    "wrf_partmc_amd", os.path.join(_d, "__init__.py"), submodule_search_locations=[_d])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["wrf_partmc_amd"] = _mod
_spec.loader.exec_module(_mod)
