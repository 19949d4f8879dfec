"""Import alias: the package directory is named ``nerf-loc_amd`` (not a valid
Python identifier), so ``import nerf_loc_amd`` is routed to it here."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nerf-loc_amd")
_spec = importlib.util.spec_from_file_location(
    "nerf_loc_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["nerf_loc_amd"] = _mod
_spec.loader.exec_module(_mod)
