"""MI355X-native NeRF volume rendering behind the nerf-pytorch function
surface that NeRF-Loc uses (run_nerf.py:27-424, run_nerf_helpers.py,
nerf_loc_coarse_fine/utils/nerf_utils.py).

Sub-modules are imported on first use so that ``nerf_loc_amd.synth`` (numpy
only) stays usable without torch or the HIP library.
"""
import importlib

__version__ = "0.1.0"

_LAZY = ("synth", "hip", "helpers", "render", "nerf_utils", "dist", "train", "raybank", "poses", "objectbank")


def __getattr__(name):
    if name in _LAZY:
        return importlib.import_module("nerf_loc_amd." + name)
    raise AttributeError(name)
