"""Import alias: pose_spherical of nerf-pytorch/load_blender.py."""
import run_nerf_helpers  # noqa: F401
from nerf_loc_amd.poses import pose_spherical  # noqa: F401
