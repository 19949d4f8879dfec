"""Import alias: the pose half of nerf-pytorch/load_llff.py (no image reading: NeRF-Loc's render path takes poses)."""
import run_nerf_helpers  # noqa: F401  (puts the engine on sys.path)
from nerf_loc_amd.poses import (normalize, viewmatrix, ptstocam, poses_avg, render_path_spiral, recenter_poses,  # noqa: F401
                                spherify_poses, parse_poses_bounds, llff_poses)
