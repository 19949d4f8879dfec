"""Import alias: run_multi_nerf.py repeats run_nerf.py's render path verbatim (run_multi_nerf.py:27-424);
NeRF-Loc imports run_network from it (nerf_loc_coarse_fine/utils/nerf_utils.py:14)."""
from run_nerf import *  # noqa: F401,F403
from run_nerf import run_network, render, render_rays, batchify, batchify_rays, raw2outputs, create_nerf, render_path  # noqa: F401
