"""Import alias: run_multi_nerf_scannet.py is the third copy of run_nerf.py's render path
(run_multi_nerf_scannet.py:27-424, the per-video driver for ScanNet scenes)."""
from run_nerf import *  # noqa: F401,F403
from run_nerf import run_network, render, render_rays, batchify, batchify_rays, raw2outputs, create_nerf, render_path  # noqa: F401
