"""Import alias for the render path of nerf-pytorch/run_nerf.py:27-424 (batchify, run_network, batchify_rays, render,
render_path, create_nerf, raw2outputs, render_rays).  The training script itself (config parsing, dataset loading,
logging; run_nerf.py:427-880) is not part of this package."""
from run_nerf_helpers import *  # noqa: F401,F403  (the reference does the same, run_nerf.py:14)
from nerf_loc_amd.render import (batchify, run_network, batchify_rays, render, render_path, create_nerf,  # noqa: F401
                                 raw2outputs, render_rays, load_checkpoint, save_checkpoint, write_video)
from nerf_loc_amd.train import train_step  # noqa: F401
from nerf_loc_amd.raybank import RayBank  # noqa: F401
