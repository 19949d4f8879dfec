"""Import alias for nerf_loc_coarse_fine/utils/nerf_utils.py (`from utils.nerf_utils import create_nerf_model,
load_nerf_model, render_path`, models/model_3detr.py:15)."""
import run_nerf_helpers  # noqa: F401  (sets up sys.path)
from nerf_loc_amd.nerf_utils import (create_nerf_model, load_nerf_model, render_path, render_path_device, detector_inputs, render,  # noqa: F401
                                     render_rays, batchify_rays, raw2outputs, get_rays)
