"""Import alias: `import run_nerf_helpers` / `from run_nerf_helpers import *` resolve to the MI355X engine
(names of nerf-pytorch/run_nerf_helpers.py:9-239).  Put this directory first on sys.path."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(1, _ROOT)

import nerf_loc_amd  # noqa: E402,F401
from nerf_loc_amd.helpers import (img2mse, mse2psnr, to8b, Embedder, get_embedder, NeRF, get_rays, get_rays_np,  # noqa: E402,F401
                                  ndc_rays, sample_pdf)
