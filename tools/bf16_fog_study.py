"""Where does the bf16 mode's error on thin-fog rays come from?  CPU study with the oracle's bf16 restatement
(oracle.mlp_forward_bf16): the scene of tests/test_gpu_parity.py::test_render_bf16_vs_oracle_scenes[64-fog] rendered
with (1) every layer in plain bf16, (2) only the density head's product in three-term bf16, (3) the last k trunk
layers + heads in three terms, (4) two terms everywhere.  Prints the worst rgb / acc error against the float32 oracle
on the rays the float64 run does not mask.  Test infrastructure: nothing here is product code."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nerf_loc_amd  # noqa: E402,F401
from nerf_loc_amd import synth  # noqa: E402
from oracle import nerf_oracle as orc  # noqa: E402


def e_rgb_of(r, f32):
    return np.abs(r["rgb_map"] - f32["rgb_map"]).max(-1)


def main(n_rays=int(os.environ.get("RAYS", "512")), seed=64):
    rays = synth.random_rays(2048, seed=21)[:n_rays]
    sc, sf = synth.make_state_dict(seed, "B"), synth.make_state_dict(seed + 1, "B")
    f32 = orc.render_rays(rays, sc, 64, 128, sd_fine=sf, white_bkgd=True, retraw=True)
    f64 = orc.render_rays(rays.astype(np.float64), sc, 64, 128, sd_fine=sf, white_bkgd=True, retraw=True)
    cond = np.abs(f64["raw"][:, -1, 3]) > 0.05
    print("%d rays, %d conditioned; acc range %.3f .. %.3f" % (n_rays, cond.sum(), f32["acc_map"].min(), f32["acc_map"].max()))
    trunk = ["pts_linears.%d" % i for i in range(8)]
    heads = ["alpha_linear", "feature_linear", "views_linears.0", "rgb_linear"]
    cases = [("bf16 everywhere", dict(terms=1)),
             ("bf16, density head 3 terms", dict(terms=1, layer_terms={"alpha_linear": 3})),
             ("bf16, layer 7 + density head 3 terms", dict(terms=1, layer_terms={"alpha_linear": 3, "pts_linears.7": 3})),
             ("bf16, layers 5-7 + density head 3 terms", dict(terms=1, layer_terms=dict({k: 3 for k in trunk[5:]}, alpha_linear=3))),
             ("bf16, whole trunk + density 3 terms, colour branch bf16", dict(terms=1, layer_terms=dict({k: 3 for k in trunk}, alpha_linear=3))),
             ("two terms everywhere", dict(terms=2)),
             ("three terms everywhere", dict(terms=3))]
    for name, kw in cases:
        t0 = time.time()
        mlp = lambda sd, f, vd, kw=kw: orc.mlp_forward_bf16(sd, f, vd, **kw)
        # the fine pass on the float32 run's own sample positions: isolates the network's error from resampling
        r = orc.render_rays(rays, sc, 64, 128, sd_fine=sf, white_bkgd=True, retraw=True, mlp=mlp,
                            z_samples_override=f32["z_samples"])
        free = orc.render_rays(rays, sc, 64, 128, sd_fine=sf, white_bkgd=True, mlp=mlp)
        e_free = np.abs(free["rgb_map"] - f32["rgb_map"]).max(-1)
        worst = int(np.argmax(e_free * cond))
        print("    free-running (own importance samples): rgb max %.2e; worst ray %d: acc %.3f (float32 %.3f), forced error there %.2e"
              % (e_free[cond].max(), worst, free["acc_map"][worst], f32["acc_map"][worst], e_rgb_of(r, f32)[worst]))
        e_rgb = np.abs(r["rgb_map"] - f32["rgb_map"]).max(-1)
        e_acc = np.abs(r["acc_map"] - f32["acc_map"])
        e_raw = np.abs(r["raw"] - f32["raw"])
        print("%-58s rgb max %.2e  acc max %.2e  mean signed acc shift %+.2e  raw sigma rms %.2e max %.2e  raw rgb max %.2e  (%.0f s)"
              % (name, e_rgb[cond].max(), e_acc[cond].max(), float((r["acc_map"] - f32["acc_map"])[cond].mean()),
                 float(np.sqrt((e_raw[..., 3] ** 2).mean())), e_raw[..., 3].max(), e_raw[..., :3].max(), time.time() - t0))


if __name__ == "__main__":
    main()
