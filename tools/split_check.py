"""Quick GPU check of the split-precision network modes (bf16x3, bf16x2): accuracy against the float32 kernel and the
oracle on the g4 fixture, ragged sizes, ray input, and the launch time on the benchmark's fine pass."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nerf_loc_amd  # noqa: E402,F401
from nerf_loc_amd import hip, synth  # noqa: E402
from nerf_loc_amd.helpers import NeRF  # noqa: E402


def net(sd, dev, prec, vd=True, out_ch=4):
    n = NeRF(D=8, W=256, input_ch=63, output_ch=out_ch, skips=[4], input_ch_views=27 if vd else 0, use_viewdirs=vd)
    n.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    n = n.to(dev).requires_grad_(False)
    n.precision = prec
    return n


def main():
    dev = torch.device("cuda:0")
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "g4_mlp.npz")))
    feat = torch.from_numpy(g["feat"]).to(dev)
    for regime in "AB":
        sd = synth.make_state_dict(40, regime)
        ref = g["out_vd_" + regime]
        scale = np.abs(ref).max()
        for prec in ("fp32", "bf16x3", "bf16x2", "bf16"):
            out = net(sd, dev, prec)(feat).cpu().numpy()
            e = np.abs(out - ref)
            print("regime %s %-7s vs reference: max %.3e (%.2e of scale %.2f), rms %.3e" % (regime, prec, e.max(), e.max() / scale, scale, np.sqrt((e ** 2).mean())), flush=True)
    # timing: the fine pass of the 800 x 800 view
    R, S = int(os.environ.get("RAYS", "640000")), 192
    rays = torch.from_numpy(synth.random_rays(R, seed=3)).to(dev)
    z = hip.z_vals(rays, S)
    sd = synth.make_state_dict(1, "B")
    for prec in ("bf16", "bf16x2", "bf16x3"):
        n = net(sd, dev, prec)
        pk = n.packed_weights()
        for _ in range(2):
            hip.mlp_forward(pk, n.precision_code(), True, 4, hip.IN_RAYS, rays, z, R, S, 11)
        torch.cuda.synchronize()
        t0 = time.time()
        reps = 3
        for _ in range(reps):
            hip.mlp_forward(pk, n.precision_code(), True, 4, hip.IN_RAYS, rays, z, R, S, 11)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / reps
        flop = R * S * 1186816.0
        terms = {"bf16": 1, "bf16x2": 3, "bf16x3": 6}[prec]
        print("%-7s fine launch %d x %d: %.2f ms = %.1f TFLOP/s algorithmic, %.1f TFLOP/s issued (%d products) = %.3f of 2.5 PFLOP/s"
              % (prec, R, S, dt * 1e3, flop / dt / 1e12, flop * terms / dt / 1e12, terms, flop * terms / dt / 2.5e15), flush=True)


if __name__ == "__main__":
    main()
