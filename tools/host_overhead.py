"""Host time of one render_rays call (64 + 128 samples, bf16) through the Python mirror (one ctypes call per entry point)
and through hip.render_rays_fused (nerf_render_rays: one call), for small batches where the GPU work is short: calls are
issued back to back without synchronising, so the figure is what the host needs per call unless the GPU is the slower
of the two (then it is the GPU's)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerf_loc_amd import helpers, hip, render, synth  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    nets = []
    for seed in (0, 1):
        n = helpers.NeRF(D=8, W=256, input_ch=63, output_ch=4, skips=[4], input_ch_views=27, use_viewdirs=True)
        n.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(seed, "B").items()})
        n = n.to(dev).requires_grad_(False)
        n.precision = "bf16"
        nets.append(n)
    q = render.NetworkQuery(helpers.get_embedder(10, 0)[0], helpers.get_embedder(4, 0)[0], 1 << 16)
    prec = nets[0].precision_code()
    pc, pf = nets[0].packed_weights(), nets[1].packed_weights()
    tv, u = hip.linspace01(64, dev), hip.linspace01(128, dev)
    out = {}
    for R in (256, 2304, 43200):
        rays = torch.from_numpy(synth.random_rays(R, seed=1)).to(dev)
        calls = {"mirror": lambda: render.render_rays(rays, nets[0], q, 64, N_importance=128, network_fine=nets[1], white_bkgd=True),
                 "one_call": lambda: hip.render_rays_fused(pc, pf, rays, 64, 128, prec, True, 4, tv, u=u, white_bkgd=True)}
        for name, fn in calls.items():
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            reps = 200
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            t_issue = (time.perf_counter() - t0) / reps
            torch.cuda.synchronize()
            t_all = (time.perf_counter() - t0) / reps
            out["%s_%d" % (name, R)] = {"host_us_per_call": 1e6 * t_issue, "wall_us_per_call": 1e6 * t_all}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
