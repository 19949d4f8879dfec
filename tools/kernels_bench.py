#!/usr/bin/env python3
"""Standalone timing of the HBM-bound kernels against their algorithmic bytes (640 000 rays, the shapes of the
64 + 128 render): raw2outputs (coarse: S = 64 with weights; fine: S = 192 without) and sample_pdf + merge."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    from nerf_loc_amd import hip
    dev = torch.device("cuda:0")
    R = 640000
    g = torch.Generator(device=dev).manual_seed(0)
    rays = torch.randn(R, 11, device=dev, generator=g)
    out = {}
    for S, want_w in ((64, True), (192, False)):
        raw = torch.randn(R, S, 4, device=dev, generator=g)
        z = torch.sort(torch.rand(R, S, device=dev, generator=g) * 4 + 2, -1)[0]
        ms = timeit(lambda: hip.raw2outputs(raw, z, rays[:, 3:6], 11, None, True, want_weights=want_w))
        nbytes = R * ((24 if want_w else 20) * S + 36)
        out["raw2outputs_S%d" % S] = {"ms": ms, "GB/s": nbytes / ms / 1e6, "frac_of_8TB/s": nbytes / ms / 1e6 / 8000}
    z = torch.sort(torch.rand(R, 64, device=dev, generator=g) * 4 + 2, -1)[0]
    w = torch.rand(R, 64, device=dev, generator=g) ** 8
    u = hip.linspace01(128, dev)
    ms = timeit(lambda: hip.sample_pdf(z, w[:, 1:-1], u, bins_are_z=True, z_vals=z, want_samples=False, want_sorted=True,
                                       want_std=True))
    nbytes = R * (256 + 248 + 768 + 4)
    out["sample_pdf_merge"] = {"ms": ms, "GB/s": nbytes / ms / 1e6, "frac_of_8TB/s": nbytes / ms / 1e6 / 8000}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
