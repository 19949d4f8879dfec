#!/usr/bin/env python3
"""Does the time of the bf16 network kernel depend on the DATA?  Same launch shape (640 000 rays x 64 samples), different
operands: the kernel executes the same instructions in every case, so any difference is the clock the chip holds under
its power limit.  usage: python tools/mlp_data_power.py [--reps 7]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from nerf_loc_amd import hip, synth, helpers  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=7)
a = ap.parse_args()
dev = torch.device("cuda:0")
R, S = 640000, 64
prec = hip.PRECISIONS["bf16"]


def net_of(sd):
    n = helpers.NeRF(D=8, W=256, input_ch=63, output_ch=4, skips=[4], input_ch_views=27, use_viewdirs=True)
    n.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return n.to(dev).requires_grad_(False)


def run(name, net, rays):
    packed = net.packed_weights(prec)
    z = hip.z_vals(rays, S)
    ts = []
    for i in range(a.reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        hip.mlp_forward(packed, prec, True, 4, hip.IN_RAYS, rays, z, R, S, rays.shape[1])
        e1.record()
        torch.cuda.synchronize()
        if i:
            ts.append(e0.elapsed_time(e1))
    t = float(np.median(ts))
    print("%-58s median %.3f ms  %.1f TFLOP/s" % (name, t, R * S * bench.FLOP_PER_SAMPLE / t / 1e9), flush=True)


H, W, K, c2w, kw = bench.build_scene(dev, "bf16", 800)
view = hip.make_rays(H, W, K, c2w=c2w, near=kw["near"], far=kw["far"], use_viewdirs=True)
rnd = torch.from_numpy(synth.random_rays(R, seed=0)).to(dev)
sd0 = synth.make_state_dict(0, "B")
zero = {k: np.zeros_like(v) for k, v in sd0.items()}
run("random rays, seeded network (tools/mlp_bench.py)", net_of(sd0), rnd)
run("view rays of the bench scene, its coarse network", kw["network_fn"], view)
run("view rays of the bench scene, its fine network", kw["network_fine"], view)
run("random rays, fine network of the bench scene", kw["network_fine"], rnd)
run("random rays, all-zero network", net_of(zero), rnd)
run("random rays, seeded network (again)", net_of(sd0), rnd)
