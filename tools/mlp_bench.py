#!/usr/bin/env python3
"""Times the fused network kernel alone (rays input mode) for kernel iteration and PMC runs.
usage: python tools/mlp_bench.py [--precision bf16] [--rays 163840] [--samples 192] [--reps 5]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_loc_amd import hip, synth, helpers  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="bf16")
ap.add_argument("--rays", type=int, default=163840)
ap.add_argument("--samples", type=int, default=192)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--novd", action="store_true")
ap.add_argument("--train", action="store_true", help="the training forward (keeps its tape) instead of the inference kernel")
a = ap.parse_args()
dev = torch.device("cuda:0")
vd = not a.novd
sd = synth.make_state_dict(0, "B") if vd else synth.make_state_dict(0, "B", use_viewdirs=False, output_ch=4, input_ch_views=0)
net = helpers.NeRF(D=8, W=256, input_ch=63, output_ch=4, skips=[4], input_ch_views=27 if vd else 0, use_viewdirs=vd)
net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
net = net.to(dev).requires_grad_(False)
prec = hip.PRECISIONS[a.precision]
packed = net.packed_weights(prec)
rays = torch.from_numpy(synth.random_rays(a.rays, seed=0)).to(dev)
if not vd:
    rays = rays[:, :8].contiguous()
z = hip.z_vals(rays, a.samples)
if a.train:
    M_ = a.rays * a.samples
    tape_buf = torch.empty((hip.lib().nerf_train_tape_bytes(M_, int(vd), prec) // 4,), dtype=torch.float32, device=dev)
    out_buf = torch.empty((M_, 4), dtype=torch.float32, device=dev)
    import ctypes

    def forward():      # the entry point itself: the binding would allocate a tape per call
        with torch.cuda.device(dev):
            hip._check(hip.lib().nerf_mlp_forward_train(ctypes.c_void_p(packed.data_ptr()), prec, int(vd), 4, hip.IN_RAYS,
                                                        hip._ptr(rays), hip._ptr(z), a.rays, a.samples, rays.shape[1],
                                                        hip._ptr(out_buf), ctypes.c_void_p(tape_buf.data_ptr()),
                                                        hip._stream(out_buf)), "nerf_mlp_forward_train")
        return out_buf
else:
    def forward():
        return hip.mlp_forward(packed, prec, vd, 4, hip.IN_RAYS, rays, z, a.rays, a.samples, rays.shape[1])
out = forward()
torch.cuda.synchronize()
ts = []
for _ in range(a.reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = forward()
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
M = a.rays * a.samples
t = float(np.median(ts))
fl = 1186816 if vd else 2 * 492032
print("mlp %s%s vd=%d: %d samples, median %.3f ms (min %.3f), %.1f TFLOP/s, %.2f ns per sample and CU, checksum %.6f"
      % (a.precision, " train" if a.train else "", vd, M, t, min(ts), M * fl / t / 1e9, t * 1e6 / (M / 256), float(out.abs().mean())))
try:   # diagnostic builds only (NERF_EXP_STAMP): per-block cycle sums of the sections of a pass
    import ctypes
    fn = hip.lib().nerf_debug_read_stamps
    buf = (ctypes.c_ulonglong * (256 * 8))()
    fn.argtypes = [ctypes.c_void_p]
    assert fn(buf) == 0
    st = np.array(buf[:], dtype=np.float64).reshape(256, 8)
    per = st[:, :5] / st[:, 5:6]
    names = ["inputs+encode", "L0-L4", "L5-L7", "heads", "store"]
    tot = per.sum(1).mean()
    print("stamps (cycles per pass, mean over blocks; total %.0f): " % tot +
          ", ".join("%s %.0f (%.1f%%)" % (n, v, 100 * v / tot) for n, v in zip(names, per.mean(0))))
except AttributeError:
    pass
