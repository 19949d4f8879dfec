"""Synthetic, seed-stable inputs for the volume-rendering path.

Everything here is generated from ``numpy.random.RandomState`` so the same
numbers come out on every host and torch version.  It is used by the bench,
by the tests and by the script that produced the golden fixtures, so the
fixtures never need to store weights or rays.

Shapes follow the reference network (``run_nerf_helpers.py:67-94``): eight
256-wide point layers with the positional encoding re-injected in front of
layer 5, a 256->1 density head, a 256->256 feature layer, one 283->128 view
layer and a 128->3 colour head; without view directions a single 256->C head.
"""
import math

import numpy as np

PTS_CH = 63    # 3 + 3*2*10, get_embedder(10)  (run_nerf_helpers.py:48-63)
VIEWS_CH = 27  # 3 + 3*2*4,  get_embedder(4)


def layer_specs(D=8, W=256, input_ch=PTS_CH, input_ch_views=VIEWS_CH, output_ch=4,
                skips=(4,), use_viewdirs=True):
    """(name, out_features, in_features) in ``state_dict()`` order of the
    reference module (attribute assignment order, run_nerf_helpers.py:79-94)."""
    specs = [("pts_linears.0", W, input_ch)]
    for i in range(D - 1):
        fan_in = W + input_ch if i in skips else W
        specs.append(("pts_linears.%d" % (i + 1), W, fan_in))
    specs.append(("views_linears.0", W // 2, input_ch_views + W))
    if use_viewdirs:
        specs.append(("feature_linear", W, W))
        specs.append(("alpha_linear", 1, W))
        specs.append(("rgb_linear", 3, W // 2))
    else:
        specs.append(("output_linear", output_ch, W))
    return specs


def make_state_dict(seed=0, regime="A", **arch):
    """Random-init weights as numpy float32, ``nn.Linear`` default bounds
    (uniform in +-1/sqrt(fan_in) for weight and bias).

    regime "A": plain random init (what the reference's create_nerf gives).
    regime "B": "trained-like": the density head is scaled by 60 and its bias
    raised by 1, the colour head weight scaled by 20, which makes rays opaque
    and the render well conditioned.
    """
    rs = np.random.RandomState(seed)
    sd = {}
    for name, fo, fi in layer_specs(**arch):
        b = 1.0 / math.sqrt(fi)
        sd[name + ".weight"] = rs.uniform(-b, b, size=(fo, fi)).astype(np.float32)
        sd[name + ".bias"] = rs.uniform(-b, b, size=(fo,)).astype(np.float32)
    if regime == "B":
        if "alpha_linear.weight" in sd:
            sd["alpha_linear.weight"] = (sd["alpha_linear.weight"] * np.float32(60.0)).astype(np.float32)
            sd["alpha_linear.bias"] = (sd["alpha_linear.bias"] * np.float32(60.0) + np.float32(1.0)).astype(np.float32)
            sd["rgb_linear.weight"] = (sd["rgb_linear.weight"] * np.float32(20.0)).astype(np.float32)
        else:
            w = sd["output_linear.weight"]
            w[3] *= np.float32(60.0)
            w[:3] *= np.float32(20.0)
            sd["output_linear.bias"][3] = sd["output_linear.bias"][3] * np.float32(60.0) + np.float32(1.0)
    elif regime != "A":
        raise ValueError("regime must be 'A' or 'B'")
    return sd


def pose_spherical(theta_deg, phi_deg, radius):
    """Camera-to-world 4x4 (float32) on a sphere looking at the origin: the
    Blender-scene render path of the reference (load_blender.py:29-34)."""
    th = theta_deg / 180.0 * math.pi
    ph = phi_deg / 180.0 * math.pi
    t = np.eye(4, dtype=np.float32)
    t[2, 3] = radius
    rp = np.array([[1, 0, 0, 0],
                   [0, math.cos(ph), -math.sin(ph), 0],
                   [0, math.sin(ph), math.cos(ph), 0],
                   [0, 0, 0, 1]], dtype=np.float32)
    rt = np.array([[math.cos(th), 0, -math.sin(th), 0],
                   [0, 1, 0, 0],
                   [math.sin(th), 0, math.cos(th), 0],
                   [0, 0, 0, 1]], dtype=np.float32)
    flip = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=np.float32)
    return (flip @ (rt @ (rp @ t))).astype(np.float32)


def intrinsics(H, W, focal):
    """3x3 pinhole matrix as the drivers build it (run_nerf.py:622-627)."""
    return np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]], dtype=np.float64)


LEGO_FOCAL_800 = 0.5 * 800 / math.tan(0.5 * 0.6911112070083618)  # load_blender.py:72-73


def lego_camera(res=400, theta=30.0):
    """H, W, K, c2w[3,4], near, far of the Blender 'lego' configuration
    (configs/lego.txt, run_nerf.py:581-582) at res x res pixels."""
    focal = LEGO_FOCAL_800 * res / 800.0
    return res, res, intrinsics(res, res, focal), pose_spherical(theta, -30.0, 4.0)[:3, :4].copy(), 2.0, 6.0


def objectron_camera(pose_seed=0, coarse=False):
    """H=240, W=180 Objectron view (datasets/nerf.py:183) with the ARKit focal
    divided by pose_factor 8; the coarse field of view divides the focal by
    1.2 (main.py:176).  Pose: a camera about 1.5 units from the origin."""
    rs = np.random.RandomState(1000 + pose_seed)
    focal = 1571.66 / 8.0
    if coarse:
        focal = focal / 1.2
    K = np.array([[focal, 0, 90.0], [0, focal, 120.0], [0, 0, 1]], dtype=np.float64)
    c2w = pose_spherical(rs.uniform(-180, 180), rs.uniform(-60, -10), rs.uniform(1.2, 1.8))[:3, :4].copy()
    return 240, 180, K, c2w, 0.1, 9.0


def random_rays(n, seed=0, near=2.0, far=6.0):
    """[n, 11] float32 ray batch (origin, direction, near, far, unit view
    direction; layout of run_nerf.py:117-123): origins around (0,0,4),
    directions roughly towards the world origin with length in [0.9, 1.1]."""
    rs = np.random.RandomState(seed)
    o = rs.uniform(-1, 1, size=(n, 3)) + np.array([0.0, 0.0, 4.0])
    tgt = rs.normal(size=(n, 3)) * 0.6
    d = tgt - o
    d = d / np.linalg.norm(d, axis=-1, keepdims=True)
    d = d * rs.uniform(0.9, 1.1, size=(n, 1))
    o = o.astype(np.float32)
    d = d.astype(np.float32)
    vd = (d / np.sqrt((d * d).sum(-1, keepdims=True, dtype=np.float32))).astype(np.float32)
    nf = np.empty((n, 2), np.float32)
    nf[:, 0] = near
    nf[:, 1] = far
    return np.concatenate([o, d, nf, vd], -1).astype(np.float32)
