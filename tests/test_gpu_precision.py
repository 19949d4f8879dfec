"""GPU: the reduced-precision network modes against the oracle's restatement of their arithmetic.

`oracle.mlp_forward_bf16(terms=N)` evaluates NeRF.forward (run_nerf_helpers.py:96-119) with every matrix-product operand
a sum of N bfloat16 numbers.  A kernel that agrees with it to rounding has no error of its own: what separates it from
the reference's float32 result is then the operand format.  The reference has no such modes, so no golden fixture pins
the restatement itself; its float32 limit (terms=3) is checked against the pinned float32 oracle on CPU
(tests/test_oracle_golden.py).
"""
import numpy as np
import pytest
import torch

from nerf_loc_amd import synth
from oracle import nerf_oracle as orc

from test_gpu_parity import BF16_KINK, N, T, _query, dev, load_net, maxabs  # noqa: F401  (dev is a fixture)

pytestmark = pytest.mark.gpu
F32 = np.float32


def _rms(x):
    return float(np.sqrt(np.mean(np.asarray(x, np.float64) ** 2)))


@pytest.mark.parametrize("regime", ["A", "B"])
def test_bf16_kernel_matches_bf16_restatement(dev, golden, regime):
    """nerf_mlp_forward(NERF_PREC_BF16) on encoded features against the restatement with one bf16 term per operand.
    Both round the same operands at the same places; they differ by the order of the float32 sums and by the
    activations that lie within that of a bf16 rounding boundary.  The kernel's distance from the restatement must be a
    small fraction of the restatement's distance from float32 (the format's error)."""
    g = golden("g4_mlp")
    feat = g["feat"]
    sd = synth.make_state_dict(40, regime)
    out = N(load_net(sd, dev, precision="bf16")(T(feat, dev)))
    rest = orc.mlp_forward_bf16(sd, feat, terms=1)
    f32 = g["out_vd_" + regime]
    for name, cols in (("rgb", slice(0, 3)), ("sigma", slice(3, 4))):
        fmt, own = rest[:, cols] - f32[:, cols], out[:, cols] - rest[:, cols]
        print("regime %s raw %s: format error rms %.3e max %.3e | kernel vs restatement rms %.3e max %.3e (%.1f %% of the format's rms)"
              % (regime, name, _rms(fmt), np.abs(fmt).max(), _rms(own), np.abs(own).max(), 100 * _rms(own) / _rms(fmt)))
        assert _rms(own) < 0.12 * _rms(fmt), (regime, name)      # measured 3.3 - 6.9 %
        assert np.abs(own).max() < 0.5 * np.abs(fmt).max(), (regime, name)
    # and the kernel is no further from float32 than the format is
    assert _rms(out - f32) < 1.1 * _rms(rest - f32)


def test_bf16_kernel_own_encodings_match_restatement(dev):
    """The same comparison with the kernel computing positions and encodings itself (ray input: v_fract / v_sin /
    v_cos and double-angle octaves instead of sinf / cosf): its encodings are within rounding of a bf16 value too."""
    from nerf_loc_amd import hip
    rays = synth.random_rays(64, seed=77)
    sd = synth.make_state_dict(41, "B")
    net = load_net(sd, dev, precision="bf16")
    rt = T(rays, dev)
    z = hip.z_vals(rt, 64)
    raw = N(hip.mlp_forward(net.packed_weights(), net.precision_code(), True, 4, hip.IN_RAYS, rt, z, 64, 64, 11)).reshape(64, 64, 4)
    zn = N(z)
    pts = (rays[:, None, 0:3] + rays[:, None, 3:6] * zn[:, :, None]).astype(F32)
    rest = orc.run_network(sd, pts, rays[:, 8:11], mlp=lambda s, f, vd: orc.mlp_forward_bf16(s, f, vd, terms=1))
    f32 = orc.run_network(sd, pts, rays[:, 8:11])
    fmt, own = rest - f32, raw - rest
    print("ray input: format error rms %.3e | kernel vs restatement rms %.3e (%.1f %%), max %.3e vs %.3e"
          % (_rms(fmt), _rms(own), 100 * _rms(own) / _rms(fmt), np.abs(own).max(), np.abs(fmt).max()))
    assert _rms(own) < 0.25 * _rms(fmt)      # measured 13.5 %
    assert _rms(raw - f32) < 1.15 * _rms(fmt)


def _fog_scene():
    rays = synth.random_rays(2048, seed=21)
    return rays, synth.make_state_dict(64, "B"), synth.make_state_dict(65, "B")


def test_fog_scene_bf16_on_reference_samples(dev):
    """The thin-fog scene (rays with accumulated opacity 0.03-0.3 through four units of barely positive density), bf16
    mode, fine pass evaluated at the float32 run's own depths: rgb and acc within 1e-2 on every conditioned ray.  The
    network's bf16 error is inside the bar; what exceeds it in a free-running render is where the importance samples
    land (next test)."""
    from nerf_loc_amd import hip, render
    rays, sc, sf = _fog_scene()
    f = orc.render_rays(rays, sc, 64, 128, sd_fine=sf, white_bkgd=True, retraw=True)
    f64 = orc.render_rays(rays.astype(np.float64), sc, 64, 128, sd_fine=sf, white_bkgd=True, retraw=True)
    cond = np.abs(f64["raw"][:, -1, 3]) > BF16_KINK["B"]
    nf = load_net(sf, dev, precision="bf16")
    rt, z = T(rays, dev), T(f["z_vals"], dev)
    raw = hip.mlp_forward(nf.packed_weights(), nf.precision_code(), True, 4, hip.IN_RAYS, rt, z, len(rays), 192, 11)
    rgb, _, acc, _, _ = render.raw2outputs(raw.reshape(len(rays), 192, 4), z, rt[:, 3:6], white_bkgd=True)
    e_rgb = np.abs(N(rgb) - f["rgb_map"]).max(-1)
    e_acc = np.abs(N(acc) - f["acc_map"])
    print("fog scene, bf16 fine pass on the float32 run's depths: rgb max %.2e, acc max %.2e on %d conditioned rays (%d masked)"
          % (e_rgb[cond].max(), e_acc[cond].max(), cond.sum(), (~cond).sum()))
    assert e_rgb[cond].max() < 1e-2 and e_acc[cond].max() < 1e-2
    # float32 against float64 of the oracle itself, free-running, on the same rays: the scene's own conditioning
    e_self = np.abs(f["rgb_map"] - f64["rgb_map"]).max(-1)[cond].max()
    print("  the oracle's float32 and float64 free-running renders of this scene differ by %.2e" % e_self)
    assert e_self > 1e-3      # one importance sample changing its bin moves a ray by more than ten float32 bars
