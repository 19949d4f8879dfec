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


# ------------------------------------------------------------------ float32 accuracy on the bf16 matrix cores
@pytest.mark.parametrize("prec,bar", [("bf16x3", 2e-5), ("bf16x2", 3e-4)])
def test_split_network_vs_reference(dev, golden, prec, bar):
    """NeRF.forward of the reference (g4: its float32 outputs on these features and weights) in the split-operand modes:
    three bf16 terms per operand hold the float32 kernel's own bar (2e-5 of the output scale), two terms 3e-4; all three
    architectures (view directions; none with 5 and with 4 output rows)."""
    g = golden("g4_mlp")
    feat = T(g["feat"], dev)
    worst = 0.0
    for regime in "AB":
        sd = synth.make_state_dict(40, regime)
        out = N(load_net(sd, dev, precision=prec)(feat))
        ref = g["out_vd_" + regime]
        e = maxabs(out, ref) / max(1.0, np.abs(ref).max())
        own = maxabs(out, orc.mlp_forward_bf16(sd, g["feat"], terms=int(prec[-1]))) / max(1.0, np.abs(ref).max())
        print("%s regime %s: max error vs reference %.2e of scale, vs the %s-term restatement %.2e" % (prec, regime, e, prec[-1], own))
        worst = max(worst, e)
        assert own < bar
    n5 = load_net(synth.make_state_dict(41, "A", use_viewdirs=False, output_ch=5, input_ch_views=0), dev, False, 5, precision=prec)
    worst = max(worst, maxabs(N(n5(feat[:, :63].contiguous())), g["out_novd5_A"]))
    n4 = load_net(synth.make_state_dict(42, "B", use_viewdirs=False, output_ch=4, input_ch_views=0), dev, False, 4, precision=prec)
    ref = g["out_novd4_B"]
    worst = max(worst, maxabs(N(n4(feat[:, :63].contiguous())), ref) / max(1.0, np.abs(ref).max()))
    assert worst < bar, worst


def test_split_network_input_modes_agree(dev, golden):
    """Features, points and rays as input of the bf16x3 kernel: its own float32 sine / cosine encodings against the
    oracle's, and positions o + d z formed in registers."""
    from nerf_loc_amd import hip
    rays = synth.random_rays(96, seed=5)
    sd = synth.make_state_dict(43, "B")
    net = load_net(sd, dev, precision="bf16x3")
    rt = T(rays, dev)
    z = hip.z_vals(rt, 48)
    raw_rays = N(hip.mlp_forward(net.packed_weights(), net.precision_code(), True, 4, hip.IN_RAYS, rt, z, 96, 48, 11))
    pts = (rays[:, None, 0:3] + rays[:, None, 3:6] * N(z)[:, :, None]).astype(F32)
    raw_pts = N(hip.mlp_forward(net.packed_weights(), net.precision_code(), True, 4, hip.IN_POINTS, T(pts.reshape(-1, 3), dev),
                                rt[:, 8:11].contiguous(), 96, 48))
    ref = orc.run_network(sd, pts, rays[:, 8:11]).reshape(-1, 4)
    scale = max(1.0, np.abs(ref).max())
    print("bf16x3 ray input vs oracle %.2e, point input %.2e (scale %.1f)" % (maxabs(raw_rays, ref), maxabs(raw_pts, ref), scale))
    assert np.array_equal(raw_rays, raw_pts)
    assert maxabs(raw_rays, ref) < 2e-5 * scale


@pytest.mark.parametrize("prec", ["bf16x3", "bf16x2"])
@pytest.mark.parametrize("regime", ["A", "B"])
def test_render_bf16x3_vs_reference_golden(dev, golden, regime, prec):
    """The float32 contract (every map within 1e-4 absolute of the reference's render_rays, g5, free-running: importance
    samples from the kernel's own coarse weights) in the bf16x3 mode.  bf16x2 (operands good to 2^-16) is NOT a float32
    substitute: measured rgb 6.1e-5 and disp 1.2e-4 on the trained-like weights, so its bar here is 3e-4."""
    bar = {"bf16x3": 1e-4, "bf16x2": 3e-4}[prec]
    from nerf_loc_amd import render
    g = golden("g5_render_rays")
    rays = g["rays"]
    nc = load_net(synth.make_state_dict(50, regime), dev, precision=prec)
    nf = load_net(synth.make_state_dict(51, regime), dev, precision=prec)
    r = render.render_rays(T(rays, dev), nc, _query(dev), 64, retraw=True, N_importance=128, network_fine=nf, white_bkgd=True)
    r = {k: N(v) for k, v in r.items()}
    tag = "cf_" + regime
    assert (np.abs(g["cf64_%s_rawlast" % regime]) > 1e-4).all()      # no ray of the fixture needs masking
    for k in ("rgb0", "acc0", "disp0", "rgb_map", "acc_map", "disp_map"):
        e = maxabs(r[k], g["%s_%s" % (tag, k)])
        print("  %s regime %s %s max abs err %.3e" % (prec, regime, k, e))
        assert e < bar, (prec, k)
    assert maxabs(r["z_std"], g[tag + "_z_std"]) < 2 * bar


@pytest.mark.parametrize("prec", ["bf16x3", "bf16x2"])
def test_bf16x3_view_rays_and_loc_image(dev, golden, prec):
    """4096 pixels of the benchmark's 800 x 800 view (g12) and NeRF-Loc's render() of a 240 x 180 view (g13), both
    rendered by the reference: the bf16x3 mode inside the float32 bars (1e-4 on every map, rays within 1e-4 of the ReLU
    kink at the last sample masked by the reference's float64 run and counted)."""
    import types
    from nerf_loc_amd import nerf_utils, render
    g = golden("g12_view_rays")
    rays, rawlast = g["rays"], g["f64_rawlast"]
    nc = load_net(synth.make_state_dict(0, "B"), dev, precision=prec)
    nf = load_net(synth.make_state_dict(1, "B"), dev, precision=prec)
    r = render.render_rays(T(rays, dev), nc, _query(dev), 64, N_importance=128, network_fine=nf, white_bkgd=True)
    r = {k: N(v) for k, v in r.items()}
    cond = np.abs(rawlast) > 1e-4
    assert (~cond).sum() <= 8
    for k in ("rgb_map", "acc_map", "rgb0", "acc0"):
        e = np.abs(r[k] - g["f32_" + k]).reshape(len(rays), -1).max(-1)[cond].max()
        print("  %s view rays %s max abs err %.2e" % (prec, k, e))
        assert e < 1e-4, k
    assert np.abs(r["z_std"] - g["f32_z_std"])[cond].max() < 2e-4
    g = golden("g13_loc_image")
    H, W = [int(v) for v in g["hw"]]
    idx = g["idx"]
    args = types.SimpleNamespace(nerf_render_rgb=True, nerf_render_depth=True)
    _, kw, _ = nerf_utils.create_nerf_model(args, ckpt_path=None, device=dev, near=0.1, far=9.0)
    kw["network_fn"].load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(100, "B").items()})
    kw["network_fn"].requires_grad_(False)
    kw["network_fn"].precision = prec
    rgb, disp, acc, depth, _ = nerf_utils.render(H, W, g["K"], chunk=1024 * 32, c2w=T(g["c2w"], dev), **kw)
    pick = lambda v: N(v).reshape(H * W, -1)[idx]
    cond = np.abs(g["raw_sigma_last"]) > 1e-4
    assert (~cond).sum() <= 4
    e_rgb = np.abs(pick(rgb) - g["rgb"]).max(-1)[cond].max()
    e_acc = np.abs(pick(acc)[:, 0] - g["acc"])[cond].max()
    e_depth = np.abs(pick(depth)[:, 0] - g["depth"])[cond].max()
    print("  %s NeRF-Loc 240x180: rgb %.2e acc %.2e depth %.2e" % (prec, e_rgb, e_acc, e_depth))
    assert e_rgb < 1e-4 and e_acc < 1e-4 and e_depth < 1e-4


def test_fog_scene_coarse_network_in_two_terms(dev):
    """The thin-fog scene free-running.  In plain bf16 the render exceeds the 1e-2 bar (3.6e-2, kept as a strict xfail in
    test_gpu_parity.py) because the coarse weights decide where the 128 importance samples land, and one sample changing
    its bin moves such a ray by 1e-3 .. 1e-2: the oracle's own float32 and float64 renders of this scene differ by
    1.5e-2.  With the COARSE network in the bf16x2 mode (a quarter of the evaluations at three products instead of
    one: about 1.5 x the frame time) and the fine network in bf16 the render is inside the scene's own conditioning:
    bar = the larger of 1e-2 and that float32 / float64 difference (measured rgb 9.2e-3, acc 1.1e-2)."""
    from nerf_loc_amd import render
    rays, sc, sf = _fog_scene()
    f = orc.render_rays(rays, sc, 64, 128, sd_fine=sf, white_bkgd=True)
    f64 = orc.render_rays(rays.astype(np.float64), sc, 64, 128, sd_fine=sf, white_bkgd=True, retraw=True)
    cond = np.abs(f64["raw"][:, -1, 3]) > BF16_KINK["B"]
    nc, nf = load_net(sc, dev, precision="bf16x2"), load_net(sf, dev, precision="bf16")
    b = render.render_rays(T(rays, dev), nc, _query(dev), 64, N_importance=128, network_fine=nf, white_bkgd=True)
    e_rgb = np.abs(N(b["rgb_map"]) - f["rgb_map"]).max(-1)
    e_acc = np.abs(N(b["acc_map"]) - f["acc_map"])
    e_self = float(np.abs(f["rgb_map"] - f64["rgb_map"]).max(-1)[cond].max())
    print("fog scene free-running, coarse bf16x2 + fine bf16: rgb max %.2e, acc max %.2e on %d conditioned rays; the oracle's "
          "float32 vs float64: %.2e" % (e_rgb[cond].max(), e_acc[cond].max(), cond.sum(), e_self))
    bar = max(1e-2, e_self)
    assert e_rgb[cond].max() < bar and e_acc[cond].max() < bar
