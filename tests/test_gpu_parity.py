"""GPU parity: every HIP entry point against the numpy oracle and the golden
fixtures, called the way a user of the reference would call them.

Tolerances: deterministic sample_pdf indices and
samples bit-exact for identical inputs; float32 maps within 1e-4 absolute; the
bf16-MFMA mode within 1e-2 absolute and 0.1 dB PSNR.
"""
import os

import numpy as np
import pytest
import torch

from nerf_loc_amd import synth
from oracle import nerf_oracle as orc

pytestmark = pytest.mark.gpu
F32 = np.float32


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from nerf_loc_amd import hip
    hip.lib()
    return torch.device("cuda:0")


def T(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def N(x):
    return x.detach().cpu().numpy()


def maxabs(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    both_nan = np.isnan(a) & np.isnan(b)
    d = np.abs(a - b)
    d[both_nan] = 0.0
    assert not np.isnan(d).any(), "NaN pattern differs"
    return float(d.max()) if d.size else 0.0


def load_net(sd, dev, use_viewdirs=True, output_ch=4, precision="fp32"):
    from nerf_loc_amd.helpers import NeRF
    net = NeRF(D=8, W=256, input_ch=63, output_ch=output_ch, skips=[4], input_ch_views=27 if use_viewdirs else 0,
               use_viewdirs=use_viewdirs)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net = net.to(dev).requires_grad_(False)
    net.precision = precision
    return net


def np_rand(shape, scale=1.0):
    np.random.seed(0)
    return (np.random.rand(*shape) * scale).astype(F32)


# ------------------------------------------------------------------ basics
def test_mfma_lane_maps(dev):
    from nerf_loc_amd import hip
    hip.selftest()


def test_make_rays_and_ndc(dev, golden):
    from nerf_loc_amd import helpers
    g = golden("g6_rays")
    for tag in ("lego40", "lego400", "obj", "objc"):
        H, W = [int(v) for v in g[tag + "_hw"]]
        o, d = helpers.get_rays(H, W, g[tag + "_K"], T(g[tag + "_c2w"], dev))
        assert o.shape == (H, W, 3) and d.shape == (H, W, 3)
        o, d = N(o).reshape(-1, 3), N(d).reshape(-1, 3)
        if tag + "_idx" in g:
            o, d = o[g[tag + "_idx"]], d[g[tag + "_idx"]]
        assert maxabs(o, g[tag + "_o"].reshape(-1, 3)) == 0.0
        assert maxabs(d, g[tag + "_d"].reshape(-1, 3)) < 2e-7, tag
    H, W, f = g["ndc_hwf"]
    no, nd = helpers.ndc_rays(int(H), int(W), float(f), 1.0, T(g["ndc_o_in"], dev), T(g["ndc_d_in"], dev))
    assert maxabs(N(no), g["ndc_o"]) < 1e-5 * max(1, np.abs(g["ndc_o"]).max())
    assert maxabs(N(nd), g["ndc_d"]) < 1e-5 * max(1, np.abs(g["ndc_d"]).max())


def test_static_camera_rays(dev):
    """c2w_staticcam (run_nerf.py:102-107): consulted only with use_viewdirs; geometry then comes from the static
    camera, view directions from c2w (or from the given ray batch, which must then have H*W rays)."""
    from nerf_loc_amd import render
    H, W, K, c2w, near, far = synth.lego_camera(16, theta=40.0)
    static = synth.pose_spherical(-75.0, -20.0, 3.5)[:3, :4].astype(F32)
    so, sd = orc.get_rays(H, W, K, static)
    vo, vd = orc.get_rays(H, W, K, c2w.astype(F32))
    want = orc.pack_rays(so, sd, near, far, True, viewdirs_from=vd)
    flat, sh = render._build_rays(H, W, K, None, T(c2w.astype(F32), dev), False, near, far, True, T(static, dev))
    assert sh == [H, W] and maxabs(N(flat)[:, :8], want[:, :8]) < 1e-6 and maxabs(N(flat)[:, 8:], want[:, 8:]) < 2e-7
    # without view directions the static camera is ignored
    flat0, _ = render._build_rays(H, W, K, None, T(c2w.astype(F32), dev), False, near, far, False, T(static, dev))
    assert maxabs(N(flat0), orc.pack_rays(vo, vd, near, far, False)) < 1e-6
    # a ray batch as the source of the view directions
    rb = (T(vo.reshape(-1, 3), dev), T(vd.reshape(-1, 3), dev))
    flat2, sh2 = render._build_rays(H, W, K, rb, None, False, near, far, True, T(static, dev))
    assert sh2 == [H, W] and torch.equal(flat2, flat)
    with pytest.raises(ValueError):
        render._build_rays(H, W, K, (rb[0][:7], rb[1][:7]), None, False, near, far, True, T(static, dev))
    flat3, sh3 = render._build_rays(H, W, K, (rb[0][:7], rb[1][:7]), None, False, near, far, False, T(static, dev))
    assert sh3 == [7] and maxabs(N(flat3), orc.pack_rays(vo, vd, near, far, False)[:7]) < 1e-6


def test_ndc_rays_any_near_plane(dev):
    """ndc_rays takes the near plane as an argument (run_nerf_helpers.py:175-192); render() passes 1.0, callers may not."""
    from nerf_loc_amd import helpers
    rs = np.random.RandomState(4)
    o = (rs.uniform(-1, 1, size=(64, 3)) + np.array([0, 0, 0.5])).astype(F32)
    d = rs.normal(size=(64, 3)).astype(F32)
    d[:, 2] = -np.abs(d[:, 2]) - 0.2
    for near in (1.0, 0.5, 2.0):
        no, nd = helpers.ndc_rays(378, 504, 407.5, near, T(o, dev), T(d, dev))
        ro, rd = orc.ndc_rays(378, 504, 407.5, near, o, d)
        assert maxabs(N(no), ro) < 1e-5 * max(1, np.abs(ro).max()) and maxabs(N(nd), rd) < 1e-5 * max(1, np.abs(rd).max())


def test_z_vals(dev):
    from nerf_loc_amd import hip
    rays = synth.random_rays(300, seed=3, near=0.1, far=9.0)
    rays[:, 6] += np.linspace(0, 1, 300).astype(F32)
    for S in (64, 5, 192):
        for lindisp in (False, True):
            z = N(hip.z_vals(T(rays, dev), S, lindisp))
            assert np.array_equal(z, orc.z_vals_for(rays, S, lindisp)), (S, lindisp)
            tr = np_rand((300, S))
            z = N(hip.z_vals(T(rays, dev), S, lindisp, T(tr, dev)))
            assert maxabs(z, orc.z_vals_for(rays, S, lindisp, tr)) < 1e-6


def test_embed(dev, golden):
    from nerf_loc_amd import helpers
    g = golden("g3_embed")
    e10, c10 = helpers.get_embedder(10, 0)
    e4, c4 = helpers.get_embedder(4, 0)
    assert (c10, c4) == (63, 27)
    assert maxabs(N(e10(T(g["x"], dev))), g["embed10"]) < 5e-7
    assert maxabs(N(e4(T(g["d"], dev))), g["embed4"]) < 5e-7
    ident, c = helpers.get_embedder(10, -1)
    assert c == 3 and isinstance(ident, torch.nn.Identity)


# ------------------------------------------------------------------ compositing
@pytest.mark.parametrize("S", [64, 192, 5])
def test_raw2outputs_golden(dev, golden, S):
    from nerf_loc_amd import render, nerf_utils
    g = golden("g2_raw2outputs")
    raw, z, d = g["s%d_raw" % S], g["s%d_z" % S], g["s%d_d" % S]
    for tag, kw in (("k", dict(white_bkgd=False)), ("w", dict(white_bkgd=True)),
                    ("n", dict(white_bkgd=False, raw_noise_std=1.0, pytest=True))):
        out = render.raw2outputs(T(raw, dev), T(z, dev), T(d, dev), **kw)
        for name, v in zip(("rgb", "disp", "acc", "weights", "depth"), out):
            ref = g["s%d_%s_%s" % (S, tag, name)]
            v = N(v)
            if name == "disp":
                assert np.array_equal(np.isnan(v), np.isnan(ref)), "disp must be NaN exactly where acc == 0"
                fin = np.isfinite(ref) & (np.abs(ref) < 1e6)
                assert maxabs(v[fin] / ref[fin], np.ones(fin.sum())) < 2e-5
            else:
                assert maxabs(v, ref) < 1e-5, (tag, name, maxabs(v, ref))
    # NeRF-Loc variant: switched-off maps are None
    r = nerf_utils.raw2outputs(T(raw, dev), T(z, dev), T(d, dev), nerf_render_rgb=False)
    assert r[0] is None and maxabs(N(r[4]), g["s%d_k_depth" % S]) < 1e-5
    assert nerf_utils.raw2outputs(T(raw, dev), T(z, dev), T(d, dev), nerf_render_rgb=False,
                                  nerf_render_depth=False) == (None,) * 5


def test_raw2outputs_5ch_and_large(dev):
    from nerf_loc_amd import render
    rs = np.random.RandomState(7)
    R, S = 5000, 192
    raw = rs.normal(size=(R, S, 5)).astype(F32) * 3
    z = np.sort(rs.uniform(2, 6, size=(R, S)), -1).astype(F32)
    d = rs.normal(size=(R, 3)).astype(F32)
    out = render.raw2outputs(T(raw, dev), T(z, dev), T(d, dev), white_bkgd=True)
    ref = orc.raw2outputs(raw, z, d, white_bkgd=True)
    for a, b, name in zip(out, ref, ("rgb", "disp", "acc", "weights", "depth")):
        if name == "disp":
            continue
        assert maxabs(N(a), b) < 1e-5, name
    w = N(out[3])
    assert maxabs(w.sum(-1), N(out[2])) < 1e-5          # acc is the sum of the weights
    assert (w >= 0).all() and (N(out[2]) <= 1 + 1e-5).all()


def test_make_rays_views_equals_pose_by_pose(dev):
    """Several poses with their own intrinsics in one launch: the same bits as one make_rays per pose."""
    from nerf_loc_amd import hip
    H, W = 13, 9
    poses = [synth.pose_spherical(40.0 * i - 60.0, -25.0 - 3 * i, 3.5 + 0.2 * i)[:3, :4].astype(F32) for i in range(5)]
    Ks = [np.array([[20.0 + i, 0, 4.5], [0, 21.0 + 2 * i, 6.5], [0, 0, 1]]) for i in range(5)]
    for vd in (True, False):
        table = np.stack([np.concatenate([p.reshape(-1), [K[0][0], K[1][1], K[0][2], K[1][2]]]).astype(F32)
                          for p, K in zip(poses, Ks)])
        got = N(hip.make_rays_views(H, W, T(table, dev), 0.1, 9.0, vd))
        want = np.concatenate([N(hip.make_rays(H, W, K, c2w=T(p, dev), near=0.1, far=9.0, use_viewdirs=vd))
                               for p, K in zip(poses, Ks)], 0)
        assert got.shape == want.shape == (5 * H * W, 11 if vd else 8)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert hip.make_rays_views(H, W, torch.zeros((0, 16), device=dev), 0.1, 9.0, True).shape == (0, 11)
    with pytest.raises(ValueError):
        hip.make_rays_views(H, W, torch.zeros((3, 12), device=dev), 0.1, 9.0, True)


# ------------------------------------------------------------------ sample_pdf
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_sample_pdf_bit_exact(dev, golden, tag):
    from nerf_loc_amd import hip
    g, lin = golden("g1_sample_pdf"), golden("g7_linspace")
    bins, w, ns = g[tag + "_bins"], g[tag + "_weights"], int(g[tag + "_nsamples"])
    np.random.seed(0)
    u_rnd = np.random.rand(bins.shape[0], ns).astype(F32)
    for mode, u in (("det", lin["lin%d" % ns]), ("rnd", u_rnd), ("detpy", lin["nplin%d" % ns])):
        zs, inds, _, _ = hip.sample_pdf(T(bins, dev), T(w, dev), T(u, dev), want_inds=True)
        assert np.array_equal(N(inds).astype(np.int64), g["%s_%s_inds" % (tag, mode)].astype(np.int64)), mode
        assert np.array_equal(N(zs).view(np.uint32), g["%s_%s_samples" % (tag, mode)].view(np.uint32)), mode


def test_sample_pdf_api_and_merge(dev, golden):
    from nerf_loc_amd import helpers, hip
    g = golden("g1_sample_pdf")
    bins, w = g["a_bins"], g["a_weights"]
    s = helpers.sample_pdf(T(bins, dev), T(w, dev), 128, det=True)
    assert maxabs(N(s), g["a_det_samples"]) < 1e-6      # host linspace may differ in the last bit across hosts
    s = helpers.sample_pdf(T(bins, dev), T(w, dev), 128, det=False, pytest=True)
    assert np.array_equal(N(s), g["a_rnd_samples"])
    # fused mid-points + strided weights + sort(cat) + z_std against the oracle, unsorted variates included
    rs = np.random.RandomState(11)
    R, S, NI = 700, 64, 128
    z = np.sort(rs.uniform(2, 6, size=(R, S)), -1).astype(F32)
    wfull = (rs.uniform(0, 1, size=(R, S)) ** 4).astype(F32)
    wfull[::7] = 0
    for u in (orc.linspace01(NI), rs.uniform(0, 1, size=(R, NI)).astype(F32)):
        mids = (F32(0.5) * (z[:, 1:] + z[:, :-1])).astype(F32)
        ref, ref_i = orc.sample_pdf(mids, wfull[:, 1:-1], u, return_inds=True)
        wt = T(wfull, dev)
        zs, inds, zsorted, std = hip.sample_pdf(T(z, dev), wt[:, 1:-1], T(u, dev), bins_are_z=True, z_vals=T(z, dev),
                                                want_inds=True, want_sorted=True, want_std=True)
        assert np.array_equal(N(inds), ref_i)
        assert np.array_equal(N(zs), ref)
        assert np.array_equal(N(zsorted), np.sort(np.concatenate([z, ref], -1), -1))
        assert maxabs(N(std), ref.astype(np.float64).std(-1)) < 2e-6


def test_sample_pdf_row_kernel_cases(dev):
    """The shape render_rays uses (64 depths -> 63 bins, 128 shared variates, merge) on inputs that steer the
    row-per-ray kernel through its side paths: a ragged last group of rays, all-zero weights (flat pdf, every cdf step
    below the 1e-5 guard), one-hot weights, weights of very different magnitude (the float64 running sum is then not
    exact in any order: sequential path), repeated coarse depths (samples equal to mid-points and depths: checked
    merge), and shared variates that are not torch's table: one value off by an ulp (still an ascending near-grid: the
    row kernel compares against the actual values) and a shuffled table (the bisection kernel must take it).
    Everything bit-exact against the oracle."""
    from nerf_loc_amd import hip
    rs = np.random.RandomState(23)
    R, S, NI = 1021, 64, 128
    z = np.sort(rs.uniform(2, 6, size=(R, S)), -1).astype(F32)
    wfull = (rs.uniform(0, 1, size=(R, S)) ** 8).astype(F32)
    wfull[0::11] = 0                                            # flat pdf
    wfull[1::11] = 0
    wfull[1::11, 30] = 1.0                                      # one-hot
    wfull[2::11, 5:9] *= F32(1e12)                              # pdf entries below 2^-28: the exact-sum shortcut must refuse
    z[3::11, 20:24] = z[3::11, 20:21]                           # repeated depths
    z[4::11, :] = z[4::11, :1]                                  # a degenerate ray: one depth 64 times
    wfull[5::11, -2] = 0.0
    wfull[5::11, 1] = F32(3.0)                                  # mass at the near end
    grid = orc.linspace01(NI)
    near_grid = grid.copy()
    near_grid[37] = np.nextafter(near_grid[37], F32(1))          # one variate off by an ulp
    shuffled = grid[rs.permutation(NI)].copy()
    for name, u in (("grid", grid), ("near", near_grid), ("shuffled", shuffled)):
        mids = (F32(0.5) * (z[:, 1:] + z[:, :-1])).astype(F32)
        ref, ref_i = orc.sample_pdf(mids, wfull[:, 1:-1], u, return_inds=True)
        wt = T(wfull, dev)
        zs, inds, zsorted, std = hip.sample_pdf(T(z, dev), wt[:, 1:-1], T(u, dev), bins_are_z=True, z_vals=T(z, dev),
                                                want_inds=True, want_sorted=True, want_std=True)
        assert np.array_equal(N(inds), ref_i), name
        assert np.array_equal(N(zs).view(np.uint32), ref.view(np.uint32)), name
        assert np.array_equal(N(zsorted), np.sort(np.concatenate([z, ref], -1), -1)), name
        assert maxabs(N(std), ref.astype(np.float64).std(-1)) < 2e-6, name
        # without the optional outputs (what render_rays asks for)
        _, _, zsorted2, _ = hip.sample_pdf(T(z, dev), wt[:, 1:-1], T(u, dev), bins_are_z=True, z_vals=T(z, dev),
                                           want_samples=False, want_sorted=True, want_std=False)
        assert np.array_equal(N(zsorted2), N(zsorted)), name
    # non-finite weights: nothing to compare, but the launch has to come back with every depth written
    wbad = wfull.copy()
    wbad[::5, 10] = np.nan
    wbad[1::5, 12] = np.inf
    _, _, zsorted, _ = hip.sample_pdf(T(z, dev), T(wbad, dev)[:, 1:-1], T(grid, dev), bins_are_z=True, z_vals=T(z, dev),
                                      want_samples=False, want_sorted=True, want_std=True)
    torch.cuda.synchronize()
    assert N(zsorted).shape == (R, S + NI)


# ------------------------------------------------------------------ network
def test_mlp_fp32_golden(dev, golden):
    g = golden("g4_mlp")
    feat = T(g["feat"], dev)
    for regime in "AB":
        net = load_net(synth.make_state_dict(40, regime), dev)
        out = N(net(feat))
        ref = g["out_vd_" + regime]
        assert maxabs(out, ref) < 2e-5 * max(1.0, np.abs(ref).max()), (regime, maxabs(out, ref))
    n5 = load_net(synth.make_state_dict(41, "A", use_viewdirs=False, output_ch=5, input_ch_views=0), dev, False, 5)
    assert maxabs(N(n5(feat[:, :63].contiguous())), g["out_novd5_A"]) < 2e-5
    n4 = load_net(synth.make_state_dict(42, "B", use_viewdirs=False, output_ch=4, input_ch_views=0), dev, False, 4)
    ref = g["out_novd4_B"]
    assert maxabs(N(n4(feat[:, :63].contiguous())), ref) < 2e-5 * max(1.0, np.abs(ref).max())


def test_other_network_shapes_compose(dev):
    """--netdepth / --netwidth other than 8 / 256 (run_nerf.py:440-446): no fused kernel, the layers run as library
    GEMMs; forward against the oracle, a render through the generic path, and gradients through torch's autograd."""
    from nerf_loc_amd import helpers, render
    arch = dict(D=4, W=128, skips=(2,))
    sd = synth.make_state_dict(7, "B", **arch)
    net = helpers.NeRF(D=4, W=128, input_ch=63, output_ch=4, skips=[2], input_ch_views=27, use_viewdirs=True)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net = net.to(dev)
    assert not net.hip_supported()
    rs = np.random.RandomState(2)
    x = rs.uniform(-2, 2, size=(300, 3)).astype(F32)
    d = rs.normal(size=(300, 3)).astype(F32)
    feat = np.concatenate([orc.embed(x, 10), orc.embed(d, 4)], -1)
    ref = orc.mlp_forward(sd, feat, D=4, skips=(2,))
    with torch.no_grad():
        out = N(net(T(feat, dev)))
    assert maxabs(out, ref) < 2e-5 * max(1.0, np.abs(ref).max())
    rays = synth.random_rays(64, seed=4)
    with torch.no_grad():
        r = render.render_rays(T(rays, dev), net, _query(dev), 32, white_bkgd=True)
    o = orc.render_rays(rays, sd, 32, 0, white_bkgd=True, mlp=lambda s, f, vd: orc.mlp_forward(s, f, vd, D=4, skips=(2,)))
    assert maxabs(N(r["rgb_map"]), o["rgb_map"]) < 1e-4 and maxabs(N(r["acc_map"]), o["acc_map"]) < 1e-4
    r = render.render_rays(T(rays, dev), net, _query(dev), 32, white_bkgd=True)
    r["rgb_map"].sum().backward()
    assert net.pts_linears[0].weight.grad is not None and float(net.pts_linears[0].weight.grad.abs().max()) > 0


def test_mlp_run_network_modes_agree(dev, golden):
    """Features, points and rays input modes of the kernel give the same raw."""
    from nerf_loc_amd import helpers, render, hip
    g = golden("g4_mlp")
    x, d = g["x"], g["d"]
    net = load_net(synth.make_state_dict(40, "A"), dev)
    e10, _ = helpers.get_embedder(10, 0)
    e4, _ = helpers.get_embedder(4, 0)
    R, S = 64, 8
    pts = T(x, dev).reshape(R, S, 3)
    vd = T(d[::S].copy(), dev)
    out = render.run_network(pts, vd, net, e10, e4)
    assert out.shape == (R, S, 4)
    feat = np.concatenate([orc.embed(x, 10), orc.embed(np.repeat(d[::S], S, 0), 4)], -1)
    ref2 = orc.mlp_forward(synth.make_state_dict(40, "A"), feat)
    assert maxabs(N(out).reshape(-1, 4), ref2) < 2e-5
    # generic composition through a foreign callable goes through the same ops
    out_generic = render.run_network(pts, vd, lambda f: net(f), e10, e4, netchunk=100)
    assert maxabs(N(out_generic), N(out)) < 2e-5


def test_mlp_bf16_close_to_fp32(dev, golden):
    g = golden("g4_mlp")
    feat = T(g["feat"], dev)
    for regime in "AB":
        net = load_net(synth.make_state_dict(40, regime), dev, precision="bf16")
        out = N(net(feat))
        ref = g["out_vd_" + regime]
        err = np.abs(out - ref)
        scale = np.abs(ref).max()
        print("bf16 raw regime %s: max %.3e mean %.3e (scale %.2f)" % (regime, err.max(), err.mean(), scale))
        # measured 4.9e-4 (scale 0.2) and 1.6e-2 (scale 2.4, i.e. 6.6e-3 of it): bounds are twice that
        assert err.max() < 1.4e-2 * max(1.0, scale) and err.mean() < 2.5e-3 * max(1.0, scale)


def test_mlp_ragged_sizes(dev):
    """Sample counts that do not fill a tile, and more tiles than CUs."""
    from nerf_loc_amd import hip
    sd = synth.make_state_dict(44, "A")
    rs = np.random.RandomState(5)
    for prec in ("fp32", "bf16", "bf16x3", "bf16x2"):
        net = load_net(sd, dev, precision=prec)
        for n in (1, 31, 129, 257, 70000 if prec == "bf16" else 33000):
            x = rs.uniform(-3, 3, size=(n, 3)).astype(F32)
            d = rs.normal(size=(n, 3))
            d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(F32)
            feat = np.concatenate([orc.embed(x, 10), orc.embed(d, 4)], -1)
            out = N(net(T(feat, dev)))
            idx = np.unique(np.concatenate([np.arange(min(n, 300)), np.arange(max(0, n - 300), n)]))
            ref = orc.mlp_forward(sd, feat[idx])
            tol = {"fp32": 2e-5, "bf16x3": 2e-5, "bf16x2": 3e-4, "bf16": 3e-2}[prec]
            assert maxabs(out[idx], ref) < tol * max(1.0, np.abs(ref).max()), (prec, n)
            assert np.isfinite(out).all()


# ------------------------------------------------------------------ render_rays
def _query(dev):
    from nerf_loc_amd import helpers, render
    e10, _ = helpers.get_embedder(10, 0)
    e4, _ = helpers.get_embedder(4, 0)
    return render.NetworkQuery(e10, e4, 1024 * 64)


@pytest.mark.parametrize("regime", ["A", "B"])
def test_render_rays_fp32_stagewise(dev, golden, regime):
    """Coarse pass, importance sampling and fine pass, each compared on identical inputs."""
    from nerf_loc_amd import render
    g = golden("g5_render_rays")
    rays = g["rays"]
    sc, sf = synth.make_state_dict(50, regime), synth.make_state_dict(51, regime)
    nc, nf = load_net(sc, dev), load_net(sf, dev)
    r = render.render_rays(T(rays, dev), nc, _query(dev), 64, retraw=True, N_importance=128, network_fine=nf,
                           white_bkgd=True)
    r = {k: N(v) for k, v in r.items()}
    tag = "cf_" + regime
    # Conditioning mask from the reference's own float64 rerun (never from the output under test): the last interval
    # is 1e10 long, so a ray whose last density is within float32 rounding of the ReLU kink can flip its last alpha.
    raw_last = g["cf64_%s_rawlast" % regime]
    mask = np.abs(raw_last) > 1e-4
    print("regime %s: %d of %d rays masked as ill-conditioned (|last sigma| of the float64 run <= 1e-4)"
          % (regime, (~mask).sum(), mask.size))
    assert (~mask).sum() == 0            # the fixture's rays are all well conditioned: nothing is excluded below
    for k in ("rgb0", "acc0", "disp0"):
        e = maxabs(r[k], g["%s_%s" % (tag, k)])
        print("  coarse %s max abs err %.3e" % (k, e))
        assert e < 1e-4, k
    # free-running fine pass (importance samples drawn from the GPU's own coarse weights) against the reference run:
    # the north star's 1e-4 absolute on every ray
    for k in ("rgb_map", "acc_map", "disp_map"):
        e_all = maxabs(r[k], g["%s_%s" % (tag, k)])
        print("  fine %s max abs err on all %d rays %.3e" % (k, mask.size, e_all))
        assert e_all < 1e-4, k
    # z_std is not one of the contract maps (rgb, depth, acc); the reference's own float32 result is 1e-4 from its
    # float64 rerun.  Measured 9.0e-5 (A) and 2.6e-5 (B): the bound is twice the larger.
    e = maxabs(r["z_std"], g[tag + "_z_std"])
    e64 = maxabs(g[tag + "_z_std"], g["cf64_%s_z_std" % regime].astype(np.float32))
    print("  z_std max abs err %.3e (the reference's float32 vs its float64 run: %.3e)" % (e, e64))
    assert e < 2e-4


@pytest.mark.parametrize("regime", ["A", "B"])
def test_render_rays_fp32_teacher_forced(dev, golden, regime):
    """The 1e-4 bar of the north star, each pass fed identical inputs: the GPU's own coarse weights go
    through the oracle's sample_pdf (bit-exact), then the fine pass is compared on identical depths."""
    from nerf_loc_amd import hip, render
    g = golden("g5_render_rays")
    rays = g["rays"]
    sc, sf = synth.make_state_dict(50, regime), synth.make_state_dict(51, regime)
    nc, nf = load_net(sc, dev), load_net(sf, dev)
    rt = T(rays, dev)
    z = hip.z_vals(rt, 64)
    assert np.array_equal(N(z), orc.z_vals_for(rays, 64))
    raw = render.run_network(rt[:, None, 0:3] + rt[:, None, 3:6] * z[:, :, None], rt[:, 8:11].contiguous(), nc,
                             *_embedders())
    rgb0, _, acc0, w0, depth0 = render.raw2outputs(raw, z, rt[:, 3:6], white_bkgd=True)
    o = orc.render_rays(rays, sc, 64, 0, white_bkgd=True, retraw=True)
    scale = max(1.0, np.abs(o["raw"]).max())
    print("regime %s coarse raw err %.3e (scale %.1f)" % (regime, maxabs(N(raw), o["raw"]), scale))
    assert maxabs(N(raw), o["raw"]) < 2e-5 * scale
    assert maxabs(N(rgb0), o["rgb_map"]) < 1e-4 and maxabs(N(acc0), o["acc_map"]) < 1e-4
    e = maxabs(N(depth0), o["depth_map"])
    print("  coarse depth max abs err %.3e (depths 2..6)" % e)
    assert e < 1e-4
    # importance sampling on the GPU's weights: bit-exact against the oracle on the same weights
    u = orc.linspace01(128)
    w_np, z_np = N(w0), N(z)
    mids = (F32(0.5) * (z_np[:, 1:] + z_np[:, :-1])).astype(F32)
    zs_ref, inds_ref = orc.sample_pdf(mids, w_np[:, 1:-1], u, return_inds=True)
    zs, inds, zsorted, std = hip.sample_pdf(z, w0[:, 1:-1], T(u, dev), bins_are_z=True, z_vals=z, want_inds=True,
                                            want_sorted=True, want_std=True)
    assert np.array_equal(N(inds), inds_ref) and np.array_equal(N(zs), zs_ref)
    assert maxabs(N(std), zs_ref.std(-1)) < 1e-5      # torch.std(unbiased=False), run_nerf.py:418
    # fine pass on identical depths
    zf = N(zsorted)
    pts = rays[:, None, 0:3] + rays[:, None, 3:6] * zf[:, :, None]
    raw_ref = orc.run_network(sf, pts, rays[:, 8:11])
    raw_f = render.run_network(T(pts, dev), rt[:, 8:11].contiguous(), nf, *_embedders())
    scale = max(1.0, np.abs(raw_ref).max())
    print("regime %s fine raw err %.3e (scale %.1f)" % (regime, maxabs(N(raw_f), raw_ref), scale))
    assert maxabs(N(raw_f), raw_ref) < 2e-5 * scale
    ref = orc.raw2outputs(raw_ref, zf, rays[:, 3:6], white_bkgd=True)
    got = render.raw2outputs(raw_f, zsorted, rt[:, 3:6], white_bkgd=True)
    sens = np.abs(raw_ref[:, -1, 3]) > 1e-3        # alpha of the last (1e10 long) interval flips at sigma = 0
    for a, b, name in zip(got, ref, ("rgb", "disp", "acc", "weights", "depth")):
        if name == "disp":
            continue
        e = maxabs(N(a)[sens], b[sens])
        print("  fine %s max abs err %.3e (%d rays excluded: last sigma within 1e-3 of the relu kink)"
              % (name, e, (~sens).sum()))
        assert e < 1e-4, name


def _embedders():
    from nerf_loc_amd import helpers
    return helpers.get_embedder(10, 0)[0], helpers.get_embedder(4, 0)[0]


def test_render_rays_variants(dev, golden):
    from nerf_loc_amd import render
    g = golden("g5_render_rays")
    rays = g["rays"]
    sc, sf = synth.make_state_dict(50, "B"), synth.make_state_dict(51, "B")
    nc, nf = load_net(sc, dev), load_net(sf, dev)
    q = _query(dev)
    # coarse only, raw returned
    r = render.render_rays(T(rays, dev), nc, q, 64, retraw=True, white_bkgd=True)
    assert set(r) == {"rgb_map", "disp_map", "acc_map", "raw"}
    assert maxabs(N(r["rgb_map"]), g["c64_B_rgb_map"]) < 1e-4
    assert maxabs(N(r["raw"])[:32], g["c64_B_raw"]) < 2e-5 * np.abs(g["c64_B_raw"]).max()
    # lindisp + shared network for the fine pass
    r = render.render_rays(T(rays[:64], dev), nc, q, 64, N_importance=64, lindisp=True)
    assert set(r) == {"rgb_map", "disp_map", "acc_map", "rgb0", "disp0", "acc0", "z_std"}
    assert maxabs(N(r["rgb0"]), g["lindisp_B_rgb0"]) < 1e-4
    assert maxabs(N(r["disp0"]), g["lindisp_B_disp0"]) < 1e-4
    assert r["rgb_map"].shape == (64, 3) and r["z_std"].shape == (64,)
    # no view directions, 5-channel head, 8-column rays
    from nerf_loc_amd import helpers
    s5c = synth.make_state_dict(52, "B", use_viewdirs=False, output_ch=5, input_ch_views=0)
    s5f = synth.make_state_dict(53, "B", use_viewdirs=False, output_ch=5, input_ch_views=0)
    n5c, n5f = load_net(s5c, dev, False, 5), load_net(s5f, dev, False, 5)
    q5 = render.NetworkQuery(helpers.get_embedder(10, 0)[0], None, 65536)
    r = render.render_rays(T(rays[:64, :8].copy(), dev), n5c, q5, 64, retraw=True, N_importance=128, network_fine=n5f,
                           white_bkgd=True)
    assert r["raw"].shape == (64, 192, 5)
    assert maxabs(N(r["rgb0"]), g["novd_B_rgb0"]) < 1e-4 and maxabs(N(r["acc0"]), g["novd_B_acc0"]) < 1e-4
    # stochastic branches through the numpy hooks (pytest=True): coarse pass is exactly reproducible
    r = render.render_rays(T(rays[:64], dev), nc, q, 64, N_importance=128, network_fine=nf, white_bkgd=True,
                           perturb=1.0, raw_noise_std=1.0, pytest=True)
    assert maxabs(N(r["rgb0"]), g["stoch_B_rgb0"]) < 1e-4 and maxabs(N(r["acc0"]), g["stoch_B_acc0"]) < 1e-4
    # stochastic branches with the device RNG: statistically sane, different from the deterministic render
    torch.manual_seed(0)
    r2 = render.render_rays(T(rays[:64], dev), nc, q, 64, N_importance=128, network_fine=nf, white_bkgd=True,
                            perturb=1.0, raw_noise_std=0.0)
    assert np.isfinite(N(r2["rgb_map"])).all() and maxabs(N(r2["rgb0"]), g["cf_B_rgb0"][:64]) > 1e-6


def test_render_image_both_flavours(dev, golden):
    from nerf_loc_amd import render, nerf_utils
    import types
    g = golden("g9_render")
    H, W = [int(v) for v in g["lego_hw"]]
    sc, sf = synth.make_state_dict(90, "B"), synth.make_state_dict(91, "B")
    nc, nf = load_net(sc, dev), load_net(sf, dev)
    kw = dict(network_query_fn=_query(dev), perturb=0.0, N_importance=128, network_fine=nf, N_samples=64,
              network_fn=nc, use_viewdirs=True, white_bkgd=True, raw_noise_std=0.0, ndc=False, lindisp=False,
              near=2.0, far=6.0)
    rgb, disp, acc, extras = render.render(H, W, g["lego_K"], chunk=150, c2w=T(g["lego_c2w"], dev), **kw)
    assert rgb.shape == (H, W, 3) and disp.shape == (H, W) and acc.shape == (H, W)
    assert set(extras) == {"rgb0", "disp0", "acc0", "z_std"}
    assert maxabs(N(extras["rgb0"]), g["lego_rgb0"]) < 1e-4 and maxabs(N(extras["acc0"]), g["lego_acc0"]) < 1e-4
    e = maxabs(N(rgb), g["lego_rgb"])
    print("render() 20x20 coarse+fine free-running rgb err %.3e" % e)
    assert e < 1e-4 and maxabs(N(acc), g["lego_acc"]) < 1e-4      # free-running importance samples, every pixel
    # rays= entry: [2, N, 3]
    rays = g["batch_rays"]
    ro, rd = T(rays[:, :3].copy(), dev), T(rays[:, 3:6].copy(), dev)
    rgb, disp, acc, extras = render.render(H, W, g["lego_K"], chunk=1024, rays=torch.stack([ro, rd]), **kw)
    assert rgb.shape == (96, 3)
    assert maxabs(N(extras["rgb0"]), g["batch_rgb0"]) < 1e-4
    # NeRF-Loc flavour
    args = types.SimpleNamespace(nerf_render_rgb=True, nerf_render_depth=True)
    _, kw_test, _ = nerf_utils.create_nerf_model(args, ckpt_path=None, device=dev, near=0.1, far=9.0)
    kw_test["network_fn"].load_state_dict({k: torch.from_numpy(v) for k, v in sc.items()})
    kw_test["network_fn"].requires_grad_(False)
    H, W = [int(v) for v in g["loc_hw"]]
    rgb, disp, acc, depth, extras = nerf_utils.render(H, W, g["loc_K"], chunk=1024 * 32, c2w=T(g["loc_c2w"], dev),
                                                      **kw_test)
    assert maxabs(N(rgb), g["loc_rgb"]) < 1e-4 and maxabs(N(acc), g["loc_acc"]) < 1e-4
    e = maxabs(N(depth), g["loc_depth"])
    print("NeRF-Loc render() depth max abs err %.3e (depths 0.1..9)" % e)
    assert e < 1e-4
    assert maxabs(N(extras["raw"]), g["loc_raw"]) < 2e-5 * max(1.0, np.abs(g["loc_raw"]).max())
    poses = torch.stack([T(g["loc_c2w"], dev)] * 2)
    rgbs, _, raws, depths = nerf_utils.render_path(poses, [H, W, float(g["loc_K"][0, 0])], g["loc_K"], 1024 * 32,
                                                       kw_test)
    assert rgbs.shape == (2, H, W, 3) and raws.shape == (2, H, W, 64, 4) and depths.shape == (2, H, W)
    assert isinstance(rgbs, np.ndarray) and maxabs(rgbs[1], g["loc_rgb"]) < 1e-4
    # all poses as one ray batch, results on the device: same numbers as pose by pose
    d_rgbs, d_disps, d_raws, d_depths = nerf_utils.render_path_device(poses, [H, W, float(g["loc_K"][0, 0])], g["loc_K"],
                                                                       kw_test)
    assert d_rgbs.is_cuda and tuple(d_raws.shape) == (2, H, W, 64, 4)
    assert np.array_equal(N(d_rgbs), rgbs) and np.array_equal(N(d_depths), depths) and np.array_equal(N(d_raws), raws)


def test_render_path_writes_frames(dev, golden, tmp_path):
    """render_path(savedir=...) leaves '{:06d}.png' (the 8-bit image of the returned rgb) and, with retraw,
    '{:06d}.npy' of the raw volume per pose (run_nerf.py:170-175)."""
    from nerf_loc_amd import render, poses as P
    g = golden("g9_render")
    H, W = [int(v) for v in g["lego_hw"]]
    nc = load_net(synth.make_state_dict(90, "B"), dev)
    kw = dict(network_query_fn=_query(dev), perturb=0.0, N_importance=0, network_fine=None, N_samples=64,
              network_fn=nc, use_viewdirs=True, white_bkgd=True, raw_noise_std=0.0, ndc=False, lindisp=False,
              near=2.0, far=6.0, retraw=True)
    c2w = T(g["lego_c2w"], dev)
    rgbs, disps = render.render_path(torch.stack([c2w, c2w]), [H, W, float(g["lego_K"][0, 0])], g["lego_K"], 1024, kw,
                                     savedir=str(tmp_path))
    assert rgbs.shape == (2, H, W, 3)
    assert sorted(os.listdir(tmp_path)) == ["000000.npy", "000000.png", "000001.npy", "000001.png"]
    for i in range(2):
        img = P.read_png(str(tmp_path / ("%06d.png" % i)))
        assert np.array_equal(img, (255 * np.clip(rgbs[i], 0, 1)).astype(np.uint8))
        assert np.load(str(tmp_path / ("%06d.npy" % i))).shape == (H, W, 64, 4)


# ------------------------------------------------------------------ bf16 mode end to end
def _psnr(a, b):
    return -10.0 * np.log10(float(((a.astype(np.float64) - b) ** 2).mean()))


# |last raw density| below this is within the bf16 rounding of the network output: three times the measured worst
# error of the raw output (4.9e-4 for plain random weights, raw values of magnitude 0.2; 1.6e-2 for the "trained-like"
# weights, magnitude 2.4; see test_mlp_bf16_close_to_fp32)
BF16_KINK = {"A": 1.5e-3, "B": 0.05}


@pytest.mark.parametrize("regime", ["A", "B"])
def test_render_bf16_vs_reference_golden(dev, golden, regime):
    """bf16-MFMA mode against the REFERENCE's float32 outputs (g5: render_rays of nerf-pytorch on these rays and
    weights): rgb / acc within 1e-2 absolute, PSNR against a common 30 dB target within 0.1 dB.  Rays are excluded
    only by the reference's own float64 rerun (last density within bf16 rounding of the ReLU kink, where the 1e10-long
    last interval switches on or off); the count is asserted."""
    from nerf_loc_amd import render
    g = golden("g5_render_rays")
    rays = g["rays"]
    nc = load_net(synth.make_state_dict(50, regime), dev, precision="bf16")
    nf = load_net(synth.make_state_dict(51, regime), dev, precision="bf16")
    r = render.render_rays(T(rays, dev), nc, _query(dev), 64, N_importance=128, network_fine=nf, white_bkgd=True)
    r = {k: N(v) for k, v in r.items()}
    cond = np.abs(g["cf64_%s_rawlast" % regime]) > BF16_KINK[regime]
    n_masked = int((~cond).sum())
    ref_rgb, ref_acc = g["cf_%s_rgb_map" % regime], g["cf_%s_acc_map" % regime]
    err = np.abs(r["rgb_map"] - ref_rgb).max(-1)
    e_acc = np.abs(r["acc_map"] - ref_acc)
    np.random.seed(5)
    target = ref_rgb.astype(np.float64) + np.random.randn(*ref_rgb.shape) * 10 ** (-30 / 20)
    dpsnr = abs(_psnr(r["rgb_map"][cond], target[cond]) - _psnr(ref_rgb[cond], target[cond]))
    print("regime %s bf16 vs reference: rgb max %.3e, acc max %.3e on %d conditioned rays (%d masked by the float64 run; "
          "all rays rgb %.3e), coarse rgb0 max %.3e, PSNR shift at 30 dB %.4f dB"
          % (regime, err[cond].max(), e_acc[cond].max(), cond.sum(), n_masked, err.max(),
             np.abs(r["rgb0"] - g["cf_%s_rgb0" % regime]).max(), dpsnr))
    assert n_masked == {"A": 5, "B": 0}[regime]
    assert err[cond].max() < 1e-2 and e_acc[cond].max() < 1e-2
    assert dpsnr < 0.1


def test_render_4096_view_rays_vs_reference(dev, golden):
    """4096 pixels of the benchmark's own 800x800 view with the benchmark's networks, against the reference's float32
    render of the same rays (g12): float32 mode within 1e-4 absolute on every map, bf16 mode within 1e-2 and 0.1 dB.
    Rays whose last density lies within rounding of the ReLU kink are taken out by the reference's float64 rerun
    (float32: where float32 and float64 reference disagree about the sign; bf16: |raw| below the bf16 error of the
    density output), and counted."""
    from nerf_loc_amd import render
    g = golden("g12_view_rays")
    rays, rawlast = g["rays"], g["f64_rawlast"]
    sc, sf = synth.make_state_dict(0, "B"), synth.make_state_dict(1, "B")
    res = {}
    for prec in ("fp32", "bf16"):
        nc, nf = load_net(sc, dev, precision=prec), load_net(sf, dev, precision=prec)
        r = render.render_rays(T(rays, dev), nc, _query(dev), 64, N_importance=128, network_fine=nf, white_bkgd=True)
        res[prec] = {k: N(v) for k, v in r.items()}
    # float32: every ray whose float64 last density is not within 1e-4 of zero
    cond = np.abs(rawlast) > 1e-4
    r = res["fp32"]
    errs = {k: np.abs(r[k] - g["f32_" + k]).reshape(len(rays), -1).max(-1) for k in ("rgb_map", "acc_map", "rgb0", "acc0")}
    finite = np.isfinite(g["f32_disp_map"]) & np.isfinite(r["disp_map"])
    e_disp = np.abs(r["disp_map"] - g["f32_disp_map"])[finite & cond]
    print("fp32 vs reference on 4096 view rays: rgb max %.2e (median %.1e), acc max %.2e, coarse rgb0 max %.2e, disp max %.2e, "
          "z_std max %.2e; %d rays masked by the float64 run"
          % (errs["rgb_map"][cond].max(), np.median(errs["rgb_map"]), errs["acc_map"][cond].max(), errs["rgb0"].max(),
             e_disp.max(), np.abs(r["z_std"] - g["f32_z_std"])[cond].max(), int((~cond).sum())))
    assert (~cond).sum() <= 8
    for k in errs:
        assert errs[k][cond].max() < 1e-4, k
    assert np.abs(r["z_std"] - g["f32_z_std"])[cond].max() < 2e-4
    # bf16
    cond16 = np.abs(rawlast) > BF16_KINK["B"]
    b = res["bf16"]
    e_rgb = np.abs(b["rgb_map"] - g["f32_rgb_map"]).max(-1)
    e_acc = np.abs(b["acc_map"] - g["f32_acc_map"])
    np.random.seed(12)
    ref_rgb = g["f32_rgb_map"]
    target = ref_rgb.astype(np.float64) + np.random.randn(*ref_rgb.shape) * 10 ** (-30 / 20)
    dpsnr = abs(_psnr(b["rgb_map"][cond16], target[cond16]) - _psnr(ref_rgb[cond16], target[cond16]))
    print("bf16 vs reference on 4096 view rays: rgb max %.2e (99.9th percentile %.1e, median %.1e), acc max %.2e on %d conditioned "
          "rays (%d masked); PSNR shift at 30 dB %.4f dB"
          % (e_rgb[cond16].max(), np.percentile(e_rgb[cond16], 99.9), np.median(e_rgb), e_acc[cond16].max(), cond16.sum(),
             int((~cond16).sum()), dpsnr))
    assert (~cond16).sum() <= 0.02 * len(rays)
    assert e_rgb[cond16].max() < 1e-2 and e_acc[cond16].max() < 1e-2 and dpsnr < 0.1


def test_loc_image_240x180_vs_reference(dev, golden):
    """NeRF-Loc's render() at its real image size (g13: one 240 x 180 Objectron view through the reference's
    nerf_utils.render, 64 samples, near 0.1, far 9): rgb, acc and DEPTH within 1e-4 absolute in float32 mode, disp where
    it is finite; bf16 within 1e-2 on rgb / acc."""
    from nerf_loc_amd import nerf_utils
    import types
    g = golden("g13_loc_image")
    H, W = [int(v) for v in g["hw"]]
    idx = g["idx"]
    args = types.SimpleNamespace(nerf_render_rgb=True, nerf_render_depth=True)
    for prec in ("fp32", "bf16"):
        _, kw, _ = nerf_utils.create_nerf_model(args, ckpt_path=None, device=dev, near=0.1, far=9.0)
        kw["network_fn"].load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(100, "B").items()})
        kw["network_fn"].requires_grad_(False)
        kw["network_fn"].precision = prec
        rgb, disp, acc, depth, extras = nerf_utils.render(H, W, g["K"], chunk=1024 * 32, c2w=T(g["c2w"], dev), **kw)
        pick = lambda v: N(v).reshape(H * W, -1)[idx]
        # the last interval is 1e10 long: where the last density is within rounding of zero the ray may flip
        thr = 1e-4 if prec == "fp32" else BF16_KINK["B"]
        cond = np.abs(g["raw_sigma_last"]) > thr
        e_rgb = np.abs(pick(rgb) - g["rgb"]).max(-1)
        e_acc = np.abs(pick(acc)[:, 0] - g["acc"])
        e_depth = np.abs(pick(depth)[:, 0] - g["depth"])
        fin = np.isfinite(g["disp"]) & np.isfinite(pick(disp)[:, 0]) & cond
        e_disp = np.abs(pick(disp)[:, 0] - g["disp"])[fin]
        print("%s NeRF-Loc 240x180 vs reference (%d of %d pixels, %d masked): rgb max %.2e, acc max %.2e, depth max %.2e "
              "(depths up to %.1f), disp max %.2e" % (prec, cond.sum(), len(idx), int((~cond).sum()), e_rgb[cond].max(),
                                                      e_acc[cond].max(), e_depth[cond].max(), g["depth"].max(), e_disp.max()))
        assert (~cond).sum() <= (4 if prec == "fp32" else 0.02 * len(idx))
        if prec == "fp32":
            assert e_rgb[cond].max() < 1e-4 and e_acc[cond].max() < 1e-4 and e_depth[cond].max() < 1e-4
            assert tuple(extras["raw"].shape) == (H, W, 64, 4)
        else:
            assert e_rgb[cond].max() < 1e-2 and e_acc[cond].max() < 1e-2


_FOG = pytest.mark.xfail(strict=True, reason="known violation of the 1e-2 bar: rays through 4 units of barely positive "
                         "density (acc 0.03-0.1) sum the bf16 rounding of 192 near-zero densities coherently; measured "
                         "rgb max 3.6e-2 (PSNR shift 0.03 dB, inside its bar)")


@pytest.mark.parametrize("seed,kind", [(62, "opaque"), (66, "mixed"), pytest.param(64, "fog", marks=_FOG)])
def test_render_bf16_vs_oracle_scenes(dev, seed, kind):
    """bf16-MFMA mode on three synthetic scenes against the numpy oracle (the restatement pinned to the reference
    by the golden fixtures) in float32; the conditioning mask comes from the oracle's float64 run of the same rays,
    never from the output under test.  Bars: 1e-2 absolute on rgb and acc, 0.1 dB PSNR.  The opaque scene (every ray
    ends in dense matter) must hold them on every ray."""
    from nerf_loc_amd import render
    rays = synth.random_rays(2048, seed=21)
    sc, sf = synth.make_state_dict(seed, "B"), synth.make_state_dict(seed + 1, "B")
    nc, nf = load_net(sc, dev, precision="bf16"), load_net(sf, dev, precision="bf16")
    b = render.render_rays(T(rays, dev), nc, _query(dev), 64, retraw=True, N_importance=128, network_fine=nf,
                           white_bkgd=True)
    b = {k: N(v) for k, v in b.items()}
    f = orc.render_rays(rays, sc, 64, 128, sd_fine=sf, white_bkgd=True, retraw=True)
    f64 = orc.render_rays(rays.astype(np.float64), sc, 64, 128, sd_fine=sf, white_bkgd=True, retraw=True)
    cond = np.abs(f64["raw"][:, -1, 3]) > BF16_KINK["B"]
    assert kind != "opaque" or cond.all()
    err = np.abs(b["rgb_map"] - f["rgb_map"])
    np.random.seed(5)
    target = f["rgb_map"].astype(np.float64) + np.random.randn(*f["rgb_map"].shape) * 10 ** (-30 / 20)
    dpsnr = abs(_psnr(b["rgb_map"][cond], target[cond]) - _psnr(f["rgb_map"][cond], target[cond]))
    print("seed %d (%s) bf16 vs oracle: rgb max %.3e on %d conditioned rays (%d masked by the float64 run; all rays "
          "%.3e), mean %.3e, acc max %.3e, PSNR shift at 30 dB %.4f dB"
          % (seed, kind, err[cond].max(), cond.sum(), (~cond).sum(), err.max(), err.mean(),
             np.abs(b["acc_map"] - f["acc_map"])[cond].max(), dpsnr))
    worst = np.argsort(err.max(-1) * cond)[-3:]
    print("  worst conditioned rays: err %s, last raw density float64 %s bf16 %s, acc %s"
          % (err.max(-1)[worst], f64["raw"][worst, -1, 3], b["raw"][worst, -1, 3], f["acc_map"][worst]))
    assert cond.sum() > 0.5 * cond.size
    assert dpsnr < 0.1
    assert err[cond].max() < 1e-2
    assert maxabs(b["acc_map"][cond], f["acc_map"][cond]) < 1e-2


# ------------------------------------------------------------------ edge cases
def test_empty_and_single_ray_batches(dev):
    """Zero rays is a valid call everywhere (the last chunk of a batchify loop can be empty in caller code), and a
    single ray exercises every partially filled tile."""
    from nerf_loc_amd import hip, render
    net = load_net(synth.make_state_dict(3, "B"), dev)
    f = dict(dtype=torch.float32, device=dev)
    e = torch.empty((0, 11), **f)
    assert hip.z_vals(e, 64).shape == (0, 64)
    assert hip.embed(torch.empty((0, 3), **f), 10).shape == (0, 63)
    raw = hip.mlp_forward(net.packed_weights(0), 0, True, 4, hip.IN_RAYS, e, torch.empty((0, 64), **f), 0, 64, 11)
    assert raw.shape == (0, 4)
    rgb, disp, acc, w, depth = hip.raw2outputs(torch.empty((0, 64, 4), **f), torch.empty((0, 64), **f), e[:, 3:6], 11)
    assert rgb.shape == (0, 3) and w.shape == (0, 64)
    zs, _, zsorted, std = hip.sample_pdf(torch.empty((0, 64), **f), torch.empty((0, 62), **f), hip.linspace01(128, dev),
                                         bins_are_z=True, z_vals=torch.empty((0, 64), **f), want_sorted=True, want_std=True)
    assert zs.shape == (0, 128) and zsorted.shape == (0, 192) and std.shape == (0,)
    r = render.render_rays(e, net, _query(dev), 64, N_importance=128, network_fine=net, white_bkgd=True, retraw=True)
    assert r["rgb_map"].shape == (0, 3) and r["raw"].shape == (0, 192, 4)
    rays = synth.random_rays(1, seed=9)
    one = render.render_rays(T(rays, dev), net, _query(dev), 64, N_importance=128, network_fine=net, white_bkgd=True)
    o = orc.render_rays(rays, synth.make_state_dict(3, "B"), 64, 128, sd_fine=synth.make_state_dict(3, "B"), white_bkgd=True)
    assert maxabs(N(one["rgb0"]), o["rgb0"]) < 1e-4 and maxabs(N(one["acc0"]), o["acc0"]) < 1e-4


def test_chunking_does_not_change_results(dev):
    """batchify_rays / netchunk only bound memory: any chunking gives bit-identical maps (run_nerf.py:54-66)."""
    from nerf_loc_amd import render
    rays = T(synth.random_rays(1000, seed=12), dev)
    nc, nf = load_net(synth.make_state_dict(5, "B"), dev), load_net(synth.make_state_dict(6, "B"), dev)
    kw = dict(network_fn=nc, network_query_fn=_query(dev), N_samples=64, N_importance=128, network_fine=nf,
              white_bkgd=True, retraw=True)
    whole = render.batchify_rays(rays, 1 << 20, **kw)
    parts = render.batchify_rays(rays, 77, **kw)
    for k in whole:
        assert torch.equal(whole[k], parts[k]) or (torch.isnan(whole[k]) == torch.isnan(parts[k])).all() and \
            torch.equal(torch.nan_to_num(whole[k]), torch.nan_to_num(parts[k])), k


def test_invalid_arguments_raise(dev):
    from nerf_loc_amd import hip, render
    net = load_net(synth.make_state_dict(3, "B"), dev)
    rays = T(synth.random_rays(8, seed=1), dev)
    with pytest.raises(ValueError):
        render.render_rays(rays[:, :5], net, _query(dev), 64)
    with pytest.raises(ValueError):
        hip.raw2outputs(torch.zeros(4, 8, 4, device=dev), torch.zeros(4, 9, device=dev), rays[:, 3:6], 11)
    with pytest.raises(ValueError):
        hip.sample_pdf(torch.zeros(4, 9, device=dev), torch.zeros(4, 9, device=dev), hip.linspace01(4, dev))
    with pytest.raises(hip.NerfHipError):
        hip.mlp_forward(net.packed_weights(0), 0, True, 4, hip.IN_RAYS, rays[:, :8].contiguous(),
                        torch.zeros(8, 4, device=dev), 8, 4, 8)          # view directions missing from the ray rows
    with pytest.raises(ValueError):
        hip.z_vals(rays.double(), 8)
    # ray directions are passed as a strided view, never copied: a host tensor, a short batch or a wrong stride must
    # be refused before a kernel gets the pointer
    raw, z = torch.zeros(8, 4, 4, device=dev), torch.zeros(8, 4, device=dev)
    for bad in (rays[:, 3:6].cpu(), rays[:4, 3:6], rays[:, 3:6].double(), rays[:, 3:6].t().contiguous().t()):
        with pytest.raises(ValueError):
            hip.raw2outputs(raw, z, bad, 11)
    with pytest.raises(ValueError):
        render.raw2outputs(raw, z, rays[:, 3:6].cpu())
    assert hip.raw2outputs(raw, z, rays[:, 3:6], 11)[0].shape == (8, 3)
    # operand sizes that do not cover what the kernel would read
    with pytest.raises(ValueError):
        hip.mlp_forward(net.packed_weights(0), 0, True, 4, hip.IN_RAYS, rays, torch.zeros(8, 3, device=dev), 8, 4, 11)
    with pytest.raises(ValueError):
        hip.mlp_forward(net.packed_weights(0), 0, True, 4, hip.IN_POINTS, torch.zeros(8 * 4 - 1, 3, device=dev),
                        rays[:, 8:11].contiguous(), 8, 4)
    with pytest.raises(ValueError):
        hip.mlp_forward(net.packed_weights(0), 0, True, 4, hip.IN_FEATURES, torch.zeros(8, 63, device=dev), None, 8, 1)
    with pytest.raises(ValueError):
        hip.raw2outputs(raw, z, rays[:, 3:6], 11, noise=torch.zeros(8, 3, device=dev))
    with pytest.raises(ValueError):
        hip.make_rays(2, 4, np.eye(3), rays_o=rays[:, :3], rays_d=rays[:, 3:6], near=torch.zeros(5, device=dev))
    with pytest.raises(ValueError):
        hip.make_rays(2, 4, np.eye(3), rays_o=rays[:4, :3], rays_d=rays[:, 3:6])
    with pytest.raises(ValueError):
        hip.z_vals(rays, 8, t_vals=torch.zeros(4, device=dev))
    with pytest.raises(ValueError):
        hip.sample_pdf(torch.zeros(4, 9, device=dev), torch.zeros(4, 7, device=dev), hip.linspace01(4, dev), bins_are_z=True,
                       z_vals=torch.zeros(3, 9, device=dev), want_sorted=True)


# ------------------------------------------------------------------ several networks in one launch
@pytest.mark.parametrize("precision", ["fp32", "bf16", "bf16x3"])
def test_object_batch_equals_per_object_renders(dev, precision):
    """Three objects, each with its own network, rendered as one batch (nerf_mlp_forward_objects: one network launch)
    against three separate renders: bit for bit, in both precisions, also for shares that do not fill the kernel's
    tiles (320 samples in bf16 mode, 128 in float32 mode: a network's last tile may be partial in both)."""
    import types
    from nerf_loc_amd import nerf_utils
    args = types.SimpleNamespace(nerf_render_rgb=True, nerf_render_depth=True)
    kws = []
    for o in range(3):
        _, kw, _ = nerf_utils.create_nerf_model(args, ckpt_path=None, device=dev, near=0.1, far=9.0)
        kw["network_fn"].load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(70 + o, "B").items()})
        kw["network_fn"].requires_grad_(False)
        kw["network_fn"].precision = precision
        kws.append(kw)
    for H, W, P in ((16, 12, 2), (10, 9, 2), (5, 3, 1)):      # samples per object: 24 576, 11 520, 960 (7.5 x 128, 3 x 320)
        K = np.array([[14.0, 0, 0.5 * W], [0, 14.0, 0.5 * H], [0, 0, 1]])
        poses = [[T(synth.pose_spherical(40.0 * o + 25.0 * i, -25.0, 1.6)[:3, :4].astype(F32), dev) for i in range(P)]
                 for o in range(3)]
        from nerf_loc_amd import hip
        hip.MLP_EVENTS = []
        try:
            rgbs, disps, raws, depths = nerf_utils.render_path_device_objects(poses, [H, W, 14.0], K, kws)
            assert len(hip.MLP_EVENTS) == 1                        # one network launch for the three objects
        finally:
            hip.MLP_EVENTS = None
        assert tuple(raws.shape) == (3, P, H, W, 64, 4) and tuple(rgbs.shape) == (3, P, H, W, 3)
        for o in range(3):
            r1, d1, w1, z1 = nerf_utils.render_path_device(poses[o], [H, W, 14.0], K, kws[o])
            assert torch.equal(raws[o], w1) and torch.equal(rgbs[o], r1) and torch.equal(depths[o], z1)
            assert torch.equal(torch.nan_to_num(disps[o]), torch.nan_to_num(d1))
        assert float((raws[0] - raws[1]).abs().max()) > 1e-3       # different networks did render different things
    # the detector's input assembly for the batch of scans
    H, W, focal = 16, 12, 14.0
    pb = np.zeros((3, 2, 3, 5), F32)
    for o in range(3):
        for i in range(2):
            pb[o, i, :, :4] = synth.pose_spherical(40.0 * o + 25.0 * i, -25.0, 1.6)[:3, :4]
            pb[o, i, :, 4] = (H, W, focal)
    got = nerf_utils.detector_inputs_batch(T(pb, dev), kws, input_type="nerf_rendered_rgb_rendered_depth", focal_length=1.2)
    for o in range(3):
        one = nerf_utils.detector_inputs(T(pb[o], dev), kws[o], input_type="nerf_rendered_rgb_rendered_depth", focal_length=1.2)
        for k in ("raws", "coarse_raws", "rgbs", "depths", "coarse_depths"):
            assert torch.equal(got[k][o], one[k]), (o, k)


def test_object_batch_with_fine_networks(dev):
    """Lists of coarse AND fine networks: object o's share of the batch goes through its own pair in both passes,
    bit-identical to rendering each object by itself."""
    from nerf_loc_amd import render
    rays = T(synth.random_rays(192, seed=33), dev)
    cs = [load_net(synth.make_state_dict(120 + o, "B"), dev) for o in range(3)]
    fs = [load_net(synth.make_state_dict(130 + o, "B"), dev) for o in range(3)]
    q = _query(dev)
    both = render.render_rays(rays, cs, q, 64, retraw=True, N_importance=128, network_fine=fs, white_bkgd=True)
    for o in range(3):
        one = render.render_rays(rays[64 * o:64 * (o + 1)], cs[o], q, 64, retraw=True, N_importance=128, network_fine=fs[o],
                                 white_bkgd=True)
        for k in ("rgb_map", "acc_map", "rgb0", "raw", "z_std"):
            assert torch.equal(both[k][64 * o:64 * (o + 1)], one[k]), (o, k)
    assert float((both["rgb_map"][:64] - both["rgb_map"][64:128]).abs().max()) > 1e-3


def test_nonfinite_census(dev):
    from nerf_loc_amd import hip, render
    a = torch.zeros(100003, device=dev)
    b = torch.ones((7, 5), device=dev)
    c = torch.zeros(0, device=dev)
    a[5], a[100002], a[77] = float("nan"), float("inf"), -float("inf")
    assert hip.count_nonfinite([a, b, c]).tolist() == [3, 0, 0]
    many = [torch.full((3,), float("nan") if i % 3 == 0 else 1.0, device=dev) for i in range(20)]
    assert hip.count_nonfinite(many).tolist() == [3 if i % 3 == 0 else 0 for i in range(20)]
    import io, contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        render.warn_nonfinite({"rgb_map": a, "disp_map": a, "acc_map": b, "none": None}, skip=("disp_map",))
    assert buf.getvalue().count("Numerical Error") == 1 and "rgb_map" in buf.getvalue()
    # deferred mode: nothing is printed by the call itself, everything by the flush
    render.NONFINITE_REPORT = 'deferred'
    try:
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            render.warn_nonfinite({"rgb_map": a, "acc_map": b}, skip=())
            first = buf.getvalue()
            render.flush_nonfinite_reports()
        assert first == "" and buf.getvalue().count("Numerical Error") == 1 and not render._PENDING_REPORTS
    finally:
        render.NONFINITE_REPORT = 'now'


# ------------------------------------------------------------------ full size, size-independent properties
def test_full_size_view_properties(dev):
    """The 800 x 800, 64 + 128 workload of the benchmark (640 000 rays) in float32 mode: what must hold at any size
    (sortedness and permutation property of the merged depths, weights summing to acc, value ranges, independence
    of how the rays are split), plus a random sample of rays against the oracle."""
    from nerf_loc_amd import hip, render
    H, W, K, c2w, near, far = synth.lego_camera(800, theta=30.0)
    sc, sf = synth.make_state_dict(0, "B"), synth.make_state_dict(1, "B")
    nc, nf = load_net(sc, dev), load_net(sf, dev)
    rays, _ = render._build_rays(H, W, K, None, T(c2w.astype(F32), dev), False, near, far, True, None)
    assert rays.shape == (640000, 11)
    kw = dict(network_fn=nc, network_query_fn=_query(dev), N_samples=64, N_importance=128, network_fine=nf,
              white_bkgd=True)
    whole = render.batchify_rays(rays, 1 << 20, **kw)
    # (1) any split of the ray batch gives bit-identical maps
    lo = render.batchify_rays(rays[:333333], 1 << 20, **kw)
    hi = render.batchify_rays(rays[333333:], 100000, **kw)
    for k in ("rgb_map", "acc_map", "rgb0", "acc0", "z_std"):
        assert torch.equal(whole[k], torch.cat([lo[k], hi[k]], 0)), k
    # (2) ranges
    assert float(whole["acc_map"].min()) >= 0 and float(whole["acc_map"].max()) <= 1 + 1e-5
    assert float(whole["rgb_map"].min()) >= -1e-5 and float(whole["rgb_map"].max()) <= 1 + 2e-5
    assert torch.isfinite(whole["rgb_map"]).all() and torch.isfinite(whole["z_std"]).all()
    # (3) the stages on the full batch: weights sum to acc; merged depths are sorted and a permutation of the inputs
    z = hip.z_vals(rays, 64)
    raw = hip.mlp_forward(nc.packed_weights(0), 0, True, 4, hip.IN_RAYS, rays, z, 640000, 64, 11).reshape(640000, 64, 4)
    rgb0, _, acc0, w0, depth0 = hip.raw2outputs(raw, z, rays[:, 3:6], 11, None, True)
    assert float((w0.sum(-1) - acc0).abs().max()) < 2e-6 and float(w0.min()) >= 0
    assert torch.equal(rgb0, whole["rgb0"])
    zs, inds, zsorted, std = hip.sample_pdf(z, w0[:, 1:-1], hip.linspace01(128, dev), bins_are_z=True, z_vals=z,
                                            want_inds=True, want_sorted=True, want_std=True)
    assert bool((zsorted[:, 1:] >= zsorted[:, :-1]).all())
    assert torch.equal(torch.sort(torch.cat([z, zs], -1), -1)[0], zsorted)
    assert int(inds.min()) >= 1 and int(inds.max()) <= 63
    assert bool((zs >= z[:, :1]).all()) and bool((zs <= z[:, -1:]).all())
    # (4) a random sample of rays against the oracle (coarse pass exactly comparable)
    rs = np.random.RandomState(0)
    pick = rs.choice(640000, 96, replace=False)
    o = orc.render_rays(N(rays[pick]), sc, 64, 0, white_bkgd=True)
    assert maxabs(N(whole["rgb0"][pick]), o["rgb_map"]) < 1e-4 and maxabs(N(whole["acc0"][pick]), o["acc_map"]) < 1e-4


@pytest.mark.parametrize("input_type", ["nerf", "rendered_rgb", "rendered_depth", "nerf_rendered_rgb", "nerf_rendered_depth",
                                        "nerf_rendered_rgb_rendered_depth", "rendered_rgb_rendered_depth"])
def test_detector_inputs_assembly(dev, input_type):
    """nerf_utils.detector_inputs (device tensors, both fields of view) against the oracle's restatement of the
    reference's numpy assembly (model_3detr.py:619-712) applied to render_path's outputs, for all seven input types."""
    import types
    from nerf_loc_amd import nerf_utils
    args = types.SimpleNamespace(nerf_render_rgb=True, nerf_render_depth=True)
    _, kw, _ = nerf_utils.create_nerf_model(args, ckpt_path=None, device=dev, near=0.1, far=9.0)
    kw["network_fn"].load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(33, "B").items()})
    kw["network_fn"].requires_grad_(False)
    H, W, focal = 12, 9, 11.0
    poses = np.zeros((2, 3, 5), F32)
    for i in range(2):
        poses[i, :, :4] = synth.pose_spherical(50.0 * i, -20.0, 1.5)[:3, :4]
        poses[i, :, 4] = (H, W, focal)
    got = nerf_utils.detector_inputs(T(poses, dev), kw, input_type=input_type, focal_length=1.2)

    def render_side(f):
        K = np.array([[f, 0, 0.5 * W], [0, f, 0.5 * H], [0, 0, 1]])
        rgbs, _, raws, depths = nerf_utils.render_path(T(poses[:, :, :4], dev), [H, W, focal], K, 1024 * 32, kw)
        return rgbs, raws, depths

    rgbs, raws, depths = render_side(focal)
    c_rgbs, c_raws, c_depths = render_side(focal / 1.2)
    want = orc.detector_assembly(input_type, rgbs, raws, depths)
    c_want = orc.detector_assembly(input_type, c_rgbs, c_raws, c_depths, fine_depths=depths)
    assert got["raws"].is_cuda and np.array_equal(N(got["raws"]), want, equal_nan=True)
    assert np.array_equal(N(got["coarse_raws"]), c_want, equal_nan=True)
    assert not np.isnan(N(got["raws"])).any() or input_type in ("nerf", "rendered_rgb", "nerf_rendered_rgb")


@pytest.mark.parametrize("precision", ["fp32", "bf16", "bf16x3"])
def test_packed_weight_cache_follows_the_parameters(dev, precision):
    """The kernel reads a re-laid-out copy of the weights, cached on the network: every way the reference's code
    changes parameters (in-place updates of an optimizer, load_state_dict when a checkpoint is reloaded,
    run_nerf.py:221-239, 787) must be seen by the next forward."""
    sd = synth.make_state_dict(3, "A")
    net = load_net(sd, dev, precision=precision)
    x = T(np.random.RandomState(5).uniform(-1, 1, (512, 90)).astype(F32), dev)
    tol = 1e-5 if precision == "fp32" else 2e-2
    y0 = net(x)
    assert net.packed_weights() is net.packed_weights()              # unchanged parameters: no re-pack
    with torch.no_grad():
        net.rgb_linear.bias.add_(0.5)                                # what an optimizer does
    y1 = net(x)
    assert float((y1[:, :3] - y0[:, :3] - 0.5).abs().max()) < tol and torch.equal(y1[:, 3], y0[:, 3])
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    assert torch.equal(net(x), y0)
    net.alpha_linear.bias.data.add_(0.25)                            # a write behind the version counter's back
    net.weights_changed()
    y3 = net(x)
    assert float((y3[:, 3] - y0[:, 3] - 0.25).abs().max()) < tol and torch.equal(y3[:, :3], y0[:, :3])


# ------------------------------------------------------------------ shape sweep against the reference (g14)
PDF_SHAPES = ((2, 1), (3, 2), (5, 7), (9, 16), (17, 33), (33, 64), (64, 100), (100, 128), (129, 200), (200, 300))
COMPOSITE_S = (1, 2, 3, 4, 7, 16, 31, 33, 63, 65, 100, 127, 129, 191, 193, 255, 256, 257, 300, 513)


@pytest.mark.parametrize("nb,ns", PDF_SHAPES)
def test_sample_pdf_shape_sweep(dev, golden, nb, ns):
    """Bin and sample counts from 2 / 1 to 200 / 300: indices and samples bit-identical to the reference's."""
    from nerf_loc_amd import hip
    g = golden("g14_shape_sweep")
    tag = "p%d_%d" % (nb, ns)
    bins, w = g[tag + "_bins"], g[tag + "_weights"]
    np.random.seed(0)
    u_rnd = np.random.rand(bins.shape[0], ns).astype(F32)
    for mode, u in (("det", orc.linspace01(ns)), ("rnd", u_rnd)):
        zs, inds, _, _ = hip.sample_pdf(T(bins, dev), T(w, dev), T(u, dev), want_inds=True)
        assert np.array_equal(N(inds).astype(np.int64), g["%s_%s_inds" % (tag, mode)].astype(np.int64)), mode
        assert np.array_equal(N(zs).view(np.uint32), g["%s_%s_samples" % (tag, mode)].view(np.uint32)), mode


@pytest.mark.parametrize("S", COMPOSITE_S)
def test_raw2outputs_shape_sweep(dev, golden, S):
    """1 to 513 samples per ray against the reference.  One sample per ray is the reference's degenerate case: the
    interval list it builds is empty (run_nerf.py:283-284 appends the 1e10 tail with the shape of an empty slice), so
    nothing is composited: weights [R, 0], acc = depth = 0, disp NaN, rgb the background."""
    from nerf_loc_amd import render
    g = golden("g14_shape_sweep")
    out = render.raw2outputs(T(g["c%d_raw" % S], dev), T(g["c%d_z" % S], dev), T(g["c%d_d" % S], dev),
                             white_bkgd=(S % 2 == 1))
    for name, v in zip(("rgb", "disp", "acc", "weights", "depth"), out):
        ref, v = g["c%d_%s" % (S, name)], N(v)
        assert v.shape == ref.shape, (name, v.shape, ref.shape)
        if name == "disp":
            assert np.array_equal(np.isnan(v), np.isnan(ref))
            fin = np.isfinite(ref) & (np.abs(ref) < 1e6)
            assert maxabs(v[fin] / ref[fin], np.ones(fin.sum())) < 2e-5
        elif ref.size:
            assert maxabs(v, ref) < 1e-5 * max(1.0, float(np.abs(ref).max())), name


# ------------------------------------------------------------------ depths and encodings across shapes (g15, reference)
@pytest.mark.parametrize("S", (1, 2, 3, 5, 63, 64, 65, 192, 300))
def test_depths_shape_sweep(dev, golden, S):
    """The depths render_rays hands to raw2outputs in the reference, 1 to 300 samples, linear and in disparity:
    bit-equal without jitter, 1e-6 with the reference's seeded jitter."""
    from nerf_loc_amd import hip
    g = golden("g15_depths_embed")
    rays = T(g["rays"], dev)
    for lindisp in (False, True):
        tag = "z%d_%s_" % (S, "disp" if lindisp else "lin")
        assert np.array_equal(N(hip.z_vals(rays, S, lindisp)), g[tag + "det"]), tag
        z = N(hip.z_vals(rays, S, lindisp, T(np_rand((g["rays"].shape[0], S)), dev)))
        assert maxabs(z, g[tag + "jit"]) < 1e-6, tag


def test_embed_other_frequency_counts(dev, golden):
    from nerf_loc_amd import helpers, hip
    g = golden("g15_depths_embed")
    x = T(g["x"], dev)
    for L in (0, 1, 2, 6, 15):
        fn, ch = helpers.get_embedder(L, 0)
        assert ch == 3 + 6 * L
        e = N(fn(x))
        assert e.shape == g["embed%d" % L].shape
        assert maxabs(e, g["embed%d" % L]) < 5e-7, L
        assert np.array_equal(N(hip.embed(x, L)), e)
    fn, ch = helpers.get_embedder(10, -1)
    assert ch == 3 and np.array_equal(N(fn(x)), g["embed_identity"])


# ------------------------------------------------------------------ render() branches against the reference (g16)
@pytest.mark.parametrize("tag", ["lindisp", "ndc", "ndc_static", "novd", "static"])
def test_render_branches_vs_reference(dev, golden, tag):
    """render() with the NDC warp (forward-facing scenes: near 0, far 1), depths linear in disparity, a static camera
    for the geometry (alone and under the NDC warp) and a network without view directions, each against the
    reference's own render (run_nerf.py:95-123).  Coarse maps: 1e-4 on every pixel.  Fine maps, free-running: 1e-4
    on at least 99 % of the well-conditioned pixels and 2e-4 on all of them: importance samples in nearly empty bins
    move with the last bit of the coarse weights (the reference's own float32 and float64 renders of these views
    differ by up to 4e-4); on the reference's own samples the oracle agrees with it to 1e-6."""
    from test_oracle_golden import render_branch_inputs
    from nerf_loc_amd import render
    g = golden("g16_render_branches")
    H, W, K, c2w, vd, sds, static, kw = render_branch_inputs(g, tag)
    nets = [load_net(sd, dev, use_viewdirs=vd, output_ch=4 if vd else 5) for sd in sds]
    from nerf_loc_amd import helpers
    q = _query(dev) if vd else render.NetworkQuery(helpers.get_embedder(10, 0)[0], None, 1024 * 64)
    rgb, disp, acc, extras = render.render(H, W, K, chunk=200, c2w=T(c2w, dev), use_viewdirs=vd,
                                           c2w_staticcam=None if static is None else T(static, dev),
                                           network_fn=nets[0], network_fine=nets[1], network_query_fn=q, N_samples=64,
                                           N_importance=128, perturb=0., white_bkgd=True, raw_noise_std=0., **kw)
    assert rgb.shape == (H, W, 3) and acc.shape == (H, W)
    e0 = max(maxabs(N(extras["rgb0"]), g[tag + "_rgb0"]), maxabs(N(extras["acc0"]), g[tag + "_acc0"]))
    ok = np.abs(g[tag + "_rawlast64"]) > 1e-3
    err = np.maximum(np.abs(N(rgb) - g[tag + "_rgb"]).max(-1), np.abs(N(acc) - g[tag + "_acc"]))[ok]
    own = max(maxabs(g[tag + "_rgb"][ok], g[tag + "_rgb64"][ok]), maxabs(g[tag + "_acc"][ok], g[tag + "_acc64"][ok]))
    print("render() branch %s: coarse maps max abs err %.2e; fine maps max %.2e, 99th percentile %.2e (the reference's "
          "float32 vs float64: %.2e); %d of %d pixels masked" % (tag, e0, err.max(), np.percentile(err, 99), own,
                                                                  (~ok).sum(), ok.size))
    assert e0 < 1e-4
    assert np.percentile(err, 99) < 1e-4 and err.max() < 2e-4        # measured: every branch below 7.5e-5
