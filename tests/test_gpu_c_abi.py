"""The C ABI without Python in the loop: nerf-loc_amd/lib/render_c (tools/c_abi/render_c.cpp, linked against
libnerf_hip.so and the HIP runtime only) renders one view with seven library calls on its own stream, once issued
directly and once recorded into a HIP graph and replayed.  Both must equal, bit for bit, what render() of the Python
host mirror returns for the same pose and networks (run_nerf.py:69-134, 314-424): the mirror adds nothing to the
arithmetic, and no entry point on the path allocates or synchronises (otherwise the capture would fail)."""
import os
import subprocess

import numpy as np
import pytest
import torch

from nerf_loc_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BINARY = os.path.join(ROOT, "nerf-loc_amd", "lib", "render_c")

ORDER = (["pts_linears.%d.%s" % (i, p) for i in range(8) for p in ("weight", "bias")] +
         ["views_linears.0.weight", "views_linears.0.bias", "feature_linear.weight", "feature_linear.bias",
          "alpha_linear.weight", "alpha_linear.bias", "rgb_linear.weight", "rgb_linear.bias"])


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from nerf_loc_amd import hip
    hip.lib()
    if not os.path.exists(BINARY):
        raise RuntimeError("%s is missing: build it with `make -C nerf-loc_amd/csrc`" % BINARY)
    return torch.device("cuda:0")


def _write_input(path, H, W, K, c2w, near, far, precision, sds):
    from nerf_loc_amd import hip
    t_vals = hip.linspace01(64, "cpu").numpy()
    u = hip.linspace01(128, "cpu").numpy()
    with open(path, "wb") as f:
        np.array([0x4652454e, H, W, 64, 128, precision, 1], np.int32).tofile(f)
        np.array([K[0][0], K[1][1], K[0][2], K[1][2], near, far], np.float32).tofile(f)
        np.asarray(c2w, np.float32)[:3, :4].tofile(f)
        t_vals.astype(np.float32).tofile(f)
        u.astype(np.float32).tofile(f)
        for sd in sds:
            for k in ORDER:
                np.ascontiguousarray(sd[k], np.float32).tofile(f)


def _read_output(path, n):
    a = np.fromfile(path, np.float32)
    assert a.size == n * 11
    out, o = {}, 0
    for k, w in (("rgb_map", 3), ("disp_map", 1), ("acc_map", 1), ("rgb0", 3), ("disp0", 1), ("acc0", 1), ("z_std", 1)):
        out[k] = a[o:o + n * w].reshape((n, w) if w > 1 else (n,))
        o += n * w
    return out


def _python_render(dev, H, W, K, c2w, near, far, precision, sds):
    from nerf_loc_amd import helpers, render
    nets = []
    for sd in sds:
        n = helpers.NeRF(D=8, W=256, input_ch=63, output_ch=4, skips=[4], input_ch_views=27, use_viewdirs=True)
        n.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        n = n.to(dev).requires_grad_(False)
        n.precision = ("fp32", "bf16", "bf16x3")[precision]
        nets.append(n)
    q = render.NetworkQuery(helpers.get_embedder(10, 0)[0], helpers.get_embedder(4, 0)[0], 1024 * 64)
    rgb, disp, acc, extras = render.render(H, W, K, chunk=H * W, c2w=torch.from_numpy(c2w).to(dev), near=near, far=far,
                                           use_viewdirs=True, network_fn=nets[0], network_fine=nets[1],
                                           network_query_fn=q, N_samples=64, N_importance=128, perturb=0.,
                                           white_bkgd=True, raw_noise_std=0.)
    ret = dict(rgb_map=rgb, disp_map=disp, acc_map=acc, **extras)
    return {k: v.reshape(H * W, -1).squeeze(-1).cpu().numpy() for k, v in ret.items()}


def _run(args):
    p = subprocess.run([BINARY] + args, capture_output=True, text=True, timeout=100)
    assert p.returncode == 0, "render_c failed (%d): %s" % (p.returncode, p.stderr[-2000:])
    return p.stdout


@pytest.mark.parametrize("precision", [0, 1, 2], ids=["fp32", "bf16", "bf16x3"])
def test_c_program_direct_and_graph_equal_the_python_mirror(dev, tmp_path, precision):
    H, W, K, c2w, near, far = synth.lego_camera(48, theta=75.0)
    sds = [synth.make_state_dict(11, "B"), synth.make_state_dict(12, "B")]
    inp = str(tmp_path / "in.bin")
    _write_input(inp, H, W, K, c2w, near, far, precision, sds)
    want = _python_render(dev, H, W, K, c2w.astype(np.float32), near, far, precision, sds)
    torch.cuda.synchronize()
    got = {}
    for mode in ("direct", "graph", "onecall"):
        out = str(tmp_path / (mode + ".bin"))
        log = _run([inp, out, mode] + (["20"] if mode == "graph" else []))
        if log.strip():
            print("render_c %s, %s: %s" % (mode, ("fp32", "bf16", "bf16x3")[precision], log.strip()))
        got[mode] = _read_output(out, H * W)
    assert 0.05 < float(want["acc_map"].mean()) and np.isfinite(want["rgb_map"]).all()
    for k in ("rgb_map", "disp_map", "acc_map", "rgb0", "disp0", "acc0", "z_std"):
        for mode in ("direct", "graph", "onecall"):
            assert np.array_equal(got[mode][k], want[k], equal_nan=True), (k, mode)


@pytest.mark.parametrize("precision", ["fp32", "bf16", "bf16x3"])
def test_one_call_render_rays_equals_the_calls_one_by_one(dev, precision):
    """nerf_render_rays (the per-ray pipeline of run_nerf.py:314-424 as ONE C call) against render_rays of the Python
    mirror, which issues the entry points one by one: bit-identical maps, raw and depths, for the coarse-only flavour
    (NeRF-Loc), coarse + fine with a separate fine network, and with jitter and density noise handed in."""
    from nerf_loc_amd import helpers, hip, render
    rays = torch.from_numpy(synth.random_rays(777, seed=12)).to(dev)
    sds = [synth.make_state_dict(5, "B"), synth.make_state_dict(6, "B")]
    nets = []
    for sd in sds:
        n = helpers.NeRF(D=8, W=256, input_ch=63, output_ch=4, skips=[4], input_ch_views=27, use_viewdirs=True)
        n.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        n = n.to(dev).requires_grad_(False)
        n.precision = precision
        nets.append(n)
    q = render.NetworkQuery(helpers.get_embedder(10, 0)[0], helpers.get_embedder(4, 0)[0], 1 << 16)
    prec = nets[0].precision_code()
    pc, pf = nets[0].packed_weights(), nets[1].packed_weights()
    tv, u = hip.linspace01(64, dev), hip.linspace01(128, dev)
    # coarse only
    want = render.render_rays(rays, nets[0], q, 64, retraw=True, white_bkgd=True, _with_depth=True)
    got = hip.render_rays_fused(pc, None, rays, 64, 0, prec, True, 4, tv, white_bkgd=True)
    for k in ("rgb_map", "acc_map", "depth_map", "raw"):
        assert torch.equal(got[k], want[k]), (precision, k)
    assert torch.equal(torch.nan_to_num(got["disp_map"]), torch.nan_to_num(want["disp_map"]))
    # coarse + fine
    want = render.render_rays(rays, nets[0], q, 64, retraw=True, N_importance=128, network_fine=nets[1], white_bkgd=True,
                              _with_depth=True)
    got = hip.render_rays_fused(pc, pf, rays, 64, 128, prec, True, 4, tv, u=u, white_bkgd=True)
    for k in ("rgb_map", "acc_map", "depth_map", "raw", "rgb0", "acc0", "z_std"):
        assert torch.equal(got[k], want[k]), (precision, k)
    # jitter and noise handed in: against the entry points composed by hand
    g = torch.Generator(device=dev).manual_seed(3)
    t_rand = torch.rand(777, 64, device=dev, generator=g)
    n0, n1 = torch.randn(777, 64, device=dev, generator=g), torch.randn(777, 192, device=dev, generator=g)
    ur = torch.rand(777, 128, device=dev, generator=g)
    got = hip.render_rays_fused(pc, pf, rays, 64, 128, prec, True, 4, tv, u=ur, t_rand=t_rand, noise0=n0, noise1=n1,
                                lindisp=True, want_weights=True)
    z0 = hip.z_vals(rays, 64, True, t_rand)
    raw0 = hip.mlp_forward(pc, prec, True, 4, hip.IN_RAYS, rays, z0, 777, 64, 11).reshape(777, 64, 4)
    rgb0, _, acc0, w0, _ = hip.raw2outputs(raw0, z0, rays[:, 3:6], 11, n0, False)
    _, _, zs, std = hip.sample_pdf(z0, w0[:, 1:-1], ur, bins_are_z=True, z_vals=z0, want_samples=False, want_sorted=True,
                                   want_std=True)
    raw1 = hip.mlp_forward(pf, prec, True, 4, hip.IN_RAYS, rays, zs, 777, 192, 11).reshape(777, 192, 4)
    rgb1, _, acc1, _, depth1 = hip.raw2outputs(raw1, zs, rays[:, 3:6], 11, n1, False, want_weights=False)
    for a, b, k in ((got["rgb0"], rgb0, "rgb0"), (got["weights0"], w0, "weights0"), (got["z_vals"], zs, "z_vals"),
                    (got["z_std"], std, "z_std"), (got["raw"], raw1, "raw"), (got["rgb_map"], rgb1, "rgb_map"),
                    (got["depth_map"], depth1, "depth_map")):
        assert torch.equal(a, b), (precision, k)
    # refusals: sizes are checked before anything is launched
    with pytest.raises(ValueError, match="t_vals must have"):
        hip.render_rays_fused(pc, None, rays, 64, 0, prec, True, 4, hip.linspace01(32, dev))
    with pytest.raises(ValueError, match="needs u"):
        hip.render_rays_fused(pc, pf, rays, 64, 128, prec, True, 4, tv)
    with pytest.raises(ValueError, match="not a packed network"):
        hip.render_rays_fused(pc[:1000], None, rays, 64, 0, prec, True, 4, tv)
    assert hip.render_rays_fused(pc, None, rays[:0], 64, 0, prec, True, 4, tv)["rgb_map"].shape == (0, 3)
