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
        n.precision = "bf16" if precision else "fp32"
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


@pytest.mark.parametrize("precision", [0, 1], ids=["fp32", "bf16"])
def test_c_program_direct_and_graph_equal_the_python_mirror(dev, tmp_path, precision):
    H, W, K, c2w, near, far = synth.lego_camera(48, theta=75.0)
    sds = [synth.make_state_dict(11, "B"), synth.make_state_dict(12, "B")]
    inp = str(tmp_path / "in.bin")
    _write_input(inp, H, W, K, c2w, near, far, precision, sds)
    want = _python_render(dev, H, W, K, c2w.astype(np.float32), near, far, precision, sds)
    torch.cuda.synchronize()
    got = {}
    for mode in ("direct", "graph"):
        out = str(tmp_path / (mode + ".bin"))
        log = _run([inp, out, mode] + (["20"] if mode == "graph" else []))
        if log.strip():
            print("render_c %s, %s: %s" % (mode, "bf16" if precision else "fp32", log.strip()))
        got[mode] = _read_output(out, H * W)
    assert 0.05 < float(want["acc_map"].mean()) and np.isfinite(want["rgb_map"]).all()
    for k in ("rgb_map", "disp_map", "acc_map", "rgb0", "disp0", "acc0", "z_std"):
        for mode in ("direct", "graph"):
            assert np.array_equal(got[mode][k], want[k], equal_nan=True), (k, mode)
