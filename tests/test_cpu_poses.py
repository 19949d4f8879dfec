"""Pose utilities and the PNG writer against vectors produced by the reference's own load_llff.py / load_blender.py
(tests/golden/make_golden.py g11).  Host-only numpy work: runs without a GPU."""
import os

import numpy as np
import pytest

from nerf_loc_amd import poses as P

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "g11_poses.npz"))


def _parsed():
    h, w, pf = (int(v) for v in G["hw_pf"])
    return P.parse_poses_bounds(G["poses_bounds"], (h, w), pose_factor=pf)


def test_parse_poses_bounds_matches_reference_loader():
    poses, bds = _parsed()
    np.testing.assert_array_equal(poses, G["parsed_poses"])
    np.testing.assert_array_equal(bds, G["parsed_bds"])
    assert poses.shape == (3, 5, 14) and bds.shape == (2, 14)


@pytest.mark.parametrize("tag,kw", [("spiral", {}), ("sphere", dict(spherify=True)),
                                    ("norecenter", dict(recenter=False, bd_factor=None))])
def test_llff_pose_pipeline(tag, kw):
    poses, bds = _parsed()
    got_p, got_b, got_r, i_test = P.llff_poses(poses, bds, **kw)
    assert got_p.dtype == np.float32 and got_r.dtype == np.float32
    assert got_p.shape == G[tag + "_poses"].shape and got_r.shape == G[tag + "_render_poses"].shape
    # same float32/float64 mix as the reference: identical arithmetic, so only BLAS ordering can differ
    np.testing.assert_allclose(got_p, G[tag + "_poses"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(got_b, G[tag + "_bds"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(got_r, G[tag + "_render_poses"], rtol=0, atol=2e-6)
    assert i_test == int(G[tag + "_i_test"])


def test_path_zflat_gives_half_a_flat_turn():
    # the reference's own path_zflat branch cannot run with numpy >= 1.18 (a float count reaches linspace), so this
    # case has no golden: check its defining properties instead
    poses, bds = _parsed()
    _, _, flat, _ = P.llff_poses(poses, bds, path_zflat=True)
    _, _, spiral, _ = P.llff_poses(poses, bds)
    assert flat.shape == (60, 3, 5) and spiral.shape == (120, 3, 5)
    c2w = P.poses_avg(P.llff_poses(poses, bds)[0])
    # no excursion along the average camera's z axis
    z = (flat[:, :3, 3] - flat[:, :3, 3].mean(0)) @ c2w[:3, 2]
    assert np.abs(z).max() < 1e-5


def test_recenter_makes_average_pose_identity():
    poses, bds = _parsed()
    p, *_ = P.llff_poses(poses, bds)
    avg = P.poses_avg(p)
    np.testing.assert_allclose(avg[:3, :4], np.eye(4)[:3], atol=1e-5)


def test_viewmatrix_is_orthonormal_and_ptstocam_inverts():
    rs = np.random.RandomState(0)
    m = P.viewmatrix(rs.normal(size=3), rs.normal(size=3), rs.normal(size=3))
    np.testing.assert_allclose(m[:, :3].T @ m[:, :3], np.eye(3), atol=1e-12)
    pts = rs.normal(size=(5, 3))
    cam = P.ptstocam(pts, m)
    np.testing.assert_allclose(cam @ m[:, :3].T + m[:, 3], pts, atol=1e-12)


def test_pose_spherical_matches_load_blender():
    got = np.stack([P.pose_spherical(a, -30.0, 4.0).numpy() for a in np.linspace(-180, 180, 9)[:-1]])
    np.testing.assert_allclose(got, G["pose_spherical"], atol=1e-6)


@pytest.mark.parametrize("shape", [(7, 5, 3), (4, 9), (3, 2, 4), (1, 1, 3)])
def test_png_round_trip(tmp_path, shape):
    img = np.random.RandomState(1).randint(0, 256, size=shape).astype(np.uint8)
    path = str(tmp_path / "a.png")
    P.write_png(path, img)
    with open(path, "rb") as f:
        assert f.read(8) == b"\x89PNG\r\n\x1a\n"
    np.testing.assert_array_equal(P.read_png(path), img)


def test_png_rejects_other_types(tmp_path):
    with pytest.raises(ValueError):
        P.write_png(str(tmp_path / "a.png"), np.zeros((2, 2, 3), np.float32))
    with pytest.raises(ValueError):
        P.write_png(str(tmp_path / "a.png"), np.zeros((2, 2, 2), np.uint8))


def test_objectbank_reader_orders_and_checks_tensors(tmp_path):
    """Host half of the object bank: a checkpoint becomes one flat buffer in state_dict order; foreign layouts raise."""
    import torch
    from nerf_loc_amd import synth
    from nerf_loc_amd.helpers import NeRF
    from nerf_loc_amd.objectbank import read_flat, state_dict_layout
    net = NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=4, skips=[4], use_viewdirs=True)
    layout, total = state_dict_layout(net)
    assert total == 595844 and [n for n, _, _, _ in layout] == list(net.state_dict().keys())
    sd = synth.make_state_dict(5, "A")
    path = str(tmp_path / "a.tar")
    torch.save({'global_step': 77, 'network_fn_state_dict': {k: torch.from_numpy(v) for k, v in sd.items()}}, path)
    flat, step = read_flat(path, layout, total)
    assert step == 77 and flat.shape == (total,)
    for name, shape, off, n in layout:
        assert np.array_equal(flat[off:off + n].numpy().reshape(shape), sd[name]), name
    sd2 = dict(sd)
    sd2["alpha_linear.weight"] = np.zeros((2, 256), np.float32)
    torch.save({'global_step': 0, 'network_fn_state_dict': {k: torch.from_numpy(v) for k, v in sd2.items()}}, path)
    with pytest.raises(ValueError):
        read_flat(path, layout, total)
    del sd2["alpha_linear.weight"]
    torch.save({'global_step': 0, 'network_fn_state_dict': {k: torch.from_numpy(v) for k, v in sd2.items()}}, path)
    with pytest.raises(ValueError):
        read_flat(path, layout, total)


def test_avi_writer_round_trips(tmp_path):
    """render_path's videos without imageio (run_nerf.py:681, 819-820 write MP4 through imageio + ffmpeg): the AVI
    stand-in, uncompressed (exact) and Motion-JPEG (close), and write_video's choice of file."""
    rs = np.random.RandomState(0)
    frames = rs.randint(0, 256, size=(5, 12, 17, 3)).astype(np.uint8)          # odd width: rows are padded to 4 bytes
    path = str(tmp_path / "raw.avi")
    P.write_avi(path, frames, fps=30)
    got, fps = P.read_avi(path)
    assert fps == 30 and np.array_equal(got, frames)
    data = open(path, "rb").read()
    assert data[:4] == b"RIFF" and int.from_bytes(data[4:8], "little") == len(data) - 8 and b"idx1" in data
    smooth = (np.linspace(0, 255, 32 * 48).reshape(1, 32, 48, 1) * np.ones((4, 1, 1, 3))).astype(np.uint8)
    try:
        import PIL  # noqa: F401
        P.write_avi(str(tmp_path / "mj.avi"), smooth, fps=24, jpeg_quality=90)
        got, fps = P.read_avi(str(tmp_path / "mj.avi"))
        assert fps == 24 and got.shape == smooth.shape and np.abs(got.astype(int) - smooth.astype(int)).max() <= 6
    except ImportError:
        pass
    with pytest.raises(ValueError):
        P.write_avi(path, frames.astype(np.float32))
    with pytest.raises(ValueError):
        P.write_avi(path, frames[:0])
    from nerf_loc_amd import render
    out = render.write_video(str(tmp_path / "spiral_rgb.mp4"), smooth.astype(np.float32) / 255.0, fps=30, quality=8)
    assert os.path.exists(out) and os.path.getsize(out) > 0
    try:
        import imageio  # noqa: F401
        assert out.endswith(".mp4")
    except ImportError:
        assert out.endswith(".avi")
        got, fps = P.read_avi(out)
        assert fps == 30 and got.shape == smooth.shape
