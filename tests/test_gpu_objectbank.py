"""Object bank: many per-object checkpoints streamed to the GPU and rendered as one batch, against the reference's way
(load_nerf_model into the single network, then render; model_3detr.py:619-650, utils/nerf_utils.py:88-105)."""
import types

import numpy as np
import pytest
import torch

from nerf_loc_amd import synth

pytestmark = pytest.mark.gpu


def _write_ckpt(path, seed, step):
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(seed, "B").items()}
    # the dict run_nerf.py:803-811 saves; only network_fn_state_dict is read on this path
    torch.save({'global_step': step, 'network_fn_state_dict': sd, 'network_fine_state_dict': sd,
                'optimizer_state_dict': {}}, path)


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def test_bank_batch_equals_reload_per_object(dev, tmp_path):
    from nerf_loc_amd import nerf_utils
    from nerf_loc_amd.objectbank import ObjectBank
    args = types.SimpleNamespace(nerf_render_rgb=True, nerf_render_depth=True)
    _, kw_test, _ = nerf_utils.create_nerf_model(args, ckpt_path=None, device=dev, near=0.1, far=9.0)
    kw_test["network_fn"].requires_grad_(False)
    n_obj, H, W = 5, 24, 18
    paths = []
    for o in range(n_obj):
        p = str(tmp_path / ("obj%d.tar" % o))
        _write_ckpt(p, 300 + o, 1000 + o)
        paths.append([str(tmp_path / "older.tar"), p])          # a list, the last entry counts
    K = np.array([[20.0, 0, 9.0], [0, 20.0, 12.0], [0, 0, 1]])
    poses = [torch.from_numpy(np.stack([synth.pose_spherical(30.0 * i + 7 * o, -30.0, 4.0)[:3, :4] for i in range(2)])).to(dev)
             for o in range(n_obj)]

    bank = ObjectBank(kw_test, workers=3)
    kws = bank.kwargs_list([("o%d" % o, paths[o]) for o in range(n_obj)])
    assert len(bank) == n_obj and bank.stats['reads'] == n_obj and bank.stats['loads'] == n_obj
    assert [bank.global_step("o%d" % o) for o in range(n_obj)] == [1000 + o for o in range(n_obj)]
    rgbs, disps, raws, depths = nerf_utils.render_path_device_objects(poses, [H, W, 20.0], K, kws)
    for o in range(n_obj):
        # the reference's way: one network, reloaded for every object
        kw_one, start = nerf_utils.load_nerf_model(kw_test, paths[o])
        assert start == 1000 + o
        r1, d1, w1, z1 = nerf_utils.render_path_device(poses[o], [H, W, 20.0], K, kw_one)
        assert torch.equal(rgbs[o], r1) and torch.equal(raws[o], w1) and torch.equal(depths[o], z1)
        assert torch.equal(torch.nan_to_num(disps[o], nan=-1.0), torch.nan_to_num(d1, nan=-1.0))
    # a second request is served from the GPU: no file is read again
    kws2 = bank.kwargs_list([("o%d" % o, paths[o]) for o in range(n_obj)])
    assert bank.stats['reads'] == n_obj and bank.stats['hits'] == n_obj
    assert all(a["network_fn"] is b["network_fn"] for a, b in zip(kws, kws2))
    assert bank.resident_bytes() >= n_obj * 595844 * 4
    # resident objects can be asked for by key alone; unknown keys cannot
    assert bank.kwargs("o3")["network_fn"] is kws[3]["network_fn"]
    with pytest.raises(KeyError):
        bank.kwargs("nobody")
    bank.close()


def test_bank_capacity_and_refusals(dev, tmp_path):
    from nerf_loc_amd import nerf_utils
    from nerf_loc_amd.objectbank import ObjectBank
    args = types.SimpleNamespace(nerf_render_rgb=True, nerf_render_depth=True)
    _, kw_test, _ = nerf_utils.create_nerf_model(args, ckpt_path=None, device=dev, near=0.1, far=9.0)
    paths = []
    for o in range(4):
        paths.append(str(tmp_path / ("c%d.tar" % o)))
        _write_ckpt(paths[-1], 400 + o, o)
    bank = ObjectBank(kw_test, capacity=2, workers=2)
    for o in range(4):
        bank.kwargs(o, paths[o])
    assert len(bank) == 2 and 3 in bank and 2 in bank and 0 not in bank and bank.stats['evictions'] == 2
    # an evicted object comes back from its file, with the same weights
    net0 = bank.network(0, paths[0])
    want = synth.make_state_dict(400, "B")
    torch.cuda.synchronize()
    for k, v in net0.state_dict().items():
        assert np.array_equal(v.cpu().numpy(), want[k]), k
    # several threads asking for the same and for different objects at once
    import threading
    bank2 = ObjectBank(kw_test, workers=2)
    got, errs = {}, []

    def ask(i):
        try:
            got[i] = bank2.network(i % 4, paths[i % 4])
        except Exception as ex:      # noqa: BLE001
            errs.append(ex)
    threads = [threading.Thread(target=ask, args=(i,)) for i in range(12)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errs and len(bank2) == 4 and bank2.stats['loads'] == 4
    assert all(got[i] is got[i % 4] for i in range(12))
    bank2.close()
    # a checkpoint of another architecture is refused by name, not loaded
    bad = str(tmp_path / "bad.tar")
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(1, "A", use_viewdirs=False, output_ch=5).items()}
    torch.save({'global_step': 0, 'network_fn_state_dict': sd}, bad)
    with pytest.raises(ValueError):
        bank.kwargs("bad", bad)
    with pytest.raises(ValueError):
        ObjectBank({**kw_test, 'network_fn': torch.nn.Linear(3, 3)})
    bank.close()


def test_failed_read_leaves_the_bank_usable(dev, tmp_path):
    """A checkpoint of another architecture is refused; its pinned staging buffer goes back to the pool, nothing stays
    listed as on its way, and the same key loads fine from a good file afterwards.  Reader threads are capped at one
    (more are slower: the unpickler holds the interpreter lock) unless asked for explicitly."""
    from nerf_loc_amd import nerf_utils
    from nerf_loc_amd.objectbank import ObjectBank
    args = types.SimpleNamespace(nerf_render_rgb=True, nerf_render_depth=True)
    _, kw_test, _ = nerf_utils.create_nerf_model(args, ckpt_path=None, device=dev, near=0.1, far=9.0)
    kw_test["network_fn"].requires_grad_(False)
    assert ObjectBank(kw_test, workers=8).workers == 1
    assert ObjectBank(kw_test, workers=3, allow_slow_readers=True).workers == 3
    bank = ObjectBank(kw_test)
    good, bad = str(tmp_path / "good.tar"), str(tmp_path / "bad.tar")
    _write_ckpt(good, 410, 5)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(411, "B").items()}
    sd["alpha_linear.weight"] = torch.zeros(2, 256)
    torch.save({'global_step': 0, 'network_fn_state_dict': sd}, bad)
    with pytest.raises(ValueError, match="alpha_linear.weight"):
        bank.kwargs("k", bad)
    assert "k" not in bank._pending and "k" not in bank and len(bank._staging) == 1
    net = bank.kwargs("k", good)["network_fn"]
    torch.cuda.synchronize()
    assert bank.global_step("k") == 5 and "k" not in bank._pending
    assert torch.equal(net.alpha_linear.weight.cpu(), torch.from_numpy(synth.make_state_dict(410, "B")["alpha_linear.weight"]))
    bank.close()
