"""Streaming 64 per-object checkpoints to the GPU: the object bank (parallel reads, one copy + one packing launch per
object, networks stay resident) against reloading the single network per object as the reference does
(model_3detr.py:619-624, utils/nerf_utils.py:88-105).  Checkpoints have nerf-pytorch's full layout (two networks and
Adam's moments: 14 MB each).  Prints one JSON line."""
import json
import os
import sys
import tempfile
import time
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from nerf_loc_amd import nerf_utils, synth
    from nerf_loc_amd.objectbank import ObjectBank
    dev = torch.device("cuda:0")
    n_obj = int(os.environ.get("N_OBJ", 64))
    args = types.SimpleNamespace(nerf_render_rgb=True, nerf_render_depth=True)
    _, kw, _ = nerf_utils.create_nerf_model(args, ckpt_path=None, device=dev, near=0.1, far=9.0)
    kw["network_fn"].requires_grad_(False)
    with tempfile.TemporaryDirectory() as d:
        paths = []
        for o in range(n_obj):
            sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(o, "B").items()}
            adam = {'state': {i: {'step': torch.tensor(1.0), 'exp_avg': torch.zeros_like(t), 'exp_avg_sq': torch.zeros_like(t)}
                              for i, t in enumerate(list(sd.values()) * 2)}, 'param_groups': [{'lr': 5e-4}]}
            p = os.path.join(d, "%03d.tar" % o)
            torch.save({'global_step': o, 'network_fn_state_dict': sd, 'network_fine_state_dict': sd,
                        'optimizer_state_dict': adam}, p)
            paths.append([p])
        size = os.path.getsize(paths[0][0])
        torch.cuda.synchronize()
        t = time.perf_counter()
        for o in range(n_obj):
            nerf_utils.load_nerf_model(kw, paths[o])
            kw["network_fn"].packed_weights()
        torch.cuda.synchronize()
        reload_s = time.perf_counter() - t
        out = {"objects": n_obj, "checkpoint_bytes": size, "reload_per_object_ms": 1e3 * reload_s / n_obj}
        # where a cold load goes: the restricted unpickler alone, on one thread and on four at once (wall time per file)
        import concurrent.futures as cf
        from nerf_loc_amd.objectbank import read_flat, state_dict_layout
        layout, total = state_dict_layout(kw["network_fn"])
        flat_paths = [p[0] for p in paths]
        t = time.perf_counter()
        for p in flat_paths:
            read_flat(p, layout, total)
        out["read_flat_1_thread_ms_per_file"] = 1e3 * (time.perf_counter() - t) / n_obj
        with cf.ThreadPoolExecutor(4) as ex:
            t = time.perf_counter()
            list(ex.map(lambda p: read_flat(p, layout, total), flat_paths))
            out["read_flat_4_threads_ms_per_file"] = 1e3 * (time.perf_counter() - t) / n_obj
        for workers in (1, 4):
            bank = ObjectBank(kw, workers=workers, allow_slow_readers=True)
            t = time.perf_counter()
            kws = bank.kwargs_list([(o, paths[o]) for o in range(n_obj)])
            torch.cuda.synchronize()
            cold = time.perf_counter() - t
            t = time.perf_counter()
            kws = bank.kwargs_list([(o, paths[o]) for o in range(n_obj)])
            torch.cuda.synchronize()
            warm = time.perf_counter() - t
            out["bank_workers_%d" % workers] = {"cold_per_object_ms": 1e3 * cold / n_obj, "warm_per_object_ms": 1e3 * warm / n_obj,
                                                "resident_MB": bank.resident_bytes() / 1e6}
            bank.close()
        # and the render the bank feeds: all objects, both fields of view, 8 poses, one launch per field of view
        H, W, f = 240, 180, 196.46
        poses = torch.from_numpy(np.stack([np.concatenate([synth.pose_spherical(45.0 * i, -30.0, 4.0)[:3, :4],
                                                           np.array([[H], [W], [f]], np.float32)], 1) for i in range(8)]))
        n_r = min(n_obj, 8)
        bank = ObjectBank(kw)
        kws = bank.kwargs_list([(o, paths[o]) for o in range(n_r)])
        pb = poses[None].repeat(n_r, 1, 1, 1).to(dev)
        from nerf_loc_amd import helpers
        for prec in ("fp32", "bf16"):
            helpers.set_default_precision(prec)
            for _ in range(2):
                torch.cuda.synchronize()
                t = time.perf_counter()
                res = nerf_utils.detector_inputs_batch(pb, kws, 'nerf', focal_length=1.2)
                torch.cuda.synchronize()
                out["detector_inputs_batch_%d_objects_%s_ms" % (n_r, prec)] = 1e3 * (time.perf_counter() - t)
        bank.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
