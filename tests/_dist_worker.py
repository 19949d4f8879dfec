"""Rank body of the world_size-2 gloo tests (started by test_dist_gloo.py, one process per rank).
The render function is a deterministic stand-in with the same dict-of-maps contract as
render.batchify_rays, so the sharding / record packing / all-gather logic runs without a GPU."""
import json
import sys

import torch

import nerf_loc_amd
from nerf_loc_amd import dist as ndist


def fake_render(rays):
    o, d = rays[:, 0:3], rays[:, 3:6]
    s = (o * d).sum(-1)
    return {"rgb_map": torch.stack([s, 2 * s, 3 * s], -1), "disp_map": s + 1, "acc_map": s + 2,
            "depth_map": s + 3, "rgb0": torch.stack([-s, -2 * s, -3 * s], -1), "disp0": s + 4, "acc0": s + 5,
            "z_std": s + 6, "raw": torch.zeros(rays.shape[0], 4, 4)}


def main():
    out_path = sys.argv[1]
    rank, world, device = ndist.init_from_env("gloo")
    assert ndist.world() == world and ndist.rank() == rank
    res = {"rank": rank, "world": world, "cases": {}}
    for n in (1, 2, 5, 64, 1001):
        g = torch.Generator().manual_seed(n)
        rays = torch.randn(n, 11, generator=g)
        calls = []

        def fn(r):
            calls.append(r.shape[0])
            return fake_render(r)

        got = ndist.render_rays_sharded(rays, fn)
        want = fake_render(rays)
        ok = all(torch.equal(got[k], want[k]) for k in got) and sorted(got) == sorted(k for k in want if k != "raw")
        lo, hi, per = ndist.shard_bounds(n, rank, world)
        res["cases"][str(n)] = {"ok": bool(ok), "rendered": calls, "shard": [lo, hi, per]}
    # explicit key subset (coarse-only renders have no *0 maps)
    rays = torch.randn(7, 8, generator=torch.Generator().manual_seed(3))
    got = ndist.render_rays_sharded(rays, fake_render, keys=["rgb_map", "depth_map"])
    res["subset_ok"] = bool(sorted(got) == ["depth_map", "rgb_map"] and torch.equal(got["rgb_map"], fake_render(rays)["rgb_map"]))
    res["objects"] = ndist.objects_of_rank(5, rank, world)
    # data-parallel gradient exchange: the mean over ranks of per-rank mean-loss gradients equals the gradient of
    # the global mean loss for equal shards
    torch.manual_seed(0)
    nets = [torch.nn.Linear(5, 3), torch.nn.Linear(3, 2)]
    x = torch.randn(8, 5, generator=torch.Generator().manual_seed(11))
    y = torch.randn(8, 2, generator=torch.Generator().manual_seed(12))
    xs, ys = ndist.shard_batch(x, y)
    ((nets[1](torch.relu(nets[0](xs))) - ys) ** 2).mean().backward()
    n = ndist.average_gradients(nets)
    got = torch.cat([p.grad.reshape(-1) for m in nets for p in m.parameters()])
    for m in nets:
        m.zero_grad()
    ((nets[1](torch.relu(nets[0](x))) - y) ** 2).mean().backward()
    want = torch.cat([p.grad.reshape(-1) for m in nets for p in m.parameters()])
    res["dp_ok"] = bool(n == got.numel() and torch.allclose(got, want, atol=1e-6))
    res["dp_shard"] = int(xs.shape[0])
    # ragged shards (15 rays: 8 + 7) and an empty shard (1 ray on 2 ranks): each rank weights its mean-loss gradient
    # by n_local / n_global before the sum; a rank without rays contributes zeros of the same length
    res["dp_ragged"] = {}
    for total in (15, 1):
        x = torch.randn(total, 5, generator=torch.Generator().manual_seed(21))
        y = torch.randn(total, 2, generator=torch.Generator().manual_seed(22))
        for m in nets:
            m.zero_grad(set_to_none=True)
        xs, ys = ndist.shard_batch(x, y)
        if xs.shape[0]:
            ((nets[1](torch.relu(nets[0](xs))) - ys) ** 2).mean().backward()
        n = ndist.average_gradients(nets, n_local=xs.shape[0], n_global=total)
        got = torch.cat([p.grad.reshape(-1) for m in nets for p in m.parameters()])
        for m in nets:
            m.zero_grad()
        ((nets[1](torch.relu(nets[0](x))) - y) ** 2).mean().backward()
        want = torch.cat([p.grad.reshape(-1) for m in nets for p in m.parameters()])
        res["dp_ragged"][str(total)] = {"ok": bool(n == got.numel() and torch.allclose(got, want, atol=1e-6)),
                                        "shard": int(xs.shape[0])}
    # the copy-free exchange: gradients live in one flat buffer, each module's slice is all-reduced in place, the
    # second module's as soon as it is reported (what the HIP backward does for the network that finishes first)
    ex = ndist.GradientExchange(nets)
    res["ex"] = {"views": bool(all(p.grad.data_ptr() >= ex.flat.data_ptr() and
                                   p.grad.data_ptr() < ex.flat.data_ptr() + ex.flat.numel() * 4 for m in nets for p in m.parameters()))}
    for total in (8, 15, 1):
        x = torch.randn(total, 5, generator=torch.Generator().manual_seed(31))
        y = torch.randn(total, 2, generator=torch.Generator().manual_seed(32))
        xs, ys = ndist.shard_batch(x, y)
        ex.begin(n_local=xs.shape[0], n_global=total)
        if xs.shape[0]:
            ((nets[1](torch.relu(nets[0](xs))) - ys) ** 2).mean().backward()      # autograd adds into the views
        ptrs = [p.grad.data_ptr() for m in nets for p in m.parameters()]
        ex.module_ready(1)                      # the head's gradient goes out first, asynchronously
        n = ex.finish()
        got = ex.flat.clone()
        same_storage = ptrs == [p.grad.data_ptr() for m in nets for p in m.parameters()]
        ref = [torch.nn.Linear(5, 3), torch.nn.Linear(3, 2)]
        for a, b in zip(ref, nets):
            a.load_state_dict(b.state_dict())
        ((ref[1](torch.relu(ref[0](x))) - y) ** 2).mean().backward()
        want = torch.cat([p.grad.reshape(-1) for m in ref for p in m.parameters()])
        res["ex"][str(total)] = {"ok": bool(n == want.numel() and torch.allclose(got, want, atol=1e-6)),
                                 "in_place": bool(same_storage), "shard": int(xs.shape[0])}
    # per-rank random streams: the ranks of one job must not draw the same jitter
    s_used = ndist.seed_ranks(1234)
    draw = torch.rand(4)
    torch.manual_seed(1234)
    res["seed"] = {"used": s_used, "draw": draw.tolist(), "rank0_draw": torch.rand(4).tolist()}
    torch.distributed.barrier()
    with open(out_path + ".%d" % rank, "w") as f:
        json.dump(res, f)
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
