"""Multi-process path on CPU: two ranks over gloo run the same sharding code the GPU job runs
over RCCL (contiguous ray shards, one all-gather of the packed per-ray record)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

import nerf_loc_amd
from nerf_loc_amd import dist as ndist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_every_ray_once():
    for n in (0, 1, 2, 7, 8, 9, 640000, 4194304):
        for g in (1, 2, 3, 4, 8):
            seen = 0
            for r in range(g):
                lo, hi, per = ndist.shard_bounds(n, r, g)
                assert lo == min(n, r * per) and lo <= hi <= n and hi - lo <= per
                assert per == -(-n // g)
                seen += hi - lo
            assert seen == n


def test_objects_are_dealt_round_robin():
    for n in (0, 1, 5, 16):
        for g in (1, 2, 8):
            flat = sorted(sum((ndist.objects_of_rank(n, r, g) for r in range(g)), []))
            assert flat == list(range(n))


def test_record_roundtrip():
    ret = {"rgb_map": torch.rand(6, 3), "disp_map": torch.rand(6), "acc_map": torch.rand(6), "depth_map": torch.rand(6),
           "rgb0": torch.rand(6, 3), "z_std": torch.rand(6)}
    keys = [k for k in ndist.RECORD_KEYS if k in ret]
    rec = ndist.pack_record(ret, keys)
    assert rec.shape == (6, 10) and rec.is_contiguous()
    back = ndist.unpack_record(rec, keys)
    assert all(torch.equal(back[k], ret[k]) for k in keys)
    assert ndist.all_gather_rows(rec, 6, 6) is rec        # world of one: no collective


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(600)
def test_world_size_two_gloo(tmp_path):
    port = _free_port()
    out = str(tmp_path / "res.json")
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PYTHONPATH=ROOT, OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"), out], env=env,
                                      cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=420)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    res = [json.load(open(out + ".%d" % r)) for r in range(2)]
    for r, x in enumerate(res):
        assert x["rank"] == r and x["world"] == 2 and x["subset_ok"]
        assert x["dp_ok"] and x["dp_shard"] == 4
        assert x["dp_ragged"]["15"] == {"ok": True, "shard": 8 - r} and x["dp_ragged"]["1"] == {"ok": True, "shard": 1 - r}
        for n, c in x["cases"].items():
            assert c["ok"], (r, n, c)
        assert x["ex"]["views"]
        for total, shard in (("8", 4), ("15", 8 - r), ("1", 1 - r)):
            assert x["ex"][total] == {"ok": True, "in_place": True, "shard": shard}, (r, total, x["ex"][total])
        assert x["seed"]["used"] == 1234 + r
    assert res[0]["seed"]["draw"] == res[0]["seed"]["rank0_draw"]         # rank 0 of a job = the single-process stream
    assert res[1]["seed"]["draw"] != res[0]["seed"]["draw"]
    # n = 1: rank 1 holds an empty shard and renders only a 1-ray probe for the key list
    assert res[0]["cases"]["1"]["rendered"] == [1] and res[1]["cases"]["1"]["shard"][:2] == [1, 1]
    assert res[0]["cases"]["1001"]["rendered"] == [501] and res[1]["cases"]["1001"]["rendered"] == [500]
    assert res[0]["objects"] == [0, 2, 4] and res[1]["objects"] == [1, 3]


def test_gradient_exchange_world_of_one():
    """Without a process group the exchange only provides the flat storage: gradients accumulate into views of one
    buffer, begin() clears it with one fill, finish() changes nothing."""
    torch.manual_seed(0)
    nets = [torch.nn.Linear(5, 3), torch.nn.Linear(3, 2)]
    ex = ndist.GradientExchange(nets)
    x, y = torch.randn(6, 5), torch.randn(6, 2)
    ex.begin()
    ((nets[1](torch.relu(nets[0](x))) - y) ** 2).mean().backward()
    got = ex.flat.clone()
    assert ex.finish() == got.numel() == sum(p.numel() for m in nets for p in m.parameters())
    assert torch.equal(ex.flat, got)
    for m in nets:
        m.zero_grad(set_to_none=True)
    ((nets[1](torch.relu(nets[0](x))) - y) ** 2).mean().backward()
    want = torch.cat([p.grad.reshape(-1) for m in nets for p in m.parameters()])
    assert torch.allclose(got, want, atol=1e-7)
    ex.begin()                                   # re-attaches the views that zero_grad(set_to_none) dropped
    assert float(ex.flat.abs().max()) == 0.0
    assert all(p.grad.data_ptr() >= ex.flat.data_ptr() for m in nets for p in m.parameters())
    with pytest.raises(ValueError):
        ex.begin(n_local=3)
    assert ndist.seed_ranks(7) == 7
