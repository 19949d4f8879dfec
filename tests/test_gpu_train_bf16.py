"""GPU parity of the bf16 training kernels (bf16 MFMA operands, float32 accumulation, float32 parameters and
gradients).  The checker is the oracle's bf16 restatement of the network backward (every matrix-product operand
rounded to bfloat16); the float32 kernels' gradients are printed beside it: they differ by about 10 % in the first
layers because the bf16 forward switches a different set of ReLU units on, not because of the backward arithmetic.
The reference trains in float32 only, so no reference fixture exists for this mode."""
import numpy as np
import pytest
import torch

from nerf_loc_amd import synth
from oracle import nerf_oracle as orc
from test_gpu_train import T, N, load_net, dev, _query, F32   # noqa: F401

pytestmark = pytest.mark.gpu


def _grads(net, feat, d_raw, prec):
    net.train_precision = prec
    net.zero_grad()
    out = net(feat)
    out.backward(d_raw)
    return out.detach(), {k: N(p.grad).astype(np.float64) for k, p in net.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("M,vd,och", [(77, True, 4), (1000, True, 4), (40000, True, 4), (77, False, 4), (1000, False, 4),
                                      (40000, False, 4), (77, False, 5), (1000, False, 5), (40000, False, 5),
                                      (70001, True, 4)])      # the last one: some workgroups run two passes
def test_bf16_network_backward(dev, M, vd, och):
    sd = synth.make_state_dict(70 + M % 7, "B") if vd else synth.make_state_dict(70 + M % 7, "B", use_viewdirs=False, output_ch=och, input_ch_views=0)
    net = load_net(sd, dev, use_viewdirs=vd, output_ch=och)
    rs = np.random.RandomState(M)
    x = rs.uniform(-3, 3, size=(M, 3)).astype(F32)
    d = rs.normal(size=(M, 3))
    d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(F32)
    feat = np.concatenate([orc.embed(x, 10), orc.embed(d, 4)], -1) if vd else orc.embed(x, 10)
    d_raw = rs.randn(M, och).astype(F32)
    o32, g32 = _grads(net, T(feat, dev), T(d_raw, dev), "fp32")
    o16, g16 = _grads(net, T(feat, dev), T(d_raw, dev), "bf16")
    eo = float((o16 - o32).abs().max() / o32.abs().max())
    print("M=%d vd=%s och=%d: output rel err %.2e" % (M, vd, och, eo))
    worst = 0.0
    for k in g32:
        fro = float(np.linalg.norm(g16[k] - g32[k]) / max(np.linalg.norm(g32[k]), 1e-30))
        worst = max(worst, fro)
        print("   %-28s fro %.3e  |g32| %.3e |g16| %.3e" % (k, fro, np.linalg.norm(g32[k]), np.linalg.norm(g16[k])))
    assert eo < 2e-2
    ref = orc.mlp_backward(sd, feat, d_raw, use_viewdirs=vd, bf16=True)
    worst_e = 0.0
    for k in g16:
        fro = float(np.linalg.norm(g16[k] - ref[k]) / max(np.linalg.norm(ref[k]), 1e-30))
        worst_e = max(worst_e, fro)
        print("   vs bf16 restatement %-28s fro %.3e" % (k, fro))
    print("worst vs float32 kernels %.3e, vs bf16 restatement %.3e" % (worst, worst_e))
    assert worst_e < (2e-3 if M <= 1000 else 1e-2), worst_e


def test_bf16_optimisation_follows_float32(dev):
    """The same 12 Adam steps in both precisions: the losses stay together."""
    from nerf_loc_amd import train
    rays = T(synth.random_rays(512, seed=4), dev)
    rs = np.random.RandomState(4)
    target = T(rs.uniform(0, 1, size=(512, 3)).astype(F32), dev)
    curves = {}
    for prec in ("fp32", "bf16"):
        nc, nf = load_net(synth.make_state_dict(7, "A"), dev), load_net(synth.make_state_dict(8, "A"), dev)
        nc.train_precision = nf.train_precision = prec
        opt = train.FusedAdam(list(nc.parameters()) + list(nf.parameters()), lr=5e-4, betas=(0.9, 0.999))
        kw = dict(network_query_fn=_query(dev), perturb=0.0, N_importance=128, network_fine=nf, N_samples=64, network_fn=nc,
                  white_bkgd=True, raw_noise_std=0.0)
        curves[prec] = [float(train.train_step(rays, target, kw, opt, step + 1, 5e-4, 250)[0]) for step in range(12)]
        print(prec, " ".join("%.4f" % l for l in curves[prec]))
    a, b = np.array(curves["fp32"]), np.array(curves["bf16"])
    assert np.isfinite(b).all() and b.min() < 0.9 * b[0]      # (both curves overshoot at the 12th step on this batch)
    assert np.abs(a - b).max() < 0.02 * a[0]


def test_bf16_rejects_mismatched_streams(dev):
    """The bf16 backward needs the forward stream (activation scales): the binding refuses to run without it."""
    from nerf_loc_amd import hip
    with pytest.raises(ValueError):
        hip.mlp_backward(torch.zeros(16, dtype=torch.uint8, device=dev), True, 4, torch.zeros(4, 4, device=dev),
                         torch.zeros(4, device=dev), [None] * 24, precision="bf16", packed_fwd=None)
