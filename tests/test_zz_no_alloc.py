"""nerf_mlp_backward_static (the entry point for callers that cannot allocate: gradient tape and workspace in the
library's static device array) against nerf_mlp_backward with caller-provided buffers: identical gradients, bit
for bit, on three calls in a row (a smaller input after a larger one, then a larger one again); sizes beyond the limit
are refused; the call allocates no device memory and can be recorded into a HIP graph and replayed."""
import gc

import numpy as np
import pytest
import torch

from nerf_loc_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from nerf_loc_amd import hip
    hip.lib()
    return torch.device("cuda:0")


def _net(dev, vd):
    from nerf_loc_amd.helpers import NeRF
    arch = {} if vd else dict(use_viewdirs=False, output_ch=5, input_ch_views=0)
    net = NeRF(D=8, W=256, input_ch=63, output_ch=4 if vd else 5, skips=[4], input_ch_views=27 if vd else 0, use_viewdirs=vd)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(77, "A", **arch).items()})
    return net.to(dev)


@pytest.mark.parametrize("precision", ["fp32", "bf16", "bf16x3"])
@pytest.mark.parametrize("vd", [True, False])
def test_static_backward_equals_backward(dev, precision, vd):
    from nerf_loc_amd import hip, train
    net = _net(dev, vd)
    params = train._params(net)
    shapes = [None if p is None else tuple(p.shape) for p in params]
    prec = hip.PRECISIONS[precision]
    packed = net.packed_weights(prec)
    packed_bwd = hip.mlp_pack_backward([None if p is None else p.detach() for p in params], vd, net.out_channels, precision)
    rs = np.random.RandomState(3)
    for R, S in ((100, 64), (13, 64), (120, 64)):      # 6400 samples, then 832, then 7680: every call meets the array as
                                                       # the previous one left it
        rays = torch.from_numpy(synth.random_rays(R, seed=R)).to(dev)
        if not vd:
            rays = rays[:, :8].contiguous()
        from nerf_loc_amd import hip as h
        z = h.z_vals(rays, S)
        raw, tape = hip.mlp_forward_train(packed, vd, net.out_channels, hip.IN_RAYS, rays, z, R, S, rays.shape[1],
                                          precision=precision)
        d_raw = torch.from_numpy(rs.normal(size=(R * S, net.out_channels)).astype(np.float32)).to(dev)
        want = hip.mlp_backward(packed_bwd, vd, net.out_channels, d_raw, tape, shapes, precision=precision,
                                packed_fwd=packed)
        got = [None if s is None else torch.full(s, float("nan"), dtype=torch.float32, device=dev) for s in shapes]
        hip.mlp_backward_static(packed_bwd, vd, net.out_channels, d_raw, tape, got, precision=precision, packed_fwd=packed)
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(zip(got, want)):
            if b is None:
                assert a is None
                continue
            assert torch.isfinite(a).all(), i
            assert torch.equal(a, b), (precision, vd, R, i, float((a - b).abs().max()))
        assert float(want[0].abs().max()) > 0


def test_static_backward_refuses_more_than_its_limit(dev):
    from nerf_loc_amd import hip, train
    net = _net(dev, True)
    params = train._params(net)
    shapes = [tuple(p.shape) for p in params]
    packed = net.packed_weights(0)
    packed_bwd = hip.mlp_pack_backward([p.detach() for p in params], True, 4, "fp32")
    M = hip.STATIC_MAX_SAMPLES + 1
    feat = torch.zeros((M, 90), device=dev)
    raw, tape = hip.mlp_forward_train(packed, True, 4, hip.IN_FEATURES, feat, None, M, 1, 0)
    grads = [torch.zeros(s, device=dev) for s in shapes]
    with pytest.raises(hip.NerfHipError, match="exceed the static scratch"):
        hip.mlp_backward_static(packed_bwd, True, 4, torch.zeros((M, 4), device=dev), tape, grads)
    # the limit itself is served
    M = hip.STATIC_MAX_SAMPLES
    raw, tape = hip.mlp_forward_train(packed, True, 4, hip.IN_FEATURES, feat[:M].contiguous(), None, M, 1, 0)
    d_raw = torch.ones((M, 4), device=dev)
    want = hip.mlp_backward(packed_bwd, True, 4, d_raw, tape, shapes)
    hip.mlp_backward_static(packed_bwd, True, 4, d_raw, tape, grads)
    assert all(torch.equal(a, b) for a, b in zip(grads, want))


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_static_backward_allocates_nothing_and_replays_from_a_graph(dev, precision):
    """What the entry point exists for: after the first call on the device (which looks the static array's address up)
    a call neither allocates nor frees device memory (the driver's free-memory figure and torch's allocator stay as they
    were) and is legal under stream capture: recorded into a HIP graph and replayed on new gradients of the loss, it
    gives the bits of nerf_mlp_backward."""
    from nerf_loc_amd import hip, train
    net = _net(dev, True)
    params = train._params(net)
    shapes = [tuple(p.shape) for p in params]
    prec = hip.PRECISIONS[precision]
    packed = net.packed_weights(prec)
    packed_bwd = hip.mlp_pack_backward([p.detach() for p in params], True, 4, precision)
    R, S = 40, 64
    rays = torch.from_numpy(synth.random_rays(R, seed=5)).to(dev)
    z = hip.z_vals(rays, S)
    raw, tape = hip.mlp_forward_train(packed, True, 4, hip.IN_RAYS, rays, z, R, S, 11, precision=precision)
    gen = torch.Generator(device=dev).manual_seed(1)
    d_raw = torch.randn(R * S, 4, device=dev, generator=gen)
    grads = [torch.zeros(s, device=dev) for s in shapes]
    hip.mlp_backward_static(packed_bwd, True, 4, d_raw, tape, grads, precision=precision, packed_fwd=packed)   # warm-up
    torch.cuda.synchronize()

    def requests():      # running totals of what torch's allocator was asked for: they never go down, so tensors of
        st = torch.cuda.memory_stats(dev)      # earlier tests that the collector frees meanwhile do not disturb them
        return st["allocation.all.allocated"], st["segment.all.allocated"]

    gc.collect()
    free0, asked0 = torch.cuda.mem_get_info(dev)[0], requests()
    hip.mlp_backward_static(packed_bwd, True, 4, d_raw, tape, grads, precision=precision, packed_fwd=packed)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info(dev)[0] == free0
    assert requests() == asked0
    # capture: any allocation, free or synchronising call inside would abort the capture
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        hip.mlp_backward_static(packed_bwd, True, 4, d_raw, tape, grads, precision=precision, packed_fwd=packed)
    d_raw.copy_(torch.randn(R * S, 4, device=dev, generator=gen))          # new upstream gradients, same buffers
    for g in grads:
        g.fill_(float("nan"))
    graph.replay()
    torch.cuda.synchronize()
    want = hip.mlp_backward(packed_bwd, True, 4, d_raw, tape, shapes, precision=precision, packed_fwd=packed)
    for i, (a, b) in enumerate(zip(grads, want)):
        assert torch.equal(a, b), (precision, i, float((a - b).abs().max()))
