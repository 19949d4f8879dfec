"""Differentiable entry points (training step of run_nerf.py:770-796).

The backward kernels are not built yet: any call that would need a gradient
through the HIP ops raises instead of silently producing none.  Inference
(torch.no_grad(), or networks whose parameters do not require grad) never
reaches this module.
"""


def _no_backward(what):
    raise NotImplementedError(
        what + ": gradients through the HIP kernels are not implemented yet; run under torch.no_grad() "
        "or call .requires_grad_(False) on the network for inference")


def mlp_forward_autograd(model, x):
    _no_backward("NeRF.forward")


def run_network_autograd(inputs, viewdirs, fn):
    _no_backward("run_network")


def raw2outputs_autograd(*a, **k):
    _no_backward("raw2outputs")


def render_rays_autograd(*a, **k):
    _no_backward("render_rays")
