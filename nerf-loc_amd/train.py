"""Differentiable entry points: what the training step of run_nerf.py:770-796
needs from the render path (render(..., retraw=True) -> img2mse -> backward).

The reference gets gradients from PyTorch autograd over eager ops.  Here two
torch.autograd.Function nodes wrap hand-written HIP kernels:

  _NetworkFn    NeRF.forward / run_network: float32 forward that keeps its
                activations, then input-gradient chain + weight-gradient GEMMs
                (gradients for the 24 parameter tensors; none for the inputs:
                sample positions and view directions have no leaves behind them)
  _CompositeFn  raw2outputs: closed-form backward of the alpha compositing

Training computes in float32 (like the reference) unless another mode is asked
for explicitly (set_train_precision(...) or a model's ``train_precision``
attribute): "bf16x3" evaluates the forward pass of the float32 network in three
bf16 terms on the bf16 matrix cores (float32 accuracy, same tape, the float32
backward kernels); "bf16" uses bf16 MFMA operands with float32 accumulation
throughout, parameters and gradients stay float32.  The inference ``precision``
attribute plays no part.
The architecture is the one the
inference kernels implement (D=8, W=256, skips=[4], with view directions or
without and output_ch 4 / 5); others raise NotImplementedError.  Nothing here
falls back to autograd over torch ops.
"""
import numpy as np
import torch

from . import hip

_TRUNK = ["pts_linears.%d.%s" % (i, k) for i in range(8) for k in ("weight", "bias")]
_PARAM_ORDER = {True: _TRUNK + ["views_linears.0.weight", "views_linears.0.bias", "feature_linear.weight",
                                "feature_linear.bias", "alpha_linear.weight", "alpha_linear.bias", "rgb_linear.weight",
                                "rgb_linear.bias"],
                # without view directions views_linears.0 exists in the module but is never used: no gradient
                False: _TRUNK + [None, None, "output_linear.weight", "output_linear.bias"]}


TRAIN_PRECISIONS = ("fp32", "bf16x3", "bf16")
_TRAIN_PRECISION = ["fp32"]


def set_train_precision(name):
    """'fp32' (default: what the reference computes), 'bf16x3' or 'bf16' for networks without their own
    ``train_precision``."""
    if name not in TRAIN_PRECISIONS:
        raise ValueError("unknown training precision %r" % (name,))
    _TRAIN_PRECISION[0] = name


def train_precision_of(model):
    name = getattr(model, "train_precision", None) or _TRAIN_PRECISION[0]
    if name not in TRAIN_PRECISIONS:
        raise ValueError("unknown training precision %r" % (name,))
    return name


def _params(model):
    """Parameters in the kernels' tensor order (None for unused slots)."""
    if not model.hip_supported():
        raise NotImplementedError("training kernels implement D=8, W=256, skips=[4], input_ch=63 (and input_ch_views=27 "
                                  "with view directions, output_ch 4 or 5 without)")
    named = dict(model.named_parameters())
    return [None if k is None else named[k] for k in _PARAM_ORDER[bool(model.use_viewdirs)]]


def _packed_backward(model, params, precision):
    """Transposed-weight stream for the input-gradient chain, cached per weight version."""
    key = (hip.WEIGHT_GENERATION[0],) + tuple((p.data_ptr(), p._version) for p in params if p is not None)
    hit = model._pack_cache.get(("bwd", precision))
    if hit is not None and hit[0] == key:
        return hit[1]
    packed = hip.mlp_pack_backward([None if p is None else p.detach() for p in params], model.use_viewdirs,
                                   model.out_channels, precision)
    model._pack_cache[("bwd", precision)] = (key, packed)
    return packed


class _NetworkFn(torch.autograd.Function):
    """inputs: model, slot map, input mode and tensors, then the parameter tensors that exist (unused slots of the
    kernels' tensor order are left out: autograd cannot take None where a gradient is returned)."""

    @staticmethod
    def forward(ctx, model, slots, mode, in0, in1, R, S, ray_cols, *params):
        full = [None] * len(slots)
        for p, i in zip(params, [i for i, used in enumerate(slots) if used]):
            full[i] = p
        prec = train_precision_of(model)
        packed = model.packed_weights(hip.PRECISIONS[prec])
        raw, tape = hip.mlp_forward_train(packed, model.use_viewdirs, model.out_channels, mode, in0, in1, R, S, ray_cols,
                                          precision=prec)
        ctx.tape = tape
        ctx.prec = prec
        ctx.model = model
        ctx.slots = slots
        ctx.param_list = full
        ctx.packed_fwd = packed
        ctx.packed_bwd = _packed_backward(model, full, prec)
        ctx.shapes = [None if p is None else tuple(p.shape) for p in full]
        ctx.arch = (model.use_viewdirs, model.out_channels)
        return raw

    @staticmethod
    def backward(ctx, g_raw):
        # A module whose gradients live in a dist.GradientExchange buffer and still wait for this step's gradient: the
        # kernels write straight into the parameters' .grad views (no allocation, no accumulation pass) and the
        # exchange is told, so that it can start this network's all-reduce under the other network's backward.
        slot = getattr(ctx.model, "_grad_arena", None)
        if slot is not None:
            views = [None if p is None else slot.view_of(p) for p in ctx.param_list]
            direct = all((p is None) == (v is None) for p, v in zip(ctx.param_list, views)) and \
                (slot.fresh or all(p is None or p.grad is None for p in ctx.param_list))
            if direct:
                hip.mlp_backward(ctx.packed_bwd, ctx.arch[0], ctx.arch[1], g_raw.contiguous(), ctx.tape, ctx.shapes,
                                 precision=ctx.prec, packed_fwd=ctx.packed_fwd, out=views)
                ctx.tape = None
                slot.attach()
                slot.written()
                return (None,) * (8 + sum(1 for p in ctx.param_list if p is not None))
        grads = hip.mlp_backward(ctx.packed_bwd, ctx.arch[0], ctx.arch[1], g_raw.contiguous(), ctx.tape, ctx.shapes,
                                 precision=ctx.prec, packed_fwd=ctx.packed_fwd)
        ctx.tape = None
        return (None,) * 8 + tuple(g for g in grads if g is not None)


# Samples one network call may carry in bf16x3: its kernels address the tapes with 32-bit byte offsets (1 KiB rows).
MAX_SAMPLES_BF16X3 = 1 << 22


def _network(model, mode, in0, in1, R, S, ray_cols):
    params = _params(model)
    slots = tuple(p is not None for p in params)
    live = [p for p in params if p is not None]
    if train_precision_of(model) == "bf16x3" and R * S > MAX_SAMPLES_BF16X3:
        # more than one call's worth (the reference's batches are 1024-4096 rays; this is a 20 000-ray batch): whole rays
        # per piece, one autograd node each, their gradients accumulate like any other
        step = max(1, MAX_SAMPLES_BF16X3 // S)
        outs = []
        for lo in range(0, R, step):
            n = min(step, R - lo)
            if mode == hip.IN_RAYS:
                a, b = in0[lo:lo + n], in1[lo:lo + n]
            elif mode == hip.IN_POINTS:
                a, b = in0[lo * S:(lo + n) * S], (None if in1 is None else in1[lo:lo + n])
            else:
                a, b = in0[lo * S:(lo + n) * S], None
            outs.append(_NetworkFn.apply(model, slots, mode, a.contiguous(), None if b is None else b.contiguous(), n, S,
                                         ray_cols, *live))
        return torch.cat(outs, 0)
    return _NetworkFn.apply(model, slots, mode, in0, in1, R, S, ray_cols, *live)


class _CompositeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, z_vals, d, d_stride, noise, white_bkgd):
        rgb, disp, acc, weights, depth = hip.raw2outputs(raw, z_vals, d, d_stride, noise, white_bkgd)
        ctx.save_for_backward(raw, z_vals, d)
        ctx.misc = (d_stride, noise, white_bkgd)
        ctx.set_materialize_grads(False)
        return rgb, disp, acc, weights, depth

    @staticmethod
    def backward(ctx, g_rgb, g_disp, g_acc, g_weights, g_depth):
        raw, z_vals, d = ctx.saved_tensors
        d_stride, noise, white_bkgd = ctx.misc
        d_raw = hip.raw2outputs_backward(raw, z_vals, d, d_stride, noise, white_bkgd, g_rgb, g_disp, g_acc, g_depth,
                                         None if g_weights is None else g_weights.contiguous())
        return d_raw, None, None, None, None, None


def _no_input_grad(t, what):
    if isinstance(t, torch.Tensor) and t.requires_grad:
        raise NotImplementedError("%s requires grad: the HIP training kernels produce gradients for the network "
                                  "parameters only (the reference never optimises rays or poses)" % what)


# ---------------------------------------------------------------------------- entry points used by helpers / render
def mlp_forward_autograd(model, x):
    """NeRF.forward on encoded features [..., 90] with parameter gradients."""
    _no_input_grad(x, "the encoded network input")
    ch = 90 if model.use_viewdirs else 63
    flat = x.reshape(-1, x.shape[-1])[:, :ch].contiguous().float()
    out = _network(model, hip.IN_FEATURES, flat, None, flat.shape[0], 1, 0)
    return out.reshape(tuple(x.shape[:-1]) + (model.out_channels,))


def run_network_autograd(inputs, viewdirs, fn):
    """run_network (run_nerf.py:37-51) with the encodings fused, parameter gradients only."""
    _no_input_grad(inputs, "inputs")
    _no_input_grad(viewdirs, "viewdirs")
    S = inputs.shape[-2] if inputs.dim() >= 2 else 1
    pts = inputs.reshape(-1, 3).contiguous().float()
    R = pts.shape[0] // S
    vd = None
    if fn.use_viewdirs:
        vd = viewdirs.reshape(-1, 3).contiguous().float()
        if vd.shape[0] != R:
            raise ValueError("viewdirs has %d rows for %d rays" % (vd.shape[0], R))
    out = _network(fn, hip.IN_POINTS, pts, vd, R, S, 0)
    return out.reshape(list(inputs.shape[:-1]) + [fn.out_channels])


def raw2outputs_autograd(raw, z_vals, rays_d, raw_noise_std, white_bkgd, pytest):
    from .render import _density_noise
    _no_input_grad(z_vals, "z_vals")
    _no_input_grad(rays_d, "rays_d")
    d = rays_d.reshape(-1, 3)
    if d.dtype != torch.float32 or d.stride(-1) != 1:
        d = d.float().contiguous()
    noise = _density_noise(raw[..., 3].shape, raw_noise_std, pytest, raw.device)
    return _CompositeFn.apply(raw.contiguous(), z_vals.contiguous(), d, d.stride(0), noise, bool(white_bkgd))


def render_rays_autograd(rays, z_vals, network_fn, network_query_fn, N_samples, retraw, perturb, N_importance,
                         network_fine, white_bkgd, raw_noise_std, pytest, with_depth):
    """render_rays (run_nerf.py:314-424) when a network requires grad.  Same launches as the inference path
    plus the tape; the importance samples are not differentiated (z_samples.detach(), run_nerf.py:399)."""
    from .render import NetworkQuery, _density_noise
    from .helpers import _pdf_variates
    _no_input_grad(rays, "ray_batch")
    N_rays, cols = rays.shape
    dev = rays.device
    has_vd = cols > 8
    viewdirs = rays[:, -3:] if has_vd else None
    rays_d = rays[:, 3:6]
    fused = isinstance(network_query_fn, NetworkQuery)

    def query(net, z):
        needs = isinstance(net, torch.nn.Module) and any(p.requires_grad for p in net.parameters())
        if fused and network_query_fn.fusable(net, has_vd):
            if needs:
                out = _network(net, hip.IN_RAYS, rays, z, N_rays, z.shape[1], cols)
            else:
                prec = net.precision_code()
                out = hip.mlp_forward(net.packed_weights(prec), prec, net.use_viewdirs, net.out_channels, hip.IN_RAYS,
                                      rays, z, N_rays, z.shape[1], cols)
            return out.reshape(N_rays, z.shape[1], out.shape[-1])
        pts = rays[:, None, 0:3] + rays_d[:, None, :] * z[:, :, None]
        return network_query_fn(pts, viewdirs, net)

    def composite(raw, z):
        noise = _density_noise(raw[..., 3].shape, raw_noise_std, pytest, dev)
        if raw.requires_grad:
            return _CompositeFn.apply(raw.contiguous(), z, rays_d, cols, noise, bool(white_bkgd))
        return hip.raw2outputs(raw, z, rays_d, cols, noise, white_bkgd)

    raw = query(network_fn, z_vals)
    rgb_map, disp_map, acc_map, weights, depth_map = composite(raw, z_vals)
    ret0 = None
    z_std = None
    if N_importance > 0:
        ret0 = (rgb_map, disp_map, acc_map)
        u = _pdf_variates((N_rays,), N_importance, perturb == 0., pytest, dev)
        _, _, z_vals, z_std = hip.sample_pdf(z_vals, weights.detach()[:, 1:-1], u, bins_are_z=True, z_vals=z_vals,
                                             want_samples=False, want_sorted=True, want_std=True)
        run_fn = network_fn if network_fine is None else network_fine
        raw = query(run_fn, z_vals)
        rgb_map, disp_map, acc_map, weights, depth_map = composite(raw, z_vals)

    ret = {'rgb_map': rgb_map, 'disp_map': disp_map, 'acc_map': acc_map}
    if with_depth:
        ret['depth_map'] = depth_map
    if retraw:
        ret['raw'] = raw
    if ret0 is not None:
        ret['rgb0'], ret['disp0'], ret['acc0'] = ret0
        ret['z_std'] = z_std
    return ret


# ---------------------------------------------------------------------------- optimizer (run_nerf.py:225, 787)
class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam as the reference uses it (no amsgrad, no maximize), one kernel launch per step for all
    parameter tensors.  param_groups and state_dict() have torch.optim.Adam's layout ('step', 'exp_avg',
    'exp_avg_sq'), so the reference's learning-rate update loop and its checkpoints work unchanged."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            ps = [p for p in group['params'] if p.grad is not None]
            if not ps:
                continue
            for p in ps:
                st = self.state[p]
                if len(st) == 0:
                    st['step'] = torch.tensor(0.0)
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['step'] += 1
            # torch.optim.Adam keeps a step count per parameter (a tensor that first receives a gradient later, or a
            # checkpoint with differing counts): one launch per distinct count, normally exactly one
            b1, b2 = group['betas']
            by_step = {}
            for p in ps:
                by_step.setdefault(int(self.state[p]['step']), []).append(p)
            for step, qs in by_step.items():
                hip.adam_step([p.data for p in qs], [p.grad.contiguous() for p in qs],
                              [self.state[p]['exp_avg'] for p in qs], [self.state[p]['exp_avg_sq'] for p in qs],
                              step, group['lr'], b1, b2, group['eps'], group['weight_decay'])
        return loss


# ---------------------------------------------------------------------------- the step itself (run_nerf.py:770-796)
def train_step(rays, target, render_kwargs, optimizer, global_step, lrate, lrate_decay, chunk=1024 * 32, exchange=None,
               n_global=None):
    """One optimisation step as the reference's loop body does it: render the ray batch [N, 8|11], loss =
    img2mse(rgb, target) (+ img2mse(rgb0, target)), backward, optimizer.step(), then the exponential learning-rate
    decay 0.1 ** (global_step / (lrate_decay * 1000)).  Returns (loss, psnr) as tensors.
    exchange: a dist.GradientExchange over the networks (data-parallel training, or just allocation-free gradients on
    one GPU); `rays` is then this rank's shard of a global batch of n_global rays (default: equal shards)."""
    from .helpers import img2mse, mse2psnr
    from .render import batchify_rays
    kw = {k: v for k, v in render_kwargs.items() if k not in ("near", "far", "use_viewdirs", "ndc")}
    ret = batchify_rays(rays, chunk, **kw)
    if exchange is None:
        optimizer.zero_grad()
    else:
        exchange.begin(*(() if n_global is None else (rays.shape[0], n_global)))
    img_loss = img2mse(ret['rgb_map'], target)
    loss = img_loss
    psnr = mse2psnr(img_loss.detach())
    if 'rgb0' in ret:
        loss = loss + img2mse(ret['rgb0'], target)
    loss.backward()
    if exchange is not None:
        exchange.finish()
    optimizer.step()
    new_lrate = lrate * (0.1 ** (global_step / (lrate_decay * 1000)))
    for group in optimizer.param_groups:
        group['lr'] = new_lrate
    return loss.detach(), psnr
