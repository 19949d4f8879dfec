"""Drop-in for the render path of nerf-pytorch/run_nerf.py (lines 27-424; the
same code is repeated in run_multi_nerf.py): batchify, run_network,
batchify_rays, render, render_path, create_nerf, raw2outputs, render_rays.

Signatures, dictionary keys, shapes and the meaning of every argument follow
the reference.  When the network, the embedders and the query function are the
ones this package provides (the normal case: create_nerf builds them), a ray
chunk is rendered by five kernel launches per pass with no intermediate but
z_vals, raw and the per-ray maps touching memory.  Foreign callables still work:
they are composed from the same HIP ops the way the reference composes torch ops.
"""
import os
import time

import numpy as np
import torch

from . import hip
from .helpers import NeRF, EmbedFn, get_embedder, to8b, _pdf_variates

DEBUG = False


# ---------------------------------------------------------------------------- run_nerf.py:27-51
def batchify(fn, chunk):
    """Constructs a version of 'fn' that applies to smaller batches."""
    if chunk is None:
        return fn

    def ret(inputs):
        return torch.cat([fn(inputs[i:i + chunk]) for i in range(0, inputs.shape[0], chunk)], 0)
    return ret


def _fusable(fn, embed_fn, embeddirs_fn, has_viewdirs):
    if not (isinstance(fn, NeRF) and fn.hip_supported() and isinstance(embed_fn, EmbedFn)):
        return False
    e = embed_fn.embedder
    if not (e.include_input and e.num_freqs == 10):
        return False
    if fn.use_viewdirs:
        if not (has_viewdirs and isinstance(embeddirs_fn, EmbedFn)):
            return False
        ed = embeddirs_fn.embedder
        return ed.include_input and ed.num_freqs == 4
    return not has_viewdirs


def _query_objects(nets, query_fn, rays, z, has_vd):
    """Several networks, each evaluating its equal share of the ray batch: one launch (nerf_mlp_forward_objects)."""
    n, N_rays, S, cols = len(nets), rays.shape[0], z.shape[1], rays.shape[1]
    first = nets[0]
    if N_rays % n:
        raise ValueError("%d rays do not split evenly over %d networks" % (N_rays, n))
    for net in nets:
        if not query_fn.fusable(net, has_vd):
            raise NotImplementedError("a batch of networks needs this package's NeRF modules and embedders")
        if (net.use_viewdirs, net.out_channels, net.precision_code()) != (first.use_viewdirs, first.out_channels,
                                                                          first.precision_code()):
            raise ValueError("the networks of a batch must share architecture and precision")
    prec = first.precision_code()
    if N_rays:
        out = hip.mlp_forward_objects([net.packed_weights(prec) for net in nets], prec, first.use_viewdirs,
                                      first.out_channels, hip.IN_RAYS, rays, z, N_rays, S, cols)
    else:
        out = torch.empty((0, first.out_channels), dtype=torch.float32, device=rays.device)
    return out.reshape(N_rays, S, first.out_channels)


def _needs_grad(fn):
    return torch.is_grad_enabled() and isinstance(fn, torch.nn.Module) and any(p.requires_grad for p in fn.parameters())


def run_network(inputs, viewdirs, fn, embed_fn, embeddirs_fn, netchunk=1024 * 64):
    """Prepares inputs and applies network 'fn' (run_nerf.py:37-51).
    inputs [..., S, 3] (+ viewdirs [..., 3]) -> [..., S, C].  With this package's
    embedders and NeRF the encoding, the concatenation and all layers run as one
    kernel and netchunk is irrelevant."""
    if _fusable(fn, embed_fn, embeddirs_fn, viewdirs is not None):
        if _needs_grad(fn):
            from . import train
            return train.run_network_autograd(inputs, viewdirs, fn)
        S = inputs.shape[-2] if inputs.dim() >= 2 else 1
        pts = inputs.reshape(-1, 3)
        R = pts.shape[0] // S
        vd = viewdirs.reshape(-1, 3) if viewdirs is not None else None
        if vd is not None and vd.shape[0] != R:
            raise ValueError("viewdirs has %d rows for %d rays" % (vd.shape[0], R))
        prec = fn.precision_code()
        out = hip.mlp_forward(fn.packed_weights(prec), prec, fn.use_viewdirs, fn.out_channels, hip.IN_POINTS,
                              pts, vd, R, S)
        return out.reshape(list(inputs.shape[:-1]) + [out.shape[-1]])
    inputs_flat = torch.reshape(inputs, [-1, inputs.shape[-1]])
    embedded = embed_fn(inputs_flat)
    if viewdirs is not None:
        input_dirs = viewdirs[:, None].expand(inputs.shape)
        input_dirs_flat = torch.reshape(input_dirs, [-1, input_dirs.shape[-1]])
        embedded = torch.cat([embedded, embeddirs_fn(input_dirs_flat)], -1)
    outputs_flat = batchify(fn, netchunk)(embedded)
    return torch.reshape(outputs_flat, list(inputs.shape[:-1]) + [outputs_flat.shape[-1]])


class NetworkQuery:
    """What create_nerf returns as network_query_fn: callable like the
    reference's lambda (inputs, viewdirs, network_fn) and recognisable by
    render_rays, which then skips materialising the sample points."""

    def __init__(self, embed_fn, embeddirs_fn, netchunk):
        self.embed_fn, self.embeddirs_fn, self.netchunk = embed_fn, embeddirs_fn, netchunk

    def __call__(self, inputs, viewdirs, network_fn):
        return run_network(inputs, viewdirs, network_fn, embed_fn=self.embed_fn,
                           embeddirs_fn=self.embeddirs_fn, netchunk=self.netchunk)

    def fusable(self, network_fn, has_viewdirs):
        return _fusable(network_fn, self.embed_fn, self.embeddirs_fn, has_viewdirs)


# ---------------------------------------------------------------------------- run_nerf.py:268-311
def _density_noise(shape, raw_noise_std, pytest, device):
    if not raw_noise_std > 0.:
        return None
    if pytest:   # the reference's hook draws UNIFORM numbers here (run_nerf.py:294-297)
        np.random.seed(0)
        return torch.Tensor(np.random.rand(*list(shape)) * raw_noise_std).to(device)
    return torch.randn(shape, device=device) * raw_noise_std


def raw2outputs(raw, z_vals, rays_d, raw_noise_std=0, white_bkgd=False, pytest=False):
    """Transforms model's predictions to semantically meaningful values
    (run_nerf.py:268-311).  Returns rgb_map [R,3], disp_map [R], acc_map [R],
    weights [R,S], depth_map [R]."""
    if _needs_grad_tensor(raw):
        from . import train
        return train.raw2outputs_autograd(raw, z_vals, rays_d, raw_noise_std, white_bkgd, pytest)
    d = rays_d.reshape(-1, 3)
    if d.dtype != torch.float32 or d.stride(-1) != 1:
        d = d.float().contiguous()
    noise = _density_noise(raw[..., 3].shape, raw_noise_std, pytest, raw.device)
    return hip.raw2outputs(raw, z_vals, d, d.stride(0), noise, white_bkgd)


def _needs_grad_tensor(t):
    return torch.is_grad_enabled() and isinstance(t, torch.Tensor) and t.requires_grad


# ---------------------------------------------------------------------------- run_nerf.py:314-424
def render_rays(ray_batch, network_fn, network_query_fn, N_samples, retraw=False, lindisp=False, perturb=0.,
                N_importance=0, network_fine=None, white_bkgd=False, raw_noise_std=0., verbose=False, pytest=False,
                _with_depth=False, _render_rgb=True, _render_depth=True):
    """Volumetric rendering of a ray batch [N, 8|11] (run_nerf.py:314-424).
    Returns the dict of the reference: rgb_map, disp_map, acc_map, raw (if
    retraw), rgb0, disp0, acc0, z_std (if N_importance > 0).  The underscore
    arguments serve the NeRF-Loc flavour (nerf_utils.py:248-342)."""
    if ray_batch.dim() != 2 or ray_batch.shape[-1] < 8:
        raise ValueError("ray_batch must be [N, 8] or [N, 11], got %s" % (tuple(ray_batch.shape),))
    if isinstance(network_fn, (list, tuple)):
        # one network per object: object o owns the o-th equal share of THIS batch, in both passes.  A single fine
        # network would render every object's rays with the same weights.
        if network_fine is not None and not (isinstance(network_fine, (list, tuple)) and len(network_fine) == len(network_fn)):
            raise ValueError("render_rays: with a list of %d coarse networks, network_fine must be None or a list of the "
                             "same length" % len(network_fn))
    elif isinstance(network_fine, (list, tuple)):
        raise ValueError("render_rays: a list of fine networks needs a list of coarse networks of the same length")
    rays = ray_batch if (ray_batch.dtype == torch.float32 and ray_batch.is_contiguous()) else ray_batch.float().contiguous()
    N_rays, cols = rays.shape
    dev = rays.device
    has_vd = cols > 8
    viewdirs = rays[:, -3:] if has_vd else None
    rays_d = rays[:, 3:6]

    t_rand = None
    if perturb > 0.:
        if pytest:
            np.random.seed(0)
            t_rand = torch.Tensor(np.random.rand(N_rays, N_samples)).to(dev)
        else:
            t_rand = torch.rand((N_rays, N_samples), device=dev)
    z_vals = hip.z_vals(rays, N_samples, lindisp, t_rand)

    fused = isinstance(network_query_fn, NetworkQuery)
    training = not isinstance(network_fn, (list, tuple)) and (
        _needs_grad(network_fn) or (network_fine is not None and _needs_grad(network_fine)))
    if training:
        from . import train
        return train.render_rays_autograd(rays, z_vals, network_fn, network_query_fn, N_samples, retraw, perturb,
                                          N_importance, network_fine, white_bkgd, raw_noise_std, pytest, _with_depth)

    def query(net, z):
        if isinstance(net, (list, tuple)):       # one network per object: equal consecutive shares of the ray batch
            if not fused:
                raise NotImplementedError("a batch of networks needs the NetworkQuery that create_nerf builds")
            return _query_objects(net, network_query_fn, rays, z, has_vd)
        if fused and network_query_fn.fusable(net, has_vd):
            prec = net.precision_code()
            out = hip.mlp_forward(net.packed_weights(prec), prec, net.use_viewdirs, net.out_channels, hip.IN_RAYS,
                                  rays, z, N_rays, z.shape[1], cols)
            return out.reshape(N_rays, z.shape[1], net.out_channels)
        pts = rays[:, None, 0:3] + rays_d[:, None, :] * z[:, :, None]
        return network_query_fn(pts, viewdirs, net)

    def composite(raw, z, want_weights):
        noise = _density_noise(raw[..., 3].shape, raw_noise_std, pytest, dev)
        return hip.raw2outputs(raw, z, rays_d, cols, noise, white_bkgd, want_weights=want_weights,
                               want_rgb=_render_rgb, want_depth=_render_depth)

    raw = query(network_fn, z_vals)
    rgb_map, disp_map, acc_map, weights, depth_map = composite(raw, z_vals, N_importance > 0)

    if N_importance > 0:
        rgb_map_0, disp_map_0, acc_map_0 = rgb_map, disp_map, acc_map
        # mid-points, sample_pdf on weights[..., 1:-1], sort(cat) and z_std in one launch
        u = _pdf_variates((N_rays,), N_importance, perturb == 0., pytest, dev)
        _, _, z_vals, z_std = hip.sample_pdf(z_vals, weights[:, 1:-1], u, bins_are_z=True, z_vals=z_vals,
                                             want_samples=False, want_sorted=True, want_std=True)
        run_fn = network_fn if network_fine is None else network_fine
        raw = query(run_fn, z_vals)
        rgb_map, disp_map, acc_map, weights, depth_map = composite(raw, z_vals, False)

    ret = {'rgb_map': rgb_map, 'disp_map': disp_map, 'acc_map': acc_map}
    if _with_depth:
        ret['depth_map'] = depth_map
    if retraw:
        ret['raw'] = raw
    if N_importance > 0:
        ret['rgb0'] = rgb_map_0
        ret['disp0'] = disp_map_0
        ret['acc0'] = acc_map_0
        ret['z_std'] = z_std
    if DEBUG:
        warn_nonfinite(ret)
    return ret


# 'now': the warning is printed before the call returns, like the reference's (one 4-byte-per-map read-back, which
# makes the host wait for the render).  'deferred': the counts are copied to pinned memory behind the render and
# printed by the next call that finds them complete (or by flush_nonfinite_reports()); the host never waits.
NONFINITE_REPORT = 'now'
_PENDING_REPORTS = []


def _print_nonfinite(keys, counts):
    for k, c in zip(keys, counts):
        if c:
            print(f"! [Numerical Error] {k} contains nan or inf.")


def flush_nonfinite_reports(wait=True):
    """Prints the deferred NaN / Inf warnings whose counts have arrived (all of them with wait=True)."""
    keep = []
    for ev, host, keys in _PENDING_REPORTS:
        if wait:
            ev.synchronize()
        if ev.query():
            _print_nonfinite(keys, host.tolist())
        else:
            keep.append((ev, host, keys))
    _PENDING_REPORTS[:] = keep


def warn_nonfinite(ret, skip=()):
    """The reference's NaN / Inf warning (run_nerf.py:420-422), from one census launch and one read-back."""
    keys = [k for k in ret if ret[k] is not None and k not in skip and ret[k].dtype == torch.float32 and ret[k].numel()]
    if _PENDING_REPORTS:
        flush_nonfinite_reports(wait=False)
    if not keys:
        return
    counts = hip.count_nonfinite([ret[k] for k in keys])
    if NONFINITE_REPORT == 'deferred':
        host = torch.empty(counts.shape, dtype=counts.dtype, pin_memory=True)
        host.copy_(counts, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(counts.device))
        _PENDING_REPORTS.append((ev, host, keys))
        return
    _print_nonfinite(keys, counts.tolist())


# ---------------------------------------------------------------------------- run_nerf.py:54-134
def batchify_rays(rays_flat, chunk=1024 * 32, **kwargs):
    """Render rays in smaller minibatches to bound memory (run_nerf.py:54-66)."""
    if isinstance(kwargs.get('network_fn'), (list, tuple)) and chunk < rays_flat.shape[0]:
        # every chunk would be re-split over all the networks: rays of one object rendered with another's weights
        raise ValueError("batchify_rays: a list of networks shares out the WHOLE ray batch (%d rays); chunk=%d would cut "
                         "it" % (rays_flat.shape[0], chunk))
    all_ret = {}
    for i in range(0, rays_flat.shape[0], chunk):
        ret = render_rays(rays_flat[i:i + chunk], **kwargs)
        for k in ret:
            if k not in all_ret:
                all_ret[k] = []
            all_ret[k].append(ret[k])
    return {k: (v[0] if len(v) == 1 else torch.cat(v, 0)) for k, v in all_ret.items() if v[0] is not None}


def _build_rays(H, W, K, rays, c2w, ndc, near, far, use_viewdirs, c2w_staticcam):
    """Ray source selection, view directions, NDC and [N, 8|11] packing of
    run_nerf.py:95-123 in one kernel.  Returns (rays_flat, leading shape).
    c2w_staticcam is consulted only with use_viewdirs, like the reference (run_nerf.py:102-107): geometry then
    comes from the static camera's H x W grid and the view directions from c2w or from the given rays."""
    static = c2w_staticcam if use_viewdirs else None
    if c2w is not None:
        c2w = torch.as_tensor(c2w)
        if static is not None:
            flat = hip.make_rays(H, W, K, c2w=torch.as_tensor(static).to(c2w.device), near=near, far=far,
                                 use_viewdirs=True, ndc=ndc, c2w_view=c2w)
        else:
            flat = hip.make_rays(H, W, K, c2w=c2w, near=near, far=far, use_viewdirs=use_viewdirs, ndc=ndc)
        return flat, [H, W]
    rays_o, rays_d = rays
    if static is not None:
        # the reference concatenates the static camera's H*W rays with the N given view directions: N must be H*W
        n = rays_d.reshape(-1, 3).shape[0]
        if n != H * W:
            raise ValueError("c2w_staticcam with a ray batch needs H*W = %d rays, got %d" % (H * W, n))
        view = hip.make_rays(H, W, K, rays_o=rays_o.float(), rays_d=rays_d.float(), use_viewdirs=True)[:, 8:11]
        geom = hip.make_rays(H, W, K, c2w=torch.as_tensor(static).to(view.device), near=near, far=far,
                             use_viewdirs=False, ndc=ndc)
        return torch.cat([geom, view], -1), [H, W]
    sh = list(rays_d.shape[:-1])
    flat = hip.make_rays(H, W, K, rays_o=rays_o.float(), rays_d=rays_d.float(), near=near, far=far,
                         use_viewdirs=use_viewdirs, ndc=ndc)
    return flat, sh


def render(H, W, K, chunk=1024 * 32, rays=None, c2w=None, ndc=False, near=0., far=1., use_viewdirs=False,
           c2w_staticcam=None, **kwargs):
    """Render rays (run_nerf.py:69-134).
    Args: H, W, K intrinsics; chunk: rays per launch (memory bound only, does not
    change results); rays: [2, batch, 3] origins/directions, or c2w: [3, 4] pose
    to render the full image; ndc; near, far: float or [batch]; use_viewdirs;
    c2w_staticcam.  Returns [rgb_map, disp_map, acc_map, extras]."""
    rays_flat, sh = _build_rays(H, W, K, rays, c2w, ndc, near, far, use_viewdirs, c2w_staticcam)
    all_ret = batchify_rays(rays_flat, chunk, **kwargs)
    for k in all_ret:
        all_ret[k] = torch.reshape(all_ret[k], sh + list(all_ret[k].shape[1:]))
    k_extract = ['rgb_map', 'disp_map', 'acc_map']
    ret_list = [all_ret[k] for k in k_extract]
    ret_dict = {k: all_ret[k] for k in all_ret if k not in k_extract}
    return ret_list + [ret_dict]


def render_path(render_poses, hwf, K, chunk, render_kwargs, gt_imgs=None, savedir=None, render_factor=0):
    """Renders one image per pose and returns (rgbs, disps) as numpy arrays
    (run_nerf.py:137-181; run_multi_nerf.py guards the 'retraw' lookups).  With
    savedir each frame is written as '{:06d}.png' and, with retraw, its raw volume
    as '{:06d}.npy'."""
    H, W, focal = hwf
    if render_factor != 0:
        H = H // render_factor
        W = W // render_factor
        focal = focal / render_factor
    rgbs, disps = [], []
    for i, c2w in enumerate(render_poses):
        rgb, disp, acc, extra = render(H, W, K, chunk=chunk, c2w=c2w[:3, :4], **render_kwargs)
        rgbs.append(rgb.cpu().numpy())
        disps.append(disp.cpu().numpy())
        if savedir is not None:
            _save_frame(savedir, i, rgbs[-1], extra.get('raw') if render_kwargs.get('retraw') else None)
    return np.stack(rgbs, 0), np.stack(disps, 0)


def _save_frame(savedir, i, rgb, raw):
    """The frame dump of run_nerf.py:170-175: '{:06d}.png' (8-bit) and, with retraw, '{:06d}.npy' of the raw volume.
    imageio is used when installed, else the package's own PNG writer."""
    path = os.path.join(savedir, '{:06d}.png'.format(i))
    try:
        import imageio
        imageio.imwrite(path, to8b(rgb))
    except ImportError:
        from .poses import write_png
        write_png(path, to8b(rgb))
    if raw is not None:
        np.save(os.path.join(savedir, '{:06d}.npy'.format(i)), raw.cpu().numpy())


def write_video(path, frames, fps=30, quality=8):
    """The video dumps of run_nerf.py:681, 819-820: imageio.mimwrite(path, to8b(frames), fps=30, quality=8).  frames:
    float [N,H,W,3] in [0,1] (or uint8).  With imageio installed this is that call; without it an AVI file is written
    next to the requested name (same stem, '.avi': uncompressed, or Motion-JPEG at quality 10 x `quality` when Pillow
    is importable).  Returns the path written."""
    frames8 = frames if getattr(frames, "dtype", None) == np.uint8 else to8b(np.asarray(frames))
    try:
        import imageio
        imageio.mimwrite(path, frames8, fps=fps, quality=quality)
        return path
    except ImportError:
        from .poses import write_avi
        out = os.path.splitext(path)[0] + ".avi"
        write_avi(out, frames8, fps=fps, jpeg_quality=min(100, 10 * int(quality)) if quality else None)
        return out


# ---------------------------------------------------------------------------- run_nerf.py:184-265
def create_nerf(args, device=None):
    """Instantiate NeRF's MLP model(s), the query function, the optimizer, and
    reload the newest checkpoint (run_nerf.py:184-265).  `args` carries the same
    attributes as the reference's parser: multires, i_embed, use_viewdirs,
    multires_views, N_importance, netdepth, netwidth, netdepth_fine,
    netwidth_fine, netchunk, lrate, basedir, expname, ft_path, no_reload,
    perturb, N_samples, white_bkgd, raw_noise_std, dataset_type, no_ndc, lindisp."""
    if device is None:
        device = torch.device("cuda")
    embed_fn, input_ch = get_embedder(args.multires, args.i_embed)
    input_ch_views = 0
    embeddirs_fn = None
    if args.use_viewdirs:
        embeddirs_fn, input_ch_views = get_embedder(args.multires_views, args.i_embed)
    output_ch = 5 if args.N_importance > 0 else 4
    skips = [4]
    model = NeRF(D=args.netdepth, W=args.netwidth, input_ch=input_ch, output_ch=output_ch, skips=skips,
                 input_ch_views=input_ch_views, use_viewdirs=args.use_viewdirs).to(device)
    grad_vars = list(model.parameters())
    model_fine = None
    if args.N_importance > 0:
        model_fine = NeRF(D=args.netdepth_fine, W=args.netwidth_fine, input_ch=input_ch, output_ch=output_ch,
                          skips=skips, input_ch_views=input_ch_views, use_viewdirs=args.use_viewdirs).to(device)
        grad_vars += list(model_fine.parameters())
    # not a reference option: args.train_precision = "bf16" trains with bf16 MFMA operands (float32 otherwise)
    train_precision = getattr(args, 'train_precision', None)
    if train_precision not in (None, "fp32", "bf16"):
        raise ValueError("train_precision must be 'fp32' or 'bf16', got %r" % (train_precision,))
    for m in (model, model_fine):
        if m is not None and train_precision is not None:
            m.train_precision = train_precision
    network_query_fn = NetworkQuery(embed_fn, embeddirs_fn, args.netchunk)
    from .train import FusedAdam      # torch.optim.Adam's update and state layout, one launch per step
    optimizer = FusedAdam(params=grad_vars, lr=args.lrate, betas=(0.9, 0.999))

    start = 0
    basedir, expname = args.basedir, args.expname
    ft_path = getattr(args, 'ft_path', None)
    if ft_path is not None and ft_path != 'None':
        ckpts = [ft_path]
    else:
        d = os.path.join(basedir, expname)
        ckpts = [os.path.join(d, f) for f in sorted(os.listdir(d)) if 'tar' in f] if os.path.isdir(d) else []
    if len(ckpts) > 0 and not args.no_reload:
        start = load_checkpoint(ckpts[-1], model, model_fine, optimizer, device)

    render_kwargs_train = {
        'network_query_fn': network_query_fn,
        'perturb': args.perturb,
        'N_importance': args.N_importance,
        'network_fine': model_fine,
        'N_samples': args.N_samples,
        'network_fn': model,
        'use_viewdirs': args.use_viewdirs,
        'white_bkgd': args.white_bkgd,
        'raw_noise_std': args.raw_noise_std,
    }
    if args.dataset_type != 'llff' or args.no_ndc:
        render_kwargs_train['ndc'] = False
        render_kwargs_train['lindisp'] = args.lindisp
    render_kwargs_test = {k: render_kwargs_train[k] for k in render_kwargs_train}
    render_kwargs_test['perturb'] = False
    render_kwargs_test['raw_noise_std'] = 0.
    return render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer


# ---------------------------------------------------------------------------- checkpoints (run_nerf.py:221-239, 803-811)
def load_checkpoint(path, model, model_fine=None, optimizer=None, device=None):
    """Reads a nerf-pytorch '.tar' checkpoint: keys global_step,
    network_fn_state_dict, network_fine_state_dict, optimizer_state_dict.
    Only tensors and plain containers are accepted (weights_only=True)."""
    ckpt = torch.load(path, map_location=device, weights_only=True)
    model.load_state_dict(ckpt['network_fn_state_dict'])
    if model_fine is not None and ckpt.get('network_fine_state_dict') is not None:
        model_fine.load_state_dict(ckpt['network_fine_state_dict'])
    if optimizer is not None and ckpt.get('optimizer_state_dict') is not None:
        optimizer.load_state_dict(ckpt['optimizer_state_dict'])
    return ckpt['global_step']


def save_checkpoint(path, global_step, model, model_fine, optimizer):
    """Writes the layout of run_nerf.py:803-811 (a missing fine network is stored
    as None instead of crashing)."""
    torch.save({
        'global_step': global_step,
        'network_fn_state_dict': model.state_dict(),
        'network_fine_state_dict': model_fine.state_dict() if model_fine is not None else None,
        'optimizer_state_dict': optimizer.state_dict() if optimizer is not None else None,
    }, path)
