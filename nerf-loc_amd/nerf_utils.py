"""Drop-in for nerf_loc_coarse_fine/utils/nerf_utils.py: the private copy of the
render path that the NeRF-Loc detector uses to produce its rendered_rgb /
rendered_depth / raw-volume inputs (nerf_utils.py:16-395).

Differences of this flavour from run_nerf.py, all kept: the network is
hard-wired (D=8, W=256, L=10/4, view directions, 64 samples, no fine pass);
render() returns [rgb, disp, acc, depth, extras]; render_rays adds depth_map and
honours nerf_render_rgb / nerf_render_depth; render_path returns numpy arrays
(rgbs, disps, raws, depths) and always asks for the raw volume.
"""
import numpy as np
import torch

from . import hip
from . import render as _r
from .helpers import NeRF, get_embedder, get_rays, to8b, ndc_rays, sample_pdf  # re-exported names
from .render import run_network, NetworkQuery, load_checkpoint


def create_nerf_model(args, ckpt_path=None, device='cuda:0', near=0.1, far=9.):
    """nerf_utils.py:16-85.  `args` only needs nerf_render_rgb / nerf_render_depth."""
    embed_fn, input_ch = get_embedder(10, 10)
    embeddirs_fn, input_ch_views = get_embedder(4, 10)
    model = NeRF(D=8, W=256, input_ch=input_ch, output_ch=4, skips=[4], input_ch_views=input_ch_views,
                 use_viewdirs=True).to(device)
    network_query_fn = NetworkQuery(embed_fn, embeddirs_fn, 1024 * 64)
    render_kwargs_train = {
        'network_query_fn': network_query_fn,
        'perturb': 1,
        'N_importance': 0,
        'network_fine': None,
        'N_samples': 64,
        'network_fn': model,
        'use_viewdirs': True,
        'white_bkgd': False,
        'raw_noise_std': 1e0,
        'nerf_render_rgb': args.nerf_render_rgb,
        'nerf_render_depth': args.nerf_render_depth,
        'ndc': False,
    }
    render_kwargs_test = {k: render_kwargs_train[k] for k in render_kwargs_train}
    render_kwargs_test['perturb'] = False
    render_kwargs_test['raw_noise_std'] = 0.
    render_kwargs_test['retraw'] = True
    for kw in (render_kwargs_train, render_kwargs_test):
        kw.update({'near': near, 'far': far})
    start = 0
    if ckpt_path is not None:
        render_kwargs_test, start = load_nerf_model(render_kwargs_test, ckpt_path)
    return render_kwargs_train, render_kwargs_test, start


def load_nerf_model(render_kwargs_test, ckpt_path):
    """nerf_utils.py:88-105: ckpt_path is a LIST of checkpoint files; the last one
    is loaded into network_fn (only network_fn_state_dict is used)."""
    start = 0
    ckpts = ckpt_path
    if len(ckpts) > 0:
        net = render_kwargs_test['network_fn']
        dev = next(net.parameters()).device
        ckpt = torch.load(ckpts[-1], map_location=dev, weights_only=True)
        start = ckpt['global_step']
        net.load_state_dict(ckpt['network_fn_state_dict'])
    return render_kwargs_test, start


def raw2outputs(raw, z_vals, rays_d, raw_noise_std=0, white_bkgd=False, pytest=False, nerf_render_rgb=True,
                nerf_render_depth=True):
    """nerf_utils.py:344-395: as run_nerf.raw2outputs, but maps that are switched
    off come back as None (all five when both switches are off)."""
    if not nerf_render_rgb and not nerf_render_depth:
        return None, None, None, None, None
    d = rays_d.reshape(-1, 3)
    if d.dtype != torch.float32 or d.stride(-1) != 1:
        d = d.float().contiguous()
    noise = _r._density_noise(raw[..., 3].shape, raw_noise_std, pytest, raw.device)
    return hip.raw2outputs(raw, z_vals, d, d.stride(0), noise, white_bkgd, want_rgb=nerf_render_rgb,
                           want_depth=nerf_render_depth)


def render_rays(ray_batch, network_fn, network_query_fn, N_samples, retraw=False, lindisp=False, perturb=0.,
                N_importance=0, network_fine=None, white_bkgd=False, raw_noise_std=0., verbose=False, pytest=False,
                nerf_render_rgb=True, nerf_render_depth=True):
    """nerf_utils.py:248-342: coarse pass only; returns rgb_map, disp_map,
    acc_map, depth_map (+ raw).  N_importance / network_fine are accepted and
    ignored, as in the reference."""
    if not nerf_render_rgb and not nerf_render_depth:
        ret = {'rgb_map': None, 'disp_map': None, 'acc_map': None, 'depth_map': None}
        if retraw:
            ret['raw'] = _r.render_rays(ray_batch, network_fn, network_query_fn, N_samples, retraw=True,
                                        lindisp=lindisp, perturb=perturb, pytest=pytest)['raw']
        return ret
    ret = _r.render_rays(ray_batch, network_fn, network_query_fn, N_samples, retraw=retraw, lindisp=lindisp,
                         perturb=perturb, N_importance=0, network_fine=None, white_bkgd=white_bkgd,
                         raw_noise_std=raw_noise_std, verbose=verbose, pytest=pytest, _with_depth=True,
                         _render_rgb=nerf_render_rgb, _render_depth=nerf_render_depth)
    _r.warn_nonfinite(ret, skip=('disp_map',))       # nerf_utils.py:336-340: every key but disp_map, on every render
    return ret


def batchify_rays(rays_flat, chunk=1024 * 32, **kwargs):
    """nerf_utils.py:233-245."""
    all_ret = {}
    for i in range(0, rays_flat.shape[0], chunk):
        ret = render_rays(rays_flat[i:i + chunk], **kwargs)
        for k in ret:
            all_ret.setdefault(k, []).append(ret[k])
    return {k: (None if v[0] is None else (v[0] if len(v) == 1 else torch.cat(v, 0))) for k, v in all_ret.items()}


def render(H, W, K, chunk=1024 * 32, rays=None, c2w=None, ndc=False, near=0., far=1., use_viewdirs=False,
           c2w_staticcam=None, **kwargs):
    """nerf_utils.py:153-219: returns [rgb_map, disp_map, acc_map, depth_map, extras]."""
    rays_flat, sh = _r._build_rays(H, W, K, rays, c2w, ndc, near, far, use_viewdirs, c2w_staticcam)
    all_ret = batchify_rays(rays_flat, chunk, **kwargs)
    for k in all_ret:
        if all_ret[k] is not None:
            all_ret[k] = torch.reshape(all_ret[k], sh + list(all_ret[k].shape[1:]))
    k_extract = ['rgb_map', 'disp_map', 'acc_map', 'depth_map']
    ret_list = [all_ret[k] for k in k_extract]
    ret_dict = {k: all_ret[k] for k in all_ret if k not in k_extract}
    return ret_list + [ret_dict]


def render_path(render_poses, hwf, K, chunk, render_kwargs, savedir=None, render_factor=0):
    """nerf_utils.py:107-151: per pose one render; returns numpy (rgbs, disps,
    raws, depths) stacked over poses."""
    H, W, focal = hwf
    if render_factor != 0:
        H = H // render_factor
        W = W // render_factor
        focal = focal / render_factor
    rgbs, disps, raws, depths = [], [], [], []
    for i, c2w in enumerate(render_poses):
        rgb, disp, acc, depth, extra = render(H, W, K, chunk=chunk, c2w=c2w[:3, :4], **render_kwargs)
        rgbs.append(rgb.cpu().detach().numpy())
        disps.append(disp.cpu().detach().numpy())
        raws.append(extra['raw'].cpu().detach().numpy())
        depths.append(depth.cpu().detach().numpy())
        if savedir is not None:
            _r._save_frame(savedir, i, rgbs[-1], extra['raw'] if render_kwargs.get('retraw') else None)
    return np.stack(rgbs, 0), np.stack(disps, 0), np.stack(raws, 0), np.stack(depths, 0)


def _rays_of_views(H, W, Ks, poses_per_object, ndc, near, far, use_viewdirs):
    """[sum of poses * H * W, 8|11] rays of all poses of all objects (object o: intrinsics Ks[o]), object after object,
    pose after pose.  Device-resident poses and plain near / far go through ONE launch (a pose table instead of a
    get_rays per pose); everything else pose by pose through the general entry."""
    first = poses_per_object[0][0] if len(poses_per_object) and len(poses_per_object[0]) else None
    plain = (not ndc and first is not None and isinstance(first, torch.Tensor) and first.is_cuda
             and not isinstance(near, torch.Tensor) and not isinstance(far, torch.Tensor))
    if not plain:
        return torch.cat([_r._build_rays(H, W, Ks[o], None, c2w[:3, :4], ndc, near, far, use_viewdirs, None)[0]
                          for o in range(len(poses_per_object)) for c2w in poses_per_object[o]], 0)
    dev = first.device
    c2w = torch.cat([(p if isinstance(p, torch.Tensor) else torch.stack(list(p)))[:, :3, :4].reshape(-1, 12)
                     for p in poses_per_object], 0).to(dtype=torch.float32)
    counts = [len(p) for p in poses_per_object]
    kf = torch.tensor([[float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2])] for K in Ks], dtype=torch.float32)
    kf = torch.repeat_interleave(kf, torch.tensor(counts), dim=0).to(dev, non_blocking=True)
    return hip.make_rays_views(H, W, torch.cat([c2w, kf], 1), near, far, use_viewdirs)


def render_path_device(render_poses, hwf, K, render_kwargs, render_factor=0):
    """All poses of render_path in ONE ray batch, results left on the device: (rgbs [P,H,W,3], disps [P,H,W],
    raws [P,H,W,S,C] or None, depths [P,H,W]).  The reference renders pose by pose and copies every result to
    numpy (nerf_utils.py:107-151; model_3detr.py:619-724 then moves them back to the GPU); a detector that
    consumes the maps as tensors calls this instead."""
    H, W, focal = hwf
    if render_factor != 0:
        H, W = H // render_factor, W // render_factor
    kw = dict(render_kwargs)
    near, far = kw.pop('near', 0.), kw.pop('far', 1.)
    use_viewdirs, ndc = kw.pop('use_viewdirs', False), kw.pop('ndc', False)
    rays = _rays_of_views(H, W, [K], [render_poses], ndc, near, far, use_viewdirs)
    ret = batchify_rays(rays, rays.shape[0], **kw)
    P = len(render_poses)
    shp = lambda t: None if t is None else t.reshape([P, H, W] + list(t.shape[1:]))
    return shp(ret['rgb_map']), shp(ret.get('disp_map')), shp(ret.get('raw')), shp(ret.get('depth_map'))


def render_path_device_objects(render_poses, hwf, K, render_kwargs_list, render_factor=0):
    """render_path_device for SEVERAL objects at once: object o is rendered with render_kwargs_list[o] (its own
    network_fn, everything else taken from the first entry) from render_poses[o] (all objects: the same number of
    poses, one image size).  K: one 3x3 matrix for all, or one per object.  All objects' rays form one batch and the
    networks run as one launch (nerf_mlp_forward_objects); the reference renders scan after scan, pose after pose
    (model_3detr.py:619-724).  Returns (rgbs [O,P,H,W,3], disps [O,P,H,W], raws [O,P,H,W,S,C] or None, depths [O,P,H,W])."""
    H, W, focal = hwf
    if render_factor != 0:
        H, W = H // render_factor, W // render_factor
    kw = dict(render_kwargs_list[0])
    near, far = kw.pop('near', 0.), kw.pop('far', 1.)
    use_viewdirs, ndc = kw.pop('use_viewdirs', False), kw.pop('ndc', False)
    O, P = len(render_kwargs_list), len(render_poses[0])
    if len(render_poses) != O or any(len(p) != P for p in render_poses):
        raise ValueError("render_path_device_objects: %d objects need %d pose lists of equal length, got lengths %s"
                         % (O, O, [len(p) for p in render_poses]))
    Ks = K if (isinstance(K, (list, tuple)) and len(K) == O and np.ndim(K[0]) == 2) else [K] * O
    rays = _rays_of_views(H, W, Ks, render_poses, ndc, near, far, use_viewdirs)
    kw['network_fn'] = [k['network_fn'] for k in render_kwargs_list]
    ret = batchify_rays(rays, rays.shape[0], **kw)
    shp = lambda t: None if t is None else t.reshape([O, P, H, W] + list(t.shape[1:]))
    return shp(ret['rgb_map']), shp(ret.get('disp_map')), shp(ret.get('raw')), shp(ret.get('depth_map'))


_DEPTH_TYPES = ('rendered_depth', 'nerf_rendered_depth', 'nerf_rendered_rgb_rendered_depth', 'rendered_rgb_rendered_depth')


def _assemble(input_type, rgbs, depths, raws, depths_for_plain_depth):
    """The per-input_type tensor the detector consumes (model_3detr.py:657-670 / 699-712)."""
    if input_type == 'rendered_rgb':
        return rgbs
    if input_type == 'rendered_depth':
        return depths_for_plain_depth[..., None]
    rgb4 = rgbs[..., None].expand(*rgbs.shape, 4)                      # np.repeat(rgbs[..., None], 4, -1)
    dep4 = depths[..., None, None].expand(*depths.shape, 1, 4)          # np.repeat(disps[..., None, None], 4, -1)
    if input_type == 'nerf_rendered_rgb':
        return torch.cat((raws, rgb4), -2)
    if input_type == 'nerf_rendered_depth':
        return torch.cat((raws, dep4), -2)
    if input_type == 'nerf_rendered_rgb_rendered_depth':
        return torch.cat((raws, rgb4, dep4), -2)
    if input_type == 'rendered_rgb_rendered_depth':
        return torch.cat((rgb4, dep4), -2)
    return raws                                                         # 'nerf': the raw volume itself


def detector_inputs(poses, render_kwargs, input_type='nerf', focal_length=None):
    """What Model3DETR.forward builds from one scan before its backbone runs (model_3detr.py:619-724), kept on the
    device: poses [..., 3, 5] (LLFF layout, last column = H, W, focal) -> dict with
      'raws'         [P, H, W, S(+3|+1|+4), 4] (or the rgb / depth maps, by input_type),
      'coarse_raws'  the same rendered with focal / focal_length (if focal_length is given: the coarse field of view
                     of the coarse-to-fine architectures), 'rgbs', 'depths', 'disps' (+ 'coarse_*').
    NaN depths / disparities (rays that hit nothing) become 1 for the depth input types, as in the reference.  One
    quirk is kept: for input_type 'rendered_depth' the coarse tensor is built from the FINE depths
    (model_3detr.py:702)."""
    poses = poses.reshape(-1, 3, 5)
    H, W, focal = [float(v) for v in poses[0, :3, -1].cpu()]
    H, W = int(H), int(W)
    render_poses = poses[:, :3, :4]
    out = {}

    def one(f, tag):
        K = np.array([[f, 0, 0.5 * W], [0, f, 0.5 * H], [0, 0, 1]])
        rgbs, disps, raws, depths = render_path_device(render_poses, [H, W, focal], K, render_kwargs)
        if input_type in _DEPTH_TYPES:
            depths, disps = torch.nan_to_num(depths, nan=1.), torch.nan_to_num(disps, nan=1.)
        out[tag + 'rgbs'], out[tag + 'disps'], out[tag + 'depths'] = rgbs, disps, depths
        return rgbs, depths, raws

    rgbs, depths, raws = one(focal, '')
    out['raws'] = _assemble(input_type, rgbs, depths, raws, depths)
    if focal_length is not None:
        c_rgbs, c_depths, c_raws = one(focal / focal_length, 'coarse_')
        out['coarse_raws'] = _assemble(input_type, c_rgbs, c_depths, c_raws, depths)
    return out


def detector_inputs_batch(poses, render_kwargs_list, input_type='nerf', focal_length=None):
    """detector_inputs for a batch of scans, each with its own network (what one Model3DETR.forward call renders when
    none of its scans is cached, model_3detr.py:619-724): poses [B, P, 3, 5].  One network launch per field of view
    for the whole batch.  Returns the same keys as detector_inputs with a leading batch dimension."""
    B = len(render_kwargs_list)
    poses = poses.reshape(B, -1, 3, 5)
    hwf = poses[:, 0, :3, -1].cpu()
    H, W = int(hwf[0, 0]), int(hwf[0, 1])
    if not (bool((hwf[:, 0] == H).all()) and bool((hwf[:, 1] == W).all())):
        raise ValueError("the scans of a batch must share the image size")
    focals = [float(f) for f in hwf[:, 2]]
    render_poses = [poses[b, :, :3, :4] for b in range(B)]
    out = {}

    def one(divisor, tag):
        Ks = [np.array([[f / divisor, 0, 0.5 * W], [0, f / divisor, 0.5 * H], [0, 0, 1]]) for f in focals]
        rgbs, disps, raws, depths = render_path_device_objects(render_poses, [H, W, focals[0]], Ks, render_kwargs_list)
        if input_type in _DEPTH_TYPES:
            depths, disps = torch.nan_to_num(depths, nan=1.), torch.nan_to_num(disps, nan=1.)
        out[tag + 'rgbs'], out[tag + 'disps'], out[tag + 'depths'] = rgbs, disps, depths
        return rgbs, depths, raws

    rgbs, depths, raws = one(1.0, '')
    out['raws'] = _assemble(input_type, rgbs, depths, raws, depths)
    if focal_length is not None:
        c_rgbs, c_depths, c_raws = one(focal_length, 'coarse_')       # focal / focal_length, as object by object
        out['coarse_raws'] = _assemble(input_type, c_rgbs, c_depths, c_raws, depths)
    return out
