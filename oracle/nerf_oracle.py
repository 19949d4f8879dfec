"""CPU oracle for the NeRF volume-rendering path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain numpy, the algorithm of the reference
(nerf-pytorch as vendored in Jiankai-Sun/nerf-loc) so that the HIP kernels can
be checked against it on a machine where the reference itself is absent.  It is
imported only by tests/, by __graft_entry__.smoke() and by the cpu_baseline leg
of bench.py.  The product package never imports it.

Parity status: PINNED.  tests/test_oracle_golden.py checks every function here
against tests/golden/*.npz, which were produced by running the reference on CPU
(tests/golden/make_golden.py): deterministic sample_pdf indices and samples
bit-exact, everything else to <= 2e-6.  Two functions have no fixture behind
them and say so in their docstrings: detector_assembly (the reference method
it restates cannot be run without the detector's CUDA extension) and the bf16
variant of mlp_backward (the reference has no such mode).

Arithmetic notes (they matter for bit-exact sample_pdf):
  * torch.sum over a contiguous float32 row on CPU is an 8-lane, 4-accumulator
    vector reduction followed by a scalar tail and a lane-ordered final sum
    (rowsum_torch_order below).
  * torch.cumsum / torch.cumprod on CPU float32 run sequentially with a float64
    accumulator, each prefix rounded to float32 on store.
  * torch.linspace on CPU float32 is NOT i/(n-1); its table is taken from the
    host's torch (linspace01) and pinned by g7_linspace.npz.
"""
import math

import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------
# small numeric helpers
# --------------------------------------------------------------------------
def linspace01(n, dtype=np.float32):
    """torch.linspace(0, 1, n) evaluated by the host's torch on CPU
    (run_nerf.py:363, run_nerf_helpers.py:205)."""
    import torch
    td = torch.float64 if np.dtype(dtype) == np.float64 else torch.float32
    return torch.linspace(0.0, 1.0, steps=int(n), dtype=td, device="cpu").numpy().copy()


def _multi_row_sum(rows, nrows_shape):
    """ATen's cascade accumulation over a sequence of equally shaped float32
    addends (SumKernel.cpp multi_row_sum): four levels, a level is flushed into
    the next every 16 (or more, for very long rows) additions."""
    size = len(rows)
    num_levels = 4
    ceil_log2 = 0 if size <= 1 else int(size - 1).bit_length()
    level_power = max(4, ceil_log2 // num_levels)
    level_step = 1 << level_power
    level_mask = level_step - 1
    acc = [np.zeros(nrows_shape, F32) for _ in range(num_levels)]
    i = 0
    while i + level_step <= size:
        for _ in range(level_step):
            acc[0] = acc[0] + rows[i]
            i += 1
        for j in range(1, num_levels):
            acc[j] = acc[j] + acc[j - 1]
            acc[j - 1] = np.zeros(nrows_shape, F32)
            if (i & (level_mask << (j * level_power))) != 0:
                break
    while i < size:
        acc[0] = acc[0] + rows[i]
        i += 1
    for j in range(1, num_levels):
        acc[0] = acc[0] + acc[j]
    return acc[0]


def _row_sum(elems, shape):
    """ATen row_sum: elements are dealt to 4 interleaved accumulators (each a
    cascade sum), the leftover elements go to accumulator 0, then 0 += 1,2,3."""
    ilp = 4
    size_ilp = len(elems) // ilp
    groups = [np.stack([elems[j * ilp + k] for k in range(ilp)], 0) for j in range(size_ilp)]
    part = _multi_row_sum(groups, (ilp,) + shape)
    p0 = part[0]
    for i in range(size_ilp * ilp, len(elems)):
        p0 = p0 + elems[i]
    for k in range(1, ilp):
        p0 = p0 + part[k]
    return p0


def rowsum_torch_order(x):
    """Sum over the last axis of a float32 [..., S] array in the order ATen's
    CPU reduction uses for a contiguous row (run_nerf_helpers.py:199 calls
    torch.sum(weights, -1)): rows of >= 8 elements are reduced as 8-lane
    vectors by _row_sum, then the scalar tail is added, then the 8 lanes in
    order; shorter rows go through _row_sum element by element."""
    x = np.ascontiguousarray(x, dtype=F32)
    lead, S = x.shape[:-1], x.shape[-1]
    x2 = x.reshape(-1, S)
    n = x2.shape[0]
    V = 8
    if S < V:
        return _row_sum([x2[:, i] for i in range(S)], (n,)).reshape(lead)
    nvec = S // V
    p = _row_sum([x2[:, i * V:(i + 1) * V] for i in range(nvec)], (n, V))
    acc = np.zeros(n, F32)
    for k in range(nvec * V, S):
        acc = acc + x2[:, k]
    for lane in range(V):
        acc = acc + p[:, lane]
    return acc.reshape(lead)


def _cum64(x, op):
    return op(x.astype(np.float64), axis=-1)


# --------------------------------------------------------------------------
# rays                                             (run_nerf_helpers.py:153-192)
# --------------------------------------------------------------------------
def get_rays(H, W, K, c2w, dtype=F32):
    """Pixel-centre-less pinhole rays: dirs = [(i-cx)/fx, -(j-cy)/fy, -1],
    rays_d = dirs . c2w[:3,:3]^T, rays_o = c2w[:3,3].  Returns [H,W,3] x2.
    run_nerf_helpers.py:153-162."""
    K = np.asarray(K, dtype=np.float64)
    c2w = np.asarray(c2w, dtype=dtype)
    i = np.arange(W, dtype=dtype)[None, :].repeat(H, 0)
    j = np.arange(H, dtype=dtype)[:, None].repeat(W, 1)
    dx = ((i - dtype(K[0, 2])) / dtype(K[0, 0])).astype(dtype)
    dy = (-(j - dtype(K[1, 2])) / dtype(K[1, 1])).astype(dtype)
    dz = -np.ones_like(dx)
    dirs = np.stack([dx, dy, dz], -1)
    prod = (dirs[..., None, :] * c2w[:3, :3]).astype(dtype)
    rays_d = ((prod[..., 0] + prod[..., 1]) + prod[..., 2]).astype(dtype)
    rays_o = np.broadcast_to(c2w[:3, 3], rays_d.shape).copy()
    return rays_o, rays_d


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """Forward-facing normalised device coordinates, run_nerf_helpers.py:175-192."""
    dt = rays_o.dtype.type
    t = -(dt(near) + rays_o[..., 2]) / rays_d[..., 2]
    rays_o = rays_o + t[..., None] * rays_d
    sx = dt(-1.0 / (W / (2.0 * focal)))
    sy = dt(-1.0 / (H / (2.0 * focal)))
    o0 = sx * rays_o[..., 0] / rays_o[..., 2]
    o1 = sy * rays_o[..., 1] / rays_o[..., 2]
    o2 = dt(1.0) + dt(2.0 * near) / rays_o[..., 2]
    d0 = sx * (rays_d[..., 0] / rays_d[..., 2] - rays_o[..., 0] / rays_o[..., 2])
    d1 = sy * (rays_d[..., 1] / rays_d[..., 2] - rays_o[..., 1] / rays_o[..., 2])
    d2 = dt(-2.0 * near) / rays_o[..., 2]
    return np.stack([o0, o1, o2], -1), np.stack([d0, d1, d2], -1)


def pack_rays(rays_o, rays_d, near, far, use_viewdirs=True, viewdirs_from=None):
    """[N, 8|11] batch: o, d, near, far, (unit view dir).  run_nerf.py:102-123."""
    dt = rays_d.dtype
    o = rays_o.reshape(-1, 3)
    d = rays_d.reshape(-1, 3)
    cols = [o, d, np.full((d.shape[0], 1), near, dt), np.full((d.shape[0], 1), far, dt)]
    if use_viewdirs:
        v = (viewdirs_from if viewdirs_from is not None else rays_d).reshape(-1, 3)
        nrm = np.sqrt((v * v).sum(-1, keepdims=True, dtype=dt))
        cols.append((v / nrm).astype(dt))
    return np.concatenate(cols, -1).astype(dt)


# --------------------------------------------------------------------------
# positional encoding + MLP                  (run_nerf_helpers.py:15-63, 67-119)
# --------------------------------------------------------------------------
def embed(x, L):
    """[x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)] on the
    last axis ([..., 3] -> [..., 3 + 6L]).  run_nerf_helpers.py:24-45."""
    out = [x]
    for f in range(L):
        xf = x * x.dtype.type(2.0 ** f)
        out.append(np.sin(xf))
        out.append(np.cos(xf))
    return np.concatenate(out, -1).astype(x.dtype)


def _lin(sd, name, x):
    w = sd[name + ".weight"].astype(x.dtype)
    b = sd[name + ".bias"].astype(x.dtype)
    return x @ w.T + b


def mlp_forward(sd, feat, use_viewdirs=True, D=8, skips=(4,), input_ch=63):
    """NeRF.forward on already-encoded features [N, 63(+27)] -> [N, 4|C].
    run_nerf_helpers.py:96-119."""
    pts = feat[:, :input_ch]
    h = pts
    for i in range(D):
        h = np.maximum(_lin(sd, "pts_linears.%d" % i, h), 0)
        if i in skips:
            h = np.concatenate([pts, h], -1)
    if not use_viewdirs:
        return _lin(sd, "output_linear", h)
    alpha = _lin(sd, "alpha_linear", h)
    feature = _lin(sd, "feature_linear", h)
    hv = np.concatenate([feature, feat[:, input_ch:]], -1)
    hv = np.maximum(_lin(sd, "views_linears.0", hv), 0)
    rgb = _lin(sd, "rgb_linear", hv)
    return np.concatenate([rgb, alpha], -1)


def run_network(sd, pts, viewdirs, use_viewdirs=True, chunk=1 << 16, mlp=None):
    """pts [R,S,3] (+ viewdirs [R,3]) -> raw [R,S,C].  run_nerf.py:37-51.  mlp: the network evaluation to use on the
    encoded features, (sd, features, use_viewdirs) -> raw; default mlp_forward."""
    mlp = mlp or mlp_forward
    R, S = pts.shape[:2]
    flat = pts.reshape(-1, 3)
    f = embed(flat, 10)
    if use_viewdirs:
        vd = np.broadcast_to(viewdirs[:, None, :], pts.shape).reshape(-1, 3)
        f = np.concatenate([f, embed(vd, 4)], -1)
    outs = [mlp(sd, f[i:i + chunk], use_viewdirs) for i in range(0, f.shape[0], chunk)]
    out = np.concatenate(outs, 0)
    return out.reshape(R, S, out.shape[-1])


# --------------------------------------------------------------------------
# compositing                                              (run_nerf.py:268-311)
# --------------------------------------------------------------------------
def raw2outputs(raw, z_vals, rays_d, noise=None, white_bkgd=False):
    """Alpha compositing.  noise, if given, is the already-scaled additive
    density noise [R,S].  Returns rgb_map, disp_map, acc_map, weights,
    depth_map.  run_nerf.py:281-309."""
    dt = z_vals.dtype.type
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = np.concatenate([dists, np.full(dists[..., :1].shape, F32(1e10), z_vals.dtype)], -1)
    dnorm = np.sqrt((rays_d * rays_d).sum(-1, dtype=np.float64)).astype(z_vals.dtype)
    dists = dists * dnorm[..., None]
    with np.errstate(over="ignore"):
        rgb = dt(1.0) / (dt(1.0) + np.exp(-raw[..., :3]))
    sig = raw[..., 3] if noise is None else raw[..., 3] + noise.astype(raw.dtype)
    with np.errstate(over="ignore", invalid="ignore"):
        alpha = dt(1.0) - np.exp(-np.maximum(sig, 0) * dists)
    one_m = (dt(1.0) - alpha) + dt(1e-10)
    T = _cum64(np.concatenate([np.ones_like(one_m[..., :1]), one_m], -1), np.cumprod)[..., :-1]
    weights = alpha * T.astype(z_vals.dtype)
    rgb_map = (weights[..., None].astype(np.float64) * rgb).sum(-2).astype(z_vals.dtype)
    depth_map = (weights.astype(np.float64) * z_vals).sum(-1).astype(z_vals.dtype)
    acc_map = weights.sum(-1, dtype=np.float64).astype(z_vals.dtype)
    with np.errstate(divide="ignore", invalid="ignore"):
        q = depth_map / acc_map
        m = np.where(np.isnan(q), q, np.maximum(dt(1e-10), q))   # torch.max keeps NaN
        disp_map = dt(1.0) / m
    if white_bkgd:
        rgb_map = rgb_map + (dt(1.0) - acc_map[..., None])
    return rgb_map, disp_map, acc_map, weights, depth_map


# --------------------------------------------------------------------------
# hierarchical sampling                        (run_nerf_helpers.py:196-239)
# --------------------------------------------------------------------------
def sample_pdf(bins, weights, u, return_inds=False):
    """Inverse-CDF sampling.  bins [R,B], weights [R,B-1], u [R,N] or [N] in
    [0,1].  float32 path is bit-exact with the reference on CPU."""
    if bins.dtype == np.float64:
        w = weights + 1e-5
        pdf = w / w.sum(-1, keepdims=True)
        cdf = np.cumsum(pdf, -1)
    else:
        w = (weights.astype(F32) + F32(1e-5)).astype(F32)
        pdf = (w / rowsum_torch_order(w)[..., None]).astype(F32)
        cdf = _cum64(pdf, np.cumsum).astype(F32)
    cdf = np.concatenate([np.zeros_like(cdf[..., :1]), cdf], -1)
    u = np.broadcast_to(u, cdf.shape[:-1] + (u.shape[-1],)).astype(bins.dtype)
    # searchsorted(right=True): number of cdf entries <= u
    inds = (cdf[..., None, :] <= u[..., :, None]).sum(-1)
    below = np.maximum(inds - 1, 0)
    above = np.minimum(inds, cdf.shape[-1] - 1)
    cdf_b = np.take_along_axis(cdf, below, -1)
    cdf_a = np.take_along_axis(cdf, above, -1)
    bin_b = np.take_along_axis(bins, below, -1)
    bin_a = np.take_along_axis(bins, above, -1)
    dt = bins.dtype.type
    denom = cdf_a - cdf_b
    denom = np.where(denom < dt(1e-5), dt(1.0), denom)
    t = (u - cdf_b) / denom
    samples = bin_b + t * (bin_a - bin_b)
    return (samples, inds) if return_inds else samples


# --------------------------------------------------------------------------
# the whole per-ray pipeline                               (run_nerf.py:314-424)
# --------------------------------------------------------------------------
def z_vals_for(rays, N_samples, lindisp=False, t_rand=None):
    """Stratified depths [R,S] (run_nerf.py:363-385)."""
    dt = rays.dtype.type
    near, far = rays[:, 6:7], rays[:, 7:8]
    t = linspace01(N_samples, rays.dtype)
    if not lindisp:
        z = near * (dt(1.0) - t) + far * t
    else:
        z = dt(1.0) / (dt(1.0) / near * (dt(1.0) - t) + dt(1.0) / far * t)
    if t_rand is not None:
        mids = dt(0.5) * (z[..., 1:] + z[..., :-1])
        upper = np.concatenate([mids, z[..., -1:]], -1)
        lower = np.concatenate([z[..., :1], mids], -1)
        z = lower + (upper - lower) * t_rand.astype(rays.dtype)
    return z.astype(rays.dtype)


def render_rays(rays, sd_coarse, N_samples, N_importance=0, sd_fine=None, use_viewdirs=True,
                white_bkgd=False, lindisp=False, t_rand=None, u=None, noise0=None, noise1=None,
                retraw=False, z_samples_override=None, mlp=None):
    """Coarse (and optionally fine) pass over a [R, 8|11] ray batch.

    t_rand [R,S]: stratified jitter (None = perturb 0); u [R,N_imp] or [N_imp]:
    inverse-CDF variates (None = deterministic linspace); noise0/noise1: scaled
    density noise of the two passes; z_samples_override [R,N_imp]: importance
    samples to use for the fine pass instead of the ones computed here (lets a
    test compare the fine pass on identical sample positions, since positions
    inside empty bins are ill-conditioned); mlp: see run_network.  Keys of the result follow
    run_nerf.py:409-418 plus depth_map (nerf_utils.py:330), weights and z_vals.
    """
    dt = rays.dtype.type
    o, d = rays[:, 0:3], rays[:, 3:6]
    vd = rays[:, -3:] if rays.shape[-1] > 8 else None
    if use_viewdirs and vd is None:
        raise ValueError("use_viewdirs needs an 11-column ray batch")
    z = z_vals_for(rays, N_samples, lindisp, t_rand)
    pts = o[:, None, :] + d[:, None, :] * z[:, :, None]
    raw = run_network(sd_coarse, pts, vd, use_viewdirs, mlp=mlp)
    rgb, disp, acc, w, depth = raw2outputs(raw, z, d, noise0, white_bkgd)
    ret = {}
    if N_importance > 0:
        ret.update(rgb0=rgb, disp0=disp, acc0=acc, depth0=depth, weights0=w, z_vals0=z)
        if retraw:
            ret["raw0"] = raw
        mids = dt(0.5) * (z[..., 1:] + z[..., :-1])
        if u is None:
            u = linspace01(N_importance, rays.dtype)
        zs, inds = sample_pdf(mids, w[..., 1:-1], u, return_inds=True)
        ret.update(z_samples_own=zs)
        if z_samples_override is not None:
            zs = z_samples_override.astype(rays.dtype)
        z = np.sort(np.concatenate([z, zs], -1), -1)
        pts = o[:, None, :] + d[:, None, :] * z[:, :, None]
        raw = run_network(sd_fine if sd_fine is not None else sd_coarse, pts, vd, use_viewdirs, mlp=mlp)
        rgb, disp, acc, w, depth = raw2outputs(raw, z, d, noise1, white_bkgd)
        ret.update(z_samples=zs, inds=inds,
                   z_std=np.sqrt(((zs.astype(np.float64) - zs.astype(np.float64).mean(-1, keepdims=True)) ** 2)
                                 .mean(-1)).astype(rays.dtype))
    ret.update(rgb_map=rgb, disp_map=disp, acc_map=acc, depth_map=depth, weights=w, z_vals=z)
    if retraw:
        ret["raw"] = raw
    return ret


# --------------------------------------------------------------------------
# backward (the reference differentiates run_nerf.py:268-311 and
# run_nerf_helpers.py:96-119 with autograd; run_nerf.py:776-785 is the loss)
# --------------------------------------------------------------------------
def raw2outputs_backward(raw, z_vals, rays_d, noise=None, white_bkgd=False, g_rgb=None, g_disp=None, g_acc=None,
                         g_depth=None, g_weights=None):
    """dL/draw [R,S,C] for upstream gradients of the maps, in float64 closed form (what autograd yields for
    raw2outputs): c = sigmoid(raw_rgb), e = exp(-relu(sigma) dist), T = exclusive cumprod(e + 1e-10), w = (1-e) T."""
    R, S, C = raw.shape
    f = np.float64
    z = z_vals.astype(f)
    dists = np.concatenate([z[:, 1:] - z[:, :-1], np.full((R, 1), 1e10)], -1) * np.linalg.norm(rays_d.astype(f), axis=-1)[:, None]
    with np.errstate(over="ignore"):
        c = 1.0 / (1.0 + np.exp(-raw[..., :3].astype(f)))
    pre = raw[..., 3].astype(f) + (0.0 if noise is None else noise.astype(f))
    sig = np.maximum(pre, 0)
    e = np.exp(-sig * dists)
    alpha = 1.0 - e
    one_m = (1.0 - alpha) + 1e-10
    T = np.cumprod(np.concatenate([np.ones((R, 1)), one_m], -1), -1)[:, :-1]
    w = alpha * T
    acc, depth = w.sum(-1), (w * z).sum(-1)
    zero = np.zeros(R)
    g_rgb = np.zeros((R, 3)) if g_rgb is None else g_rgb.astype(f)
    ga = zero if g_acc is None else g_acc.astype(f)
    gd = zero if g_depth is None else g_depth.astype(f)
    if g_disp is not None:
        with np.errstate(divide="ignore", invalid="ignore"):
            q = depth / acc
            disp = 1.0 / q
            live = ~(q <= 1e-10) & (g_disp != 0)
            gd = gd + np.where(live, g_disp * (-disp * disp / acc), 0.0)
            ga = ga + np.where(live, g_disp * (disp * disp * depth / (acc * acc)), 0.0)
    qw = (c * g_rgb[:, None, :]).sum(-1) - (g_rgb.sum(-1)[:, None] if white_bkgd else 0.0) + ga[:, None] + gd[:, None] * z
    if g_weights is not None:              # a loss on the weights themselves (run_nerf.py:303 returns them)
        qw = qw + g_weights.astype(f)
    term = qw * w
    suffix = np.cumsum(term[:, ::-1], -1)[:, ::-1] - term          # sum over later samples
    dalpha = qw * T - suffix / one_m
    dsig = np.where(pre > 0, dalpha * dists * e, 0.0)
    out = np.zeros((R, S, C))
    out[..., :3] = w[..., None] * g_rgb[:, None, :] * c * (1.0 - c)
    out[..., 3] = dsig
    return out


def bf16_round(x):
    """Round float32 values to the nearest bfloat16 (ties to even), returned as float64."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + np.uint32(0x7FFF)
    return ((u + r) & np.uint32(0xFFFF0000)).view(np.float32).astype(np.float64)


def bf16_split(x, terms):
    """x as a sum of `terms` bfloat16 numbers, leading term first: t0 = bf16(x), t1 = bf16(x - t0), ...  (every
    difference is exact in float32).  Returned as float64 arrays; one term is plain rounding."""
    x = np.asarray(x, dtype=np.float32)
    out, rest = [], x
    for _ in range(terms):
        t = bf16_round(rest)
        out.append(t)
        rest = (rest.astype(np.float64) - t).astype(np.float32)
    return out


def _split_product(x_terms, w_terms):
    """sum over the term pairs (i, j) with i + j < number of terms of x_i @ w_j^T: one product for plain bf16, three
    for two terms, six for three.  The dropped pairs are below 2^-8 terms of the leading product."""
    n = len(x_terms)
    acc = 0.0
    for i in range(n):
        for j in range(n - i):
            acc = acc + x_terms[i] @ w_terms[j].T
    return acc


def mlp_forward_bf16(sd, feat, use_viewdirs=True, terms=1, layer_terms=None):
    """NeRF.forward (run_nerf_helpers.py:96-119) as the bf16 matrix-core kernels evaluate it: every matrix-product
    operand (weight, encoded input, stored activation) is a sum of `terms` bfloat16 numbers, products are exact and
    summed in higher precision, the bias joins the sum unrounded, and each layer's output is rounded to float32
    before ReLU and the split into operands of the next layer.  terms=1 is the plain bf16 mode, terms=3 with its six
    leading products per element the float32-accurate mode.  layer_terms: {layer name: terms} overrides per layer
    (both operands of that layer's product).  The reference has no such mode: no fixture pins this function; it is the
    yardstick that separates the format's error from a kernel's."""
    lt = layer_terms or {}
    f32 = lambda v: np.asarray(v, np.float64).astype(np.float32)

    def lin(name, x32):
        n = lt.get(name, terms)
        y = _split_product(bf16_split(x32, n), bf16_split(sd[name + ".weight"], n)) + sd[name + ".bias"].astype(np.float64)
        return f32(y)

    feat = np.asarray(feat, np.float32)
    pts, views = feat[:, :63], feat[:, 63:]
    h = pts
    for i in range(8):
        h = np.maximum(lin("pts_linears.%d" % i, h), 0)
        if i == 4:
            h = np.concatenate([pts, h], -1)
    if not use_viewdirs:
        return lin("output_linear", h)
    alpha = lin("alpha_linear", h)
    feature = lin("feature_linear", h)
    hv = np.maximum(lin("views_linears.0", np.concatenate([feature, views], -1)), 0)
    return np.concatenate([lin("rgb_linear", hv), alpha], -1)


def mlp_backward(sd, feat, d_raw, use_viewdirs=True, bf16=False):
    """Gradients of the parameter tensors for dL/draw = d_raw [N, C], float64 (C = 4 with view directions, else the
    rows of output_linear).  Returns {state_dict name: gradient}; the view branch is pinned by the reference's
    autograd gradients (g8), the branch without view directions shares the trunk and has no reference fixture.

    bf16=True restates the mixed-precision training kernels instead: every matrix-product operand (weights, layer
    inputs, stored activations, gradients) is rounded to bfloat16, sums are exact, ReLU masks come from the rounded
    activations.  No reference fixture exists for it (the reference trains in float32)."""
    f = np.float64
    q = bf16_round if bf16 else (lambda v: np.asarray(v, f))
    W = {k: (q(v) if k.endswith("weight") else v.astype(f)) for k, v in sd.items()}
    x = q(feat)
    pts, views = x[:, :63], x[:, 63:90]
    ins, act = [], []
    h = pts
    for i in range(8):
        ins.append(h)
        a = h @ W["pts_linears.%d.weight" % i].T + W["pts_linears.%d.bias" % i]
        hr = q(np.maximum(a, 0))
        act.append(hr)
        h = hr
        if i == 4:
            h = np.concatenate([pts, h], -1)
    h8 = h
    g = {}
    d = q(d_raw)
    if use_viewdirs:
        feature = q(h8 @ W["feature_linear.weight"].T + W["feature_linear.bias"])
        hv_in = np.concatenate([feature, views], -1)
        hv = q(np.maximum(hv_in @ W["views_linears.0.weight"].T + W["views_linears.0.bias"], 0))
        d_rgb, d_sig = d[:, :3], d[:, 3:4]
        g["rgb_linear.weight"], g["rgb_linear.bias"] = d_rgb.T @ hv, d_rgb.sum(0)
        d_hv = q((d_rgb @ W["rgb_linear.weight"]) * (hv > 0))
        g["views_linears.0.weight"], g["views_linears.0.bias"] = d_hv.T @ hv_in, d_hv.sum(0)
        d_feat = q((d_hv @ W["views_linears.0.weight"])[:, :256])
        g["feature_linear.weight"], g["feature_linear.bias"] = d_feat.T @ h8, d_feat.sum(0)
        g["alpha_linear.weight"], g["alpha_linear.bias"] = d_sig.T @ h8, d_sig.sum(0)
        dh = d_feat @ W["feature_linear.weight"] + d_sig @ W["alpha_linear.weight"]
    else:
        g["output_linear.weight"], g["output_linear.bias"] = d.T @ h8, d.sum(0)
        dh = d @ W["output_linear.weight"]
    for i in range(7, -1, -1):
        dz = q(dh * (act[i] > 0))
        g["pts_linears.%d.weight" % i], g["pts_linears.%d.bias" % i] = dz.T @ ins[i], dz.sum(0)
        dh = dz @ W["pts_linears.%d.weight" % i]
        if i == 5:
            dh = dh[:, 63:]
    return g


def train_step_grads(rays, target, sd_coarse, sd_fine, N_samples=64, N_importance=128, white_bkgd=True, t_rand=None,
                     u=None, noise0=None, noise1=None, use_viewdirs=True):
    """Loss img2mse(rgb_map, target) + img2mse(rgb0, target) (run_nerf.py:776-785) and its gradients with respect
    to both networks.  Returns (loss, grads_coarse, grads_fine).  use_viewdirs=False: the architecture with
    output_linear (4 or 5 rows) on 8-column rays."""
    r = render_rays(rays, sd_coarse, N_samples, N_importance, sd_fine=sd_fine, white_bkgd=white_bkgd, t_rand=t_rand,
                    u=u, noise0=noise0, noise1=noise1, retraw=True, use_viewdirs=use_viewdirs)
    R = rays.shape[0]
    d, vd = rays[:, 3:6], (rays[:, 8:11] if use_viewdirs else None)
    loss = img2mse(r["rgb_map"].astype(np.float64), target) + img2mse(r["rgb0"].astype(np.float64), target)
    out = []
    for raw, z, rgb, noise, sd in ((r["raw0"], r["z_vals0"], r["rgb0"], noise0, sd_coarse),
                                   (r["raw"], r["z_vals"], r["rgb_map"], noise1, sd_fine)):
        g_rgb = 2.0 * (rgb.astype(np.float64) - target) / (3.0 * R)
        d_raw = raw2outputs_backward(raw, z, d, noise, white_bkgd, g_rgb=g_rgb)
        S = z.shape[1]
        pts = (rays[:, None, 0:3] + d[:, None, :] * z[:, :, None]).reshape(-1, 3)
        if use_viewdirs:
            feat = np.concatenate([embed(pts, 10), embed(np.repeat(vd, S, 0), 4)], -1)
        else:
            feat = embed(pts, 10)
        out.append(mlp_backward(sd, feat, d_raw.reshape(-1, raw.shape[-1]), use_viewdirs=use_viewdirs))
    return loss, out[0], out[1]


def render_image(H, W, K, c2w, sd_coarse, near, far, use_viewdirs=True, ndc=False, c2w_staticcam=None, **kw):
    """render(H, W, K, c2w=...) of run_nerf.py:69-134: full image, maps shaped [H,W,...].  With view directions a
    static camera supplies the ray geometry while the directions fed to the network stay those of c2w (:102-107);
    the NDC warp (:112-114, near plane 1) acts on the geometry only."""
    o, d = get_rays(H, W, K, c2w)
    view = d
    if use_viewdirs and c2w_staticcam is not None:
        o, d = get_rays(H, W, K, c2w_staticcam)
    if ndc:
        o, d = ndc_rays(H, W, K[0][0], 1.0, o.reshape(-1, 3), d.reshape(-1, 3))
    rays = pack_rays(o, d, near, far, use_viewdirs, viewdirs_from=view)
    r = render_rays(rays, sd_coarse, use_viewdirs=use_viewdirs, **kw)
    return {k: v.reshape((H, W) + v.shape[1:]) for k, v in r.items()}


def detector_assembly(input_type, rgbs, raws, depths, fine_depths=None):
    """The tensor Model3DETR.forward hands to its backbone for one field of view, from render_path's numpy outputs
    rgbs [P,H,W,3], raws [P,H,W,S,4], depths [P,H,W] (models/model_3detr.py:657-670 for the fine field of view,
    :699-712 for the coarse one).  For the depth input types NaN depths are set to 1 first (:644-647, :686-690).
    fine_depths: the reference builds the coarse 'rendered_depth' tensor from the FINE depths (:702); pass them to
    reproduce that.  Restated from the source: the method cannot be run here (the detector needs its CUDA point-cloud
    extension), so this part of the checker is not pinned by fixtures."""
    if input_type in ("rendered_depth", "nerf_rendered_depth", "nerf_rendered_rgb_rendered_depth",
                      "rendered_rgb_rendered_depth"):
        depths = np.nan_to_num(depths, nan=1.0)
        if fine_depths is not None:
            fine_depths = np.nan_to_num(fine_depths, nan=1.0)
    rgb4 = np.repeat(rgbs[..., None], 4, axis=-1)
    dep4 = np.repeat(depths[..., None, None], 4, axis=-1)
    if input_type == "rendered_rgb":
        return rgbs
    if input_type == "rendered_depth":
        return (depths if fine_depths is None else fine_depths)[..., None]
    if input_type == "nerf_rendered_rgb":
        return np.concatenate((raws, rgb4), axis=-2)
    if input_type == "nerf_rendered_depth":
        return np.concatenate((raws, dep4), axis=-2)
    if input_type == "nerf_rendered_rgb_rendered_depth":
        return np.concatenate((raws, rgb4, dep4), axis=-2)
    if input_type == "rendered_rgb_rendered_depth":
        return np.concatenate((rgb4, dep4), axis=-2)
    return raws


def img2mse(x, y):
    return np.mean((x - y) ** 2, dtype=np.float64)


def mse2psnr(x):
    return -10.0 * math.log(x) / math.log(10.0)
