"""Drop-in for the names of nerf-pytorch/run_nerf_helpers.py that NeRF-Loc uses
(run_nerf_helpers.py:9-239): img2mse, mse2psnr, to8b, Embedder / get_embedder,
NeRF, get_rays, get_rays_np, ndc_rays, sample_pdf.

Same names, arguments and return values; the tensor work is done by
libnerf_hip.so on the GPU that holds the inputs.
"""
import numpy as np
import torch
import torch.nn as nn

from . import hip

# ---------------------------------------------------------------------------- misc (run_nerf_helpers.py:9-11)
img2mse = lambda x, y: torch.mean((x - y) ** 2)
mse2psnr = lambda x: -10. * torch.log(x) / torch.log(torch.tensor([10.], dtype=x.dtype, device=x.device))
to8b = lambda x: (255 * np.clip(x, 0, 1)).astype(np.uint8)

_DEFAULT_PRECISION = ["fp32"]


def set_default_precision(name):
    """'fp32' (exact float32 MFMA) or 'bf16' (bf16 MFMA, float32 accumulate) for
    networks that do not set their own ``precision`` attribute."""
    if name not in hip.PRECISIONS:
        raise ValueError("unknown precision %r" % (name,))
    _DEFAULT_PRECISION[0] = name


def get_default_precision():
    return _DEFAULT_PRECISION[0]


# ---------------------------------------------------------------------------- positional encoding
class Embedder:
    """Same keyword interface as run_nerf_helpers.py:15-45.  Only the
    configuration get_embedder builds (3 inputs, log-spaced power-of-two
    frequencies, [sin, cos]) is implemented in HIP; anything else raises."""

    def __init__(self, **kwargs):
        self.kwargs = kwargs
        d = kwargs["input_dims"]
        n = kwargs["num_freqs"]
        fns = list(kwargs["periodic_fns"])
        if d != 3 or not kwargs["log_sampling"] or kwargs["max_freq_log2"] != n - 1 or fns != [torch.sin, torch.cos]:
            raise NotImplementedError("Embedder: only input_dims=3, log_sampling=True, max_freq_log2=num_freqs-1, "
                                      "periodic_fns=[torch.sin, torch.cos] is implemented")
        self.include_input = bool(kwargs["include_input"])
        self.num_freqs = int(n)
        self.out_dim = (d if self.include_input else 0) + 2 * d * self.num_freqs

    def embed(self, inputs):
        return hip.embed(inputs, self.num_freqs, self.include_input)


class EmbedFn:
    """The callable get_embedder returns.  It is a plain function of one tensor,
    like the reference's lambda, and additionally exposes ``embedder`` so that
    run_network can fuse the encoding into the network kernel."""

    def __init__(self, embedder):
        self.embedder = embedder

    def __call__(self, x):
        return self.embedder.embed(x)


def get_embedder(multires, i=0):
    """run_nerf_helpers.py:48-63: returns (embed_fn, out_dim); i == -1 disables
    the encoding (identity, 3 channels)."""
    if i == -1:
        return nn.Identity(), 3
    embed_kwargs = {
        'include_input': True,
        'input_dims': 3,
        'max_freq_log2': multires - 1,
        'num_freqs': multires,
        'log_sampling': True,
        'periodic_fns': [torch.sin, torch.cos],
    }
    eo = Embedder(**embed_kwargs)
    return EmbedFn(eo), eo.out_dim


# ---------------------------------------------------------------------------- the network
class NeRF(nn.Module):
    """Same constructor, parameter names and state_dict layout as
    run_nerf_helpers.py:67-94, so nerf-pytorch checkpoints load unchanged.  The
    forward pass runs the fused MFMA kernel for the architecture the
    reference's configs use (D=8, W=256, skips=[4], input_ch=63,
    input_ch_views=27 with view directions, output_ch 4 or 5); other shapes
    (--netdepth, --netwidth) are evaluated layer by layer with library GEMMs."""

    def __init__(self, D=8, W=256, input_ch=3, input_ch_views=3, output_ch=4, skips=[4], use_viewdirs=False):
        super(NeRF, self).__init__()
        self.D = D
        self.W = W
        self.input_ch = input_ch
        self.input_ch_views = input_ch_views
        self.skips = skips
        self.use_viewdirs = use_viewdirs
        self.output_ch = output_ch
        self.precision = None          # None: follow set_default_precision()
        self.train_precision = None    # None: follow train.set_train_precision() (float32 unless changed)
        self._pack_cache = {}

        layers = [nn.Linear(input_ch, W)]
        for i in range(D - 1):
            layers.append(nn.Linear(W + input_ch, W) if i in self.skips else nn.Linear(W, W))
        self.pts_linears = nn.ModuleList(layers)
        self.views_linears = nn.ModuleList([nn.Linear(input_ch_views + W, W // 2)])
        if use_viewdirs:
            self.feature_linear = nn.Linear(W, W)
            self.alpha_linear = nn.Linear(W, 1)
            self.rgb_linear = nn.Linear(W // 2, 3)
        else:
            self.output_linear = nn.Linear(W, output_ch)

    # -- what the kernel can run
    def hip_supported(self):
        ok = (self.D == 8 and self.W == 256 and list(self.skips) == [4] and self.input_ch == 63)
        if self.use_viewdirs:
            return ok and self.input_ch_views == 27
        return ok and self.output_ch in (4, 5)

    @property
    def out_channels(self):
        return 4 if self.use_viewdirs else self.output_ch

    def _require_supported(self):
        if not self.hip_supported():
            raise NotImplementedError(
                "NeRF: the HIP kernel implements D=8, W=256, skips=[4], input_ch=63 and input_ch_views=27 "
                "(view directions) or output_ch in (4, 5) (no view directions); got D=%d W=%d skips=%s "
                "input_ch=%d input_ch_views=%d output_ch=%d" % (self.D, self.W, list(self.skips), self.input_ch,
                                                             self.input_ch_views, self.output_ch))

    def precision_code(self):
        return hip.PRECISIONS[self.precision or get_default_precision()]

    def packed_weights(self, precision=None):
        """Weights re-laid-out as the kernel's MFMA fragment stream, cached until
        a parameter changes (in-place updates bump Tensor._version)."""
        self._require_supported()
        prec = self.precision_code() if precision is None else precision
        sd = self.state_dict(keep_vars=True)
        order = []
        for i in range(8):
            order += ["pts_linears.%d.weight" % i, "pts_linears.%d.bias" % i]
        order += ["views_linears.0.weight", "views_linears.0.bias"]
        if self.use_viewdirs:
            order += ["feature_linear.weight", "feature_linear.bias", "alpha_linear.weight", "alpha_linear.bias",
                      "rgb_linear.weight", "rgb_linear.bias"]
        else:
            order += ["output_linear.weight", "output_linear.bias"]
        tensors = [sd[k] for k in order]
        key = (prec, hip.WEIGHT_GENERATION[0]) + tuple((t.data_ptr(), t._version) for t in tensors)
        hit = self._pack_cache.get(prec)
        if hit is not None and hit[0] == key:
            return hit[1]
        if not self.use_viewdirs:
            tensors[16] = tensors[17] = None
        packed = hip.mlp_pack(tensors, self.use_viewdirs, self.out_channels, prec)
        self._pack_cache[prec] = (key, packed)
        return packed

    def weights_changed(self):
        """Drops the packed copies.  Only needed after writes that bypass the version counters the cache is keyed on
        (`p.data.mul_(...)`, raw-pointer writes of foreign code); optimizers, `load_state_dict` and `copy_` are seen."""
        self._pack_cache.clear()        # forward streams of both precisions and the training code's transposed ones

    def _forward_library_gemms(self, x):
        """Any other shape (--netdepth / --netwidth / other skips or encodings, run_nerf.py:440-446): the layers of
        run_nerf_helpers.py:96-119 as rocBLAS GEMMs through torch.nn.functional, on the GPU the input is on.  No
        fused kernel exists for these shapes; autograd works through it as for any torch module."""
        if not x.is_cuda:
            raise ValueError("NeRF.forward: the input is on %s; this package computes on GPUs only" % x.device)
        F = torch.nn.functional
        pts, views = torch.split(x[..., :self.input_ch + self.input_ch_views], [self.input_ch, self.input_ch_views], dim=-1)
        h = pts
        for i, lin in enumerate(self.pts_linears):
            h = F.relu(lin(h))
            if i in self.skips:
                h = torch.cat([pts, h], -1)
        if not self.use_viewdirs:
            return self.output_linear(h)
        alpha = self.alpha_linear(h)
        h = torch.cat([self.feature_linear(h), views], -1)
        for lin in self.views_linears:
            h = F.relu(lin(h))
        return torch.cat([self.rgb_linear(h), alpha], -1)

    def forward(self, x):
        """x: [..., input_ch + input_ch_views] encoded features -> [..., 4|output_ch]
        (run_nerf_helpers.py:96-119)."""
        if not self.hip_supported():
            return self._forward_library_gemms(x)
        ch = self.input_ch + (self.input_ch_views if self.use_viewdirs else 0)
        if x.shape[-1] < ch:
            raise ValueError("NeRF.forward expects %d input channels, got %d" % (ch, x.shape[-1]))
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            from . import train
            return train.mlp_forward_autograd(self, x)
        flat = x.reshape(-1, x.shape[-1])
        if flat.shape[-1] != ch:
            flat = flat[:, :ch]
        prec = self.precision_code()
        out = hip.mlp_forward(self.packed_weights(prec), prec, self.use_viewdirs, self.out_channels,
                              hip.IN_FEATURES, flat.contiguous(), None, flat.shape[0], 1)
        return out.reshape(tuple(x.shape[:-1]) + (self.out_channels,))

    def load_weights_from_keras(self, weights):
        """Imports the weight list of the original TensorFlow release
        (run_nerf_helpers.py:121-148): kernels are stored [in, out] there."""
        assert self.use_viewdirs, "Not implemented if use_viewdirs=False"
        def put(lin, k):
            lin.weight.data = torch.from_numpy(np.transpose(weights[k])).to(lin.weight.device)
            lin.bias.data = torch.from_numpy(np.transpose(weights[k + 1])).to(lin.bias.device)
        for i in range(self.D):
            put(self.pts_linears[i], 2 * i)
        put(self.feature_linear, 2 * self.D)
        put(self.views_linears[0], 2 * self.D + 2)
        put(self.rgb_linear, 2 * self.D + 4)
        put(self.alpha_linear, 2 * self.D + 6)


# ---------------------------------------------------------------------------- rays
def get_rays(H, W, K, c2w):
    """run_nerf_helpers.py:153-162 / nerf_utils.py:222-231: (rays_o, rays_d), each [H, W, 3]."""
    c2w = torch.as_tensor(c2w)
    r = hip.make_rays(H, W, K, c2w=c2w, use_viewdirs=False)
    return r[:, 0:3].reshape(H, W, 3), r[:, 3:6].reshape(H, W, 3)


def get_rays_np(H, W, K, c2w):
    """numpy twin used to build the training ray bank on the host
    (run_nerf_helpers.py:165-172)."""
    i, j = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing='xy')
    dirs = np.stack([(i - K[0][2]) / K[0][0], -(j - K[1][2]) / K[1][1], -np.ones_like(i)], -1)
    rays_d = np.sum(dirs[..., np.newaxis, :] * c2w[:3, :3], -1)
    rays_o = np.broadcast_to(c2w[:3, -1], np.shape(rays_d))
    return rays_o, rays_d


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """run_nerf_helpers.py:175-192 on device tensors of shape [..., 3]."""
    sh = rays_d.shape
    K = [[focal, 0, 0], [0, focal, 0], [0, 0, 1]]
    r = hip.make_rays(H, W, K, rays_o=rays_o, rays_d=rays_d, use_viewdirs=False, ndc=True, ndc_near=float(near))
    return r[:, 0:3].reshape(sh), r[:, 3:6].reshape(sh)


# ---------------------------------------------------------------------------- hierarchical sampling
def _pdf_variates(batch_shape, N_samples, det, pytest, device):
    """The u of run_nerf_helpers.py:203-219, including the numpy-seeded test hook."""
    if pytest:
        np.random.seed(0)
        if det:
            u = np.linspace(0., 1., N_samples)
        else:
            u = np.random.rand(*(list(batch_shape) + [N_samples]))
        return torch.Tensor(u).to(device)
    if det:
        return hip.linspace01(N_samples, device)
    return torch.rand(list(batch_shape) + [N_samples], device=device)


def sample_pdf(bins, weights, N_samples, det=False, pytest=False):
    """run_nerf_helpers.py:196-239.  bins [..., B], weights [..., B-1] -> samples [..., N_samples]."""
    lead = bins.shape[:-1]
    b2 = bins.reshape(-1, bins.shape[-1])
    w2 = weights.reshape(-1, weights.shape[-1])
    u = _pdf_variates((b2.shape[0],), N_samples, det, pytest, bins.device)
    zs, _, _, _ = hip.sample_pdf(b2, w2, u)
    return zs.reshape(tuple(lead) + (N_samples,))
