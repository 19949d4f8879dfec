"""ctypes binding of libnerf_hip.so (C ABI: include/nerf_hip.h).

PyTorch is used only for device memory and streams: every wrapper takes torch
tensors, checks device / dtype / contiguity, and passes raw device pointers and
the current HIP stream to the library.  There is no CPU fallback: if the
library is missing or a tensor is not on a GPU the call raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# NERF_HIP_LIB selects another build of the same library (kernel A/B experiments)
LIB_PATH = os.environ.get("NERF_HIP_LIB") or os.path.join(_HERE, "lib", "libnerf_hip.so")

PREC_F32, PREC_BF16, PREC_BF16X3, PREC_BF16X2 = 0, 1, 2, 3
ABI_VERSION = 7      # 2: training entry points take a precision; 3: nerf_mlp_backward_static; 4: nerf_mlp_forward_objects;
                     # 5: nerf_make_rays_views; 6: the bf16x3 / bf16x2 precision codes
IN_FEATURES, IN_POINTS, IN_RAYS = 0, 1, 2
PRECISIONS = {"fp32": PREC_F32, "f32": PREC_F32, "float32": PREC_F32, "bf16": PREC_BF16, "bfloat16": PREC_BF16,
              # float32-accurate on the bf16 matrix cores: operands as sums of three (two) bf16 terms
              "bf16x3": PREC_BF16X3, "bf16x2": PREC_BF16X2}

_c_float_p = ctypes.c_void_p   # device pointers travel as integers

class RenderCfg(ctypes.Structure):
    """nerf_render_cfg of include/nerf_hip.h"""
    _fields_ = [(n, ctypes.c_int) for n in ("n_samples", "n_importance", "precision", "use_viewdirs", "output_ch", "ray_cols",
                                            "lindisp", "white_bkgd", "u_per_ray")] + \
               [(n, ctypes.c_void_p) for n in ("t_vals", "t_rand", "u", "noise0", "noise1")]


class RenderBuffers(ctypes.Structure):
    """nerf_render_buffers of include/nerf_hip.h"""
    _fields_ = [(n, ctypes.c_void_p) for n in ("rgb", "disp", "acc", "depth", "rgb0", "disp0", "acc0", "z_std", "raw", "z_vals",
                                               "z_coarse", "raw_coarse", "weights")]


_SIGNATURES = {
    "nerf_render_rays": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, _c_float_p, ctypes.c_int, ctypes.POINTER(RenderCfg),
                                        ctypes.POINTER(RenderBuffers), ctypes.c_void_p]),
    "nerf_hip_abi_version": (ctypes.c_int, []),
    "nerf_last_error": (ctypes.c_char_p, []),
    "nerf_selftest_mfma": (ctypes.c_int, []),
    "nerf_make_rays_views": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_float_p, ctypes.c_float, ctypes.c_float,
                                            ctypes.c_int, _c_float_p, ctypes.c_void_p]),
    "nerf_make_rays": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_float_p, _c_float_p, _c_float_p,
                                      _c_float_p, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                      ctypes.c_float, ctypes.c_float, _c_float_p, _c_float_p, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_float, ctypes.c_float, _c_float_p, ctypes.c_void_p]),
    "nerf_z_vals": (ctypes.c_int, [_c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_float_p, _c_float_p,
                                   ctypes.c_int, _c_float_p, ctypes.c_void_p]),
    "nerf_embed": (ctypes.c_int, [_c_float_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, _c_float_p,
                                  ctypes.c_void_p]),
    "nerf_count_nonfinite": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_longlong),
                                            ctypes.c_void_p, ctypes.c_void_p]),
    "nerf_mlp_packed_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "nerf_mlp_pack": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_void_p, ctypes.c_void_p]),
    "nerf_mlp_forward": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        _c_float_p, ctypes.c_void_p]),
    "nerf_mlp_forward_objects": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_int, _c_float_p, _c_float_p, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_int, _c_float_p, ctypes.c_void_p]),
    "nerf_raw2outputs": (ctypes.c_int, [_c_float_p, ctypes.c_int, _c_float_p, _c_float_p, ctypes.c_int, _c_float_p,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_float_p, _c_float_p,
                                        _c_float_p, _c_float_p, _c_float_p, ctypes.c_void_p]),
    "nerf_sample_pdf": (ctypes.c_int, [_c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_float_p,
                                       ctypes.c_int, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       _c_float_p, ctypes.c_int, _c_float_p, ctypes.c_void_p, _c_float_p,
                                       _c_float_p, ctypes.c_void_p]),
    "nerf_raw2outputs_backward": (ctypes.c_int, [_c_float_p, ctypes.c_int, _c_float_p, _c_float_p, ctypes.c_int,
                                                 _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_float_p,
                                                 _c_float_p, _c_float_p, _c_float_p, _c_float_p, _c_float_p, ctypes.c_void_p]),
    "nerf_train_tape_bytes": (ctypes.c_size_t, [ctypes.c_longlong, ctypes.c_int, ctypes.c_int]),
    "nerf_train_grad_tape_bytes": (ctypes.c_size_t, [ctypes.c_longlong, ctypes.c_int, ctypes.c_int]),
    "nerf_train_workspace_bytes": (ctypes.c_size_t, []),
    "nerf_mlp_forward_train": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                              _c_float_p, _c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                              _c_float_p, ctypes.c_void_p, ctypes.c_void_p]),
    "nerf_mlp_backward_packed_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "nerf_mlp_pack_backward": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_void_p, ctypes.c_void_p]),
    "nerf_mlp_backward": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         _c_float_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p, _c_float_p,
                                         ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p]),
    "nerf_mlp_backward_static": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                _c_float_p, ctypes.c_longlong, ctypes.c_void_p,
                                                ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p]),
    "nerf_adam_step": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                      ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                      ctypes.POINTER(ctypes.c_longlong), ctypes.c_longlong, ctypes.c_float,
                                      ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                      ctypes.c_void_p]),
}

_lib = None

# Incremented whenever this library writes parameter tensors through raw pointers (adam_step): torch's own version
# counters do not see those writes, and the packed-weight caches are keyed on this as well.
WEIGHT_GENERATION = [0]


class NerfHipError(RuntimeError):
    pass


def exported_symbols():
    """Names include/nerf_hip.h declares; used by the CPU-side load test."""
    return sorted(_SIGNATURES)


def lib():
    """Load the shared library once.  Raises if it has not been built
    (python __graft_entry__.py build, or make -C nerf-loc_amd/csrc)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NerfHipError("%s is missing: build it with `make -C nerf-loc_amd/csrc` "
                               "(there is no CPU fallback)" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        if L.nerf_hip_abi_version() != ABI_VERSION:
            raise NerfHipError("%s has ABI version %d, this binding needs %d: rebuild it with `make -C nerf-loc_amd/csrc`"
                               % (LIB_PATH, L.nerf_hip_abi_version(), ABI_VERSION))
        _lib = L
    return _lib


def _check(code, what):
    if code != 0:
        msg = lib().nerf_last_error().decode("utf-8", "replace")
        raise NerfHipError("%s failed (%d): %s" % (what, code, msg))


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _dir_ptr(d, dev, R, d_stride):
    """Pointer to the first direction component of ray 0: `d` may be a strided view into the ray batch (rays[:, 3:6]),
    so it is checked here instead of being made contiguous."""
    if not isinstance(d, torch.Tensor) or not d.is_cuda or d.device != dev:
        raise ValueError("rays_d must live on %s (got %s); this engine has no CPU path"
                         % (dev, d.device if isinstance(d, torch.Tensor) else type(d).__name__))
    if d.dtype != torch.float32 or d.dim() != 2 or d.shape[0] != R or d.shape[1] < 3 or (R > 1 and d.stride(0) != d_stride) \
            or d.stride(1) != 1:
        raise ValueError("rays_d must be a float32 [%d, 3] view with unit inner stride and row stride %d, got %s %s strides %s"
                         % (R, d_stride, d.dtype, tuple(d.shape), tuple(d.stride())))
    return ctypes.c_void_p(d.data_ptr())


def _need(cond, msg, *args):
    """Operand sizes are checked here, before a kernel is launched on them: an out-of-bounds access on the GPU is not
    an exception but a fault that can take the device down."""
    if not cond:
        raise ValueError(msg % args)


def _mlp_operands_ok(who, mode, use_viewdirs, in0, in1, R, S, ray_cols):
    R, S = int(R), int(S)
    _need(R >= 0 and S >= 1, "%s: bad shape R=%d S=%d", who, R, S)
    if mode == IN_RAYS:
        _need(in1 is not None and in0.numel() >= R * int(ray_cols) and in1.numel() >= R * S,
              "%s: rays need %d x %d and z_vals %d x %d elements", who, R, int(ray_cols), R, S)
    elif mode == IN_POINTS:
        _need(in0.numel() >= R * S * 3 and (in1 is None or in1.numel() >= R * 3),
              "%s: points need %d x 3 and view directions %d x 3 elements", who, R * S, R)
    else:
        ch = 90 if use_viewdirs else 63
        _need(in0.numel() >= R * S * ch, "%s: features need %d x %d elements", who, R * S, ch)


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _f32(t, name, device=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor, got %s" % (name, type(t).__name__))
    if not t.is_cuda:
        raise ValueError("%s must live on a GPU (got %s); this engine has no CPU path" % (name, t.device))
    if device is not None and t.device != device:
        raise ValueError("%s is on %s but %s was expected" % (name, t.device, device))
    if t.dtype != torch.float32:
        raise ValueError("%s must be float32, got %s" % (name, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


def selftest():
    _check(lib().nerf_selftest_mfma(), "nerf_selftest_mfma")


# --------------------------------------------------------------------------- rays
def make_rays(H, W, K, c2w=None, rays_o=None, rays_d=None, near=0.0, far=1.0, use_viewdirs=False, ndc=False,
              c2w_view=None, device=None, ndc_near=1.0):
    """[N, 8|11] ray batch from a pose (pixel grid) or from given origins/directions.  ndc_near is the near plane
    of the NDC warp (ndc_rays' `near`; render() always passes 1.0, run_nerf.py:114)."""
    fx, fy, cx, cy = float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2])
    near_arr = far_arr = None
    if isinstance(near, torch.Tensor):
        near_arr, near = _f32(near.reshape(-1), "near"), 0.0
    if isinstance(far, torch.Tensor):
        far_arr, far = _f32(far.reshape(-1), "far"), 0.0
    if rays_o is None:
        c2w = _f32(torch.as_tensor(c2w)[:3, :4], "c2w")
        dev = c2w.device
        n = int(H) * int(W)
        cv = _f32(torch.as_tensor(c2w_view)[:3, :4], "c2w_staticcam", dev) if c2w_view is not None else None
        o_in = d_in = None
    else:
        o_in, d_in = _f32(rays_o.reshape(-1, 3), "rays_o"), _f32(rays_d.reshape(-1, 3), "rays_d")
        dev = d_in.device
        n = d_in.shape[0]
        _need(o_in.shape[0] == n and o_in.device == dev, "make_rays: %d origins on %s for %d directions on %s",
              o_in.shape[0], o_in.device, n, dev)
        cv = None
    for name, arr in (("near", near_arr), ("far", far_arr)):
        _need(arr is None or (arr.numel() == n and arr.device == dev), "make_rays: per-ray %s must hold %d values on %s",
              name, n, dev)
    out = torch.empty((n, 11 if use_viewdirs else 8), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _check(lib().nerf_make_rays(n, int(H), int(W), _ptr(o_in), _ptr(d_in), _ptr(c2w) if rays_o is None else None,
                                    _ptr(cv), fx, fy, cx, cy, float(near), float(far), _ptr(near_arr), _ptr(far_arr),
                                    int(bool(use_viewdirs)), int(bool(ndc)), fx, float(ndc_near), _ptr(out), _stream(out)),
               "nerf_make_rays")
    return out


def make_rays_views(H, W, views, near, far, use_viewdirs):
    """The pixel grids of several poses in one launch.  views [V, 16] on the device: c2w (3x4 row-major), fx, fy, cx,
    cy.  -> [V*H*W, 8|11]."""
    views = _f32(views, "views")
    if views.dim() != 2 or views.shape[1] != 16:
        raise ValueError("make_rays_views: views must be [V, 16], got %s" % (tuple(views.shape),))
    V = views.shape[0]
    out = torch.empty((V * int(H) * int(W), 11 if use_viewdirs else 8), dtype=torch.float32, device=views.device)
    with torch.cuda.device(views.device):
        _check(lib().nerf_make_rays_views(V, int(H), int(W), _ptr(views), float(near), float(far), int(bool(use_viewdirs)),
                                          _ptr(out), _stream(out)), "nerf_make_rays_views")
    return out


_LINSPACE_CACHE = {}


def linspace01(n, device):
    """torch.linspace(0, 1, n) evaluated on the HOST (its float32 rounding is
    part of the reference's results) and cached per device."""
    key = (int(n), str(device))
    t = _LINSPACE_CACHE.get(key)
    if t is None:
        t = torch.linspace(0.0, 1.0, steps=int(n), dtype=torch.float32, device="cpu").to(device)
        _LINSPACE_CACHE[key] = t
    return t


def z_vals(rays, n_samples, lindisp=False, t_rand=None, t_vals=None):
    rays = _f32(rays, "rays")
    R, cols = rays.shape
    tv = _f32(t_vals, "t_vals", rays.device) if t_vals is not None else linspace01(n_samples, rays.device)
    tr = _f32(t_rand, "t_rand", rays.device) if t_rand is not None else None
    _need(rays.dim() == 2 and cols >= 8 and tv.numel() >= int(n_samples) >= 1,
          "z_vals: rays must be [R, >= 8] and t_vals hold %d values", int(n_samples))
    if tr is not None and tuple(tr.shape) != (R, n_samples):
        raise ValueError("t_rand must be [%d, %d], got %s" % (R, n_samples, tuple(tr.shape)))
    out = torch.empty((R, n_samples), dtype=torch.float32, device=rays.device)
    with torch.cuda.device(rays.device):
        _check(lib().nerf_z_vals(_ptr(rays), cols, R, int(n_samples), _ptr(tv), _ptr(tr), int(bool(lindisp)),
                                 _ptr(out), _stream(out)), "nerf_z_vals")
    return out


def embed(x, multires, include_input=True):
    x = _f32(x, "inputs")
    if x.shape[-1] != 3:
        raise ValueError("positional encoding expects [..., 3] inputs, got %s" % (tuple(x.shape),))
    flat = x.reshape(-1, 3)
    out = torch.empty((flat.shape[0], (3 if include_input else 0) + 6 * multires), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib().nerf_embed(_ptr(flat), flat.shape[0], int(multires), int(bool(include_input)), _ptr(out),
                                _stream(out)), "nerf_embed")
    return out.reshape(tuple(x.shape[:-1]) + (out.shape[-1],))


def count_nonfinite(tensors):
    """Number of NaN / Inf values in each of the given float32 device tensors: one launch per 16 tensors and no host
    synchronisation (the result is an int32 device tensor)."""
    ts = [_f32(t, "tensor %d" % i) for i, t in enumerate(tensors)]
    dev = ts[0].device
    out = torch.empty((len(ts),), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        for i in range(0, len(ts), 16):
            part = ts[i:i + 16]
            k = len(part)
            ptrs = (ctypes.c_void_p * k)(*[t.data_ptr() for t in part])
            numel = (ctypes.c_longlong * k)(*[t.numel() for t in part])
            _check(lib().nerf_count_nonfinite(k, ptrs, numel, ctypes.c_void_p(out[i:].data_ptr()), _stream(out)),
                   "nerf_count_nonfinite")
    return out


# --------------------------------------------------------------------------- network
def mlp_pack(tensors, use_viewdirs, output_ch, precision):
    """tensors: list of device tensors in state_dict order (None allowed for the
    unused views_linears of a network without view directions)."""
    dev = next(t for t in tensors if t is not None).device
    keep = [None if t is None else _f32(t.detach(), "weight %d" % i, dev) for i, t in enumerate(tensors)]
    arr = (ctypes.c_void_p * len(keep))(*[None if t is None else t.data_ptr() for t in keep])
    nbytes = lib().nerf_mlp_packed_bytes(precision, int(bool(use_viewdirs)))
    packed = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _check(lib().nerf_mlp_pack(arr, int(bool(use_viewdirs)), int(output_ch), precision,
                                   ctypes.c_void_p(packed.data_ptr()), _stream(packed)), "nerf_mlp_pack")
    return packed


# Set to a list to have every network launch bracketed by HIP events on its own stream:
# entries are (start_event, end_event, n_samples).  Used by bench.py for the roofline figure.
MLP_EVENTS = None


def mlp_forward(packed, precision, use_viewdirs, output_ch, mode, in0, in1, R, S, ray_cols=0):
    in0 = _f32(in0, "mlp input")
    in1 = _f32(in1, "mlp second input", in0.device) if in1 is not None else None
    _mlp_operands_ok("mlp_forward", mode, use_viewdirs, in0, in1, R, S, ray_cols)
    out = torch.empty((R * S, output_ch), dtype=torch.float32, device=in0.device)
    ev = MLP_EVENTS
    with torch.cuda.device(in0.device):
        if ev is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(in0.device))
        _check(lib().nerf_mlp_forward(ctypes.c_void_p(packed.data_ptr()), precision, int(bool(use_viewdirs)),
                                      int(output_ch), mode, _ptr(in0), _ptr(in1), int(R), int(S), int(ray_cols),
                                      _ptr(out), _stream(out)), "nerf_mlp_forward")
        if ev is not None:
            e1.record(torch.cuda.current_stream(in0.device))
            ev.append((e0, e1, R * S))
    return out


def mlp_forward_objects(packed_list, precision, use_viewdirs, output_ch, mode, in0, in1, R, S, ray_cols=0):
    """nerf_mlp_forward_objects: len(packed_list) networks, each evaluating its equal share of the R rays, one launch.
    packed_list: packed streams (mlp_pack) on the device of in0."""
    in0 = _f32(in0, "mlp input")
    in1 = _f32(in1, "mlp second input", in0.device) if in1 is not None else None
    n = len(packed_list)
    _mlp_operands_ok("mlp_forward_objects", mode, use_viewdirs, in0, in1, R, S, ray_cols)
    _need(all(p.is_cuda and p.device == in0.device for p in packed_list), "mlp_forward_objects: packed networks on another device")
    table = torch.tensor([p.data_ptr() for p in packed_list], dtype=torch.int64).to(in0.device)
    out = torch.empty((R * S, output_ch), dtype=torch.float32, device=in0.device)
    ev = MLP_EVENTS
    with torch.cuda.device(in0.device):
        if ev is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(in0.device))
        _check(lib().nerf_mlp_forward_objects(ctypes.cast(ctypes.c_void_p(table.data_ptr()), ctypes.POINTER(ctypes.c_void_p)),
                                              n, precision, int(bool(use_viewdirs)), int(output_ch), mode, _ptr(in0),
                                              _ptr(in1), int(R), int(S), int(ray_cols), _ptr(out), _stream(out)),
               "nerf_mlp_forward_objects")
        if ev is not None:
            e1.record(torch.cuda.current_stream(in0.device))
            ev.append((e0, e1, R * S))
    out._nerf_keepalive = (table, list(packed_list))      # the launch reads them asynchronously
    return out


# --------------------------------------------------------------------------- compositing / sampling
def raw2outputs(raw, z, d, d_stride, noise=None, white_bkgd=False, want_weights=True, want_rgb=True,
                want_depth=True):
    """raw [R,S,4|5], z [R,S]; d: tensor whose data pointer is the first direction
    component of ray 0 and whose rows are d_stride floats apart."""
    raw, z = _f32(raw, "raw"), _f32(z, "z_vals", raw.device)
    R, S, ch = raw.shape
    if tuple(z.shape) != (R, S):
        raise ValueError("z_vals %s does not match raw %s" % (tuple(z.shape), tuple(raw.shape)))
    dev = raw.device
    noise = _f32(noise, "noise", dev) if noise is not None else None
    _need(noise is None or noise.numel() == R * S, "raw2outputs: noise must hold %d x %d values", R, S)
    f = dict(dtype=torch.float32, device=dev)
    rgb = torch.empty((R, 3), **f) if want_rgb else None
    acc = torch.empty((R,), **f)
    depth = torch.empty((R,), **f) if want_depth else None
    disp = torch.empty((R,), **f) if want_depth else None
    # one sample per ray: the reference's interval list is empty and so are its weights (run_nerf.py:283-284)
    weights = torch.empty((R, S if S > 1 else 0), **f) if want_weights else None
    with torch.cuda.device(dev):
        _check(lib().nerf_raw2outputs(_ptr(raw), ch, _ptr(z), _dir_ptr(d, dev, R, int(d_stride)), int(d_stride),
                                      _ptr(noise), R, S, int(bool(white_bkgd)), _ptr(rgb), _ptr(disp), _ptr(acc),
                                      _ptr(depth), _ptr(weights), _stream(raw)), "nerf_raw2outputs")
    return rgb, disp, acc, weights, depth


def sample_pdf(bins, weights, u, bins_are_z=False, z_vals=None, want_samples=True, want_inds=False,
               want_sorted=False, want_std=False):
    """bins [R,nb] (or z_vals [R,nb+1] with bins_are_z), weights: 2-D float32
    view with unit inner stride (row stride may exceed its width); u [n] or [R,n]."""
    if not (bins.is_cuda and weights.is_cuda):
        raise ValueError("sample_pdf inputs must live on a GPU; this engine has no CPU path")
    if bins.dtype != torch.float32 or weights.dtype != torch.float32:
        raise ValueError("sample_pdf inputs must be float32")
    if bins.stride(-1) != 1:
        bins = bins.contiguous()
    if weights.stride(-1) != 1:
        weights = weights.contiguous()
    R = bins.shape[0]
    nb = bins.shape[1] - (1 if bins_are_z else 0)
    if weights.shape[0] != R or weights.shape[1] != nb - 1:
        raise ValueError("weights must be [%d, %d], got %s" % (R, nb - 1, tuple(weights.shape)))
    dev = bins.device
    u = _f32(u, "u", dev)
    per_ray = int(u.dim() == 2)
    n_u = u.shape[-1]
    if per_ray and u.shape[0] != R:
        raise ValueError("u must be [%d, n] or [n]" % R)
    f = dict(dtype=torch.float32, device=dev)
    zs = torch.empty((R, n_u), **f) if want_samples else None
    inds = torch.empty((R, n_u), dtype=torch.int32, device=dev) if want_inds else None
    n_z = 0
    zsorted = None
    if want_sorted:
        z_vals = _f32(z_vals, "z_vals", dev)
        _need(z_vals.dim() == 2 and z_vals.shape[0] == R, "sample_pdf: z_vals must be [%d, n]", R)
        n_z = z_vals.shape[1]
        zsorted = torch.empty((R, n_z + n_u), **f)
    std = torch.empty((R,), **f) if want_std else None
    with torch.cuda.device(dev):
        _check(lib().nerf_sample_pdf(_ptr(bins), bins.stride(0), int(bool(bins_are_z)), nb, _ptr(weights),
                                     weights.stride(0), _ptr(u), n_u, per_ray, R, _ptr(z_vals) if want_sorted else None,
                                     n_z, _ptr(zs), None if inds is None else ctypes.c_void_p(inds.data_ptr()),
                                     _ptr(zsorted), _ptr(std), _stream(bins)), "nerf_sample_pdf")
    return zs, inds, zsorted, std


def render_rays_fused(packed_coarse, packed_fine, rays, n_samples, n_importance, precision, use_viewdirs, output_ch,
                      t_vals, u=None, t_rand=None, noise0=None, noise1=None, lindisp=False, white_bkgd=False,
                      want_weights=False):
    """nerf_render_rays: the whole of render_rays (run_nerf.py:314-424) as one library call on rays [R, 8|11].
    Returns the reference's dict (rgb_map, disp_map, acc_map, depth_map, raw, z_vals and, with n_importance > 0, rgb0,
    disp0, acc0, z_std; weights of the coarse pass on request).  Same kernels, same bits as the calls one by one."""
    rays = _f32(rays, "rays")
    dev = rays.device
    R, cols = rays.shape
    S, NI = int(n_samples), int(n_importance)
    _need(cols >= (11 if use_viewdirs else 8), "render_rays_fused: rays need %d columns, got %d", 11 if use_viewdirs else 8, cols)
    t_vals = _f32(t_vals, "t_vals", dev)
    _need(t_vals.numel() == S, "render_rays_fused: t_vals must have %d entries", S)
    need_bytes = lib().nerf_mlp_packed_bytes(int(precision), int(bool(use_viewdirs)))
    for name, pk in (("packed_coarse", packed_coarse), ("packed_fine", packed_fine)):
        _need(pk is None or (pk.is_cuda and pk.device == dev and need_bytes > 0 and pk.numel() * pk.element_size() >= need_bytes),
              "render_rays_fused: %s is not a packed network of this precision on %s", name, dev)
    _need(packed_coarse is not None, "render_rays_fused: packed_coarse is missing")
    opt = lambda t, name, shape: None if t is None else _shaped(_f32(t, name, dev), name, shape)
    t_rand = opt(t_rand, "t_rand", (R, S))
    noise0 = opt(noise0, "noise0", (R, S))
    noise1 = opt(noise1, "noise1", (R, S + NI))
    per_ray = 0
    if NI > 0:
        _need(u is not None and S >= 3, "render_rays_fused: importance sampling needs u and at least 3 coarse samples")
        u = _f32(u, "u", dev)
        per_ray = int(u.dim() == 2)
        _need(tuple(u.shape) == ((R, NI) if per_ray else (NI,)), "render_rays_fused: u must be [%d] or [%d, %d]", NI, R, NI)
    f = dict(dtype=torch.float32, device=dev)
    ST = S + NI
    out = {"rgb_map": torch.empty((R, 3), **f), "disp_map": torch.empty((R,), **f), "acc_map": torch.empty((R,), **f),
           "depth_map": torch.empty((R,), **f), "raw": torch.empty((R, ST, output_ch), **f), "z_vals": torch.empty((R, ST), **f)}
    work = {}
    if NI > 0:
        out.update(rgb0=torch.empty((R, 3), **f), disp0=torch.empty((R,), **f), acc0=torch.empty((R,), **f),
                   z_std=torch.empty((R,), **f))
        work = {"z_coarse": torch.empty((R, S), **f), "raw_coarse": torch.empty((R, S, output_ch), **f)}
    if NI > 0 or want_weights:
        work["weights"] = torch.empty((R, S), **f)
    p = lambda t: None if t is None else t.data_ptr()
    cfg = RenderCfg(S, NI, int(precision), int(bool(use_viewdirs)), int(output_ch), int(cols), int(bool(lindisp)),
                    int(bool(white_bkgd)), per_ray, p(t_vals), p(t_rand), p(u), p(noise0), p(noise1))
    buf = RenderBuffers(p(out["rgb_map"]), p(out["disp_map"]), p(out["acc_map"]), p(out["depth_map"]), p(out.get("rgb0")),
                        p(out.get("disp0")), p(out.get("acc0")), p(out.get("z_std")), p(out["raw"]), p(out["z_vals"]),
                        p(work.get("z_coarse")), p(work.get("raw_coarse")), p(work.get("weights")))
    with torch.cuda.device(dev):
        _check(lib().nerf_render_rays(ctypes.c_void_p(packed_coarse.data_ptr()),
                                      None if packed_fine is None else ctypes.c_void_p(packed_fine.data_ptr()), _ptr(rays), R,
                                      ctypes.byref(cfg), ctypes.byref(buf), _stream(rays)), "nerf_render_rays")
    if want_weights:
        out["weights0" if NI > 0 else "weights"] = work["weights"]
    out["_work"] = work          # the launches read and write these asynchronously: keep them alive with the result
    return out


def _shaped(t, name, shape):
    _need(tuple(t.shape) == tuple(shape), "render_rays_fused: %s must be %s, got %s", name, tuple(shape), tuple(t.shape))
    return t


# --------------------------------------------------------------------------- training
def raw2outputs_backward(raw, z, d, d_stride, noise, white_bkgd, g_rgb=None, g_disp=None, g_acc=None, g_depth=None,
                         g_weights=None):
    """dL/draw [R,S,ch] from the upstream gradients of the maps and of the per-sample weights [R,S] (None = zero)."""
    raw, z = _f32(raw, "raw"), _f32(z, "z_vals", raw.device)
    R, S, ch = raw.shape
    dev = raw.device
    _need(tuple(z.shape) == (R, S), "raw2outputs_backward: z_vals %s does not match raw %s", tuple(z.shape), tuple(raw.shape))
    gs = [None if g is None else _f32(g, "upstream gradient", dev) for g in (g_rgb, g_disp, g_acc, g_depth, g_weights)]
    for g, n_el in zip(gs, (3 * R, R, R, R, R * S)):
        _need(g is None or g.numel() == n_el, "raw2outputs_backward: an upstream gradient holds %d values, %d expected",
              0 if g is None else g.numel(), n_el)
    noise = _f32(noise, "noise", dev) if noise is not None else None
    _need(noise is None or noise.numel() == R * S, "raw2outputs_backward: noise must hold %d x %d values", R, S)
    d_raw = torch.empty_like(raw)
    with torch.cuda.device(dev):
        _check(lib().nerf_raw2outputs_backward(_ptr(raw), ch, _ptr(z), _dir_ptr(d, dev, R, int(d_stride)), int(d_stride),
                                               _ptr(noise), R, S, int(bool(white_bkgd)), _ptr(gs[0]), _ptr(gs[1]),
                                               _ptr(gs[2]), _ptr(gs[3]), _ptr(gs[4]), _ptr(d_raw), _stream(raw)),
               "nerf_raw2outputs_backward")
    return d_raw


def mlp_pack_backward(tensors, use_viewdirs, output_ch, precision="fp32"):
    """tensors: state_dict-ordered list (24 with view directions, 20 without; unused entries may be None)."""
    prec = PRECISIONS[precision]
    dev = next(t for t in tensors if t is not None).device
    keep = [None if t is None else _f32(t.detach(), "weight %d" % i, dev) for i, t in enumerate(tensors)]
    arr = (ctypes.c_void_p * len(keep))(*[None if t is None else t.data_ptr() for t in keep])
    packed = torch.empty((lib().nerf_mlp_backward_packed_bytes(int(bool(use_viewdirs)), prec),), dtype=torch.uint8,
                         device=dev)
    with torch.cuda.device(dev):
        _check(lib().nerf_mlp_pack_backward(arr, int(bool(use_viewdirs)), int(output_ch), prec,
                                            ctypes.c_void_p(packed.data_ptr()), _stream(packed)), "nerf_mlp_pack_backward")
    return packed


def mlp_forward_train(packed, use_viewdirs, output_ch, mode, in0, in1, R, S, ray_cols=0, precision="fp32"):
    """Forward that keeps its activations: returns (raw [R*S, output_ch], tape).  `packed` must come from mlp_pack
    with the same precision."""
    prec = PRECISIONS[precision]
    in0 = _f32(in0, "mlp input")
    in1 = _f32(in1, "mlp second input", in0.device) if in1 is not None else None
    M = R * S
    vd = int(bool(use_viewdirs))
    _mlp_operands_ok("mlp_forward_train", mode, use_viewdirs, in0, in1, R, S, ray_cols)
    out = torch.empty((M, output_ch), dtype=torch.float32, device=in0.device)
    tape = torch.empty((lib().nerf_train_tape_bytes(M, vd, prec) // 4,), dtype=torch.float32, device=in0.device)
    with torch.cuda.device(in0.device):
        _check(lib().nerf_mlp_forward_train(ctypes.c_void_p(packed.data_ptr()), prec, vd, int(output_ch), mode, _ptr(in0),
                                            _ptr(in1), int(R), int(S), int(ray_cols), _ptr(out),
                                            ctypes.c_void_p(tape.data_ptr()), _stream(out)), "nerf_mlp_forward_train")
    return out, tape


_WORKSPACE = {}


def mlp_backward(packed_bwd, use_viewdirs, output_ch, d_raw, tape, shapes, precision="fp32", packed_fwd=None, out=None):
    """Gradients of the parameter tensors: list in state_dict order shaped like `shapes` (None entries are
    skipped and come back as None).  bf16: packed_fwd is the forward stream the tape was made with.  out: tensors to
    write the gradients into (contiguous float32, e.g. views of one flat buffer) instead of new ones."""
    prec = PRECISIONS[precision]
    if precision == "bf16" and packed_fwd is None:
        raise ValueError("mlp_backward: the bf16 backward needs packed_fwd")
    d_raw = _f32(d_raw, "d_raw")
    dev = d_raw.device
    M = d_raw.numel() // output_ch
    vd = int(bool(use_viewdirs))
    if out is not None:
        _need(len(out) == len(shapes) and all((g is None) == (s is None) for g, s in zip(out, shapes)),
              "mlp_backward: `out` does not match the parameter list")
        _need(all(g is None or (g.is_cuda and g.device == dev and g.dtype == torch.float32 and g.is_contiguous() and
                                tuple(g.shape) == tuple(s)) for g, s in zip(out, shapes)),
              "mlp_backward: `out` tensors must be contiguous float32 on %s, shaped like the parameters", dev)
    if M == 0:      # an empty shard of a data-parallel batch: zero gradients, so the all-reduce still lines up
        if out is not None:
            for g in out:
                if g is not None:
                    g.zero_()
            return list(out)
        return [None if s is None else torch.zeros(tuple(s), dtype=torch.float32, device=dev) for s in shapes]
    _need(tape.is_cuda and tape.device == dev and tape.numel() * tape.element_size() >= lib().nerf_train_tape_bytes(M, vd, prec),
          "mlp_backward: the tape does not belong to a forward of %d samples on %s", M, dev)
    gtape = torch.empty((lib().nerf_train_grad_tape_bytes(M, vd, prec) // 4,), dtype=torch.float32, device=dev)
    ws = _WORKSPACE.get(str(dev))
    if ws is None:
        ws = _WORKSPACE[str(dev)] = torch.empty((lib().nerf_train_workspace_bytes() // 4,), dtype=torch.float32,
                                                device=dev)
    grads = list(out) if out is not None else [None if s is None else torch.empty(tuple(s), dtype=torch.float32, device=dev)
                                               for s in shapes]
    arr = (ctypes.c_void_p * len(grads))(*[None if g is None else g.data_ptr() for g in grads])
    with torch.cuda.device(dev):
        _check(lib().nerf_mlp_backward(ctypes.c_void_p(packed_bwd.data_ptr()),
                                       ctypes.c_void_p(packed_fwd.data_ptr()) if packed_fwd is not None else None,
                                       prec, vd, int(output_ch), _ptr(d_raw), M, ctypes.c_void_p(tape.data_ptr()),
                                       ctypes.c_void_p(gtape.data_ptr()), _ptr(ws), ws.numel() * 4, arr,
                                       _stream(d_raw)), "nerf_mlp_backward")
    return grads


STATIC_MAX_SAMPLES = 8192


def mlp_backward_static(packed_bwd, use_viewdirs, output_ch, d_raw, tape, grads, precision="fp32", packed_fwd=None):
    """nerf_mlp_backward_static: as mlp_backward, for callers that may not allocate: the gradient tensors `grads`
    (state_dict order, None for unused slots) are passed in and overwritten, the scratch is the library's static
    device array.  At most STATIC_MAX_SAMPLES samples per call."""
    prec = PRECISIONS[precision]
    d_raw = _f32(d_raw, "d_raw")
    M = d_raw.numel() // output_ch
    vd = int(bool(use_viewdirs))
    _need(tape.is_cuda and tape.device == d_raw.device and
          tape.numel() * tape.element_size() >= lib().nerf_train_tape_bytes(max(M, 1), vd, prec),
          "mlp_backward_static: the tape does not belong to a forward of %d samples on %s", M, d_raw.device)
    _need(all(g is None or (g.is_cuda and g.device == d_raw.device and g.dtype == torch.float32 and g.is_contiguous())
              for g in grads), "mlp_backward_static: gradient tensors must be contiguous float32 on %s", d_raw.device)
    arr = (ctypes.c_void_p * len(grads))(*[None if g is None else g.data_ptr() for g in grads])
    with torch.cuda.device(d_raw.device):
        _check(lib().nerf_mlp_backward_static(ctypes.c_void_p(packed_bwd.data_ptr()),
                                              ctypes.c_void_p(packed_fwd.data_ptr()) if packed_fwd is not None else None,
                                              prec, int(bool(use_viewdirs)), int(output_ch), _ptr(d_raw), M,
                                              ctypes.c_void_p(tape.data_ptr()), arr, _stream(d_raw)),
               "nerf_mlp_backward_static")
    return grads


def adam_step(params, grads, exp_avgs, exp_avg_sqs, step, lr, beta1, beta2, eps, weight_decay=0.0):
    """In-place Adam update of up to 64 float32 tensors per launch (all on one device, contiguous)."""
    n = len(params)
    for i in range(0, n, 64):
        sl = slice(i, min(n, i + 64))
        ps, gs, ms, vs = params[sl], grads[sl], exp_avgs[sl], exp_avg_sqs[sl]
        for t in ps + gs + ms + vs:
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.device == ps[0].device):
                raise ValueError("adam_step needs contiguous float32 tensors on one GPU")
        for p_, g_, m_, v_ in zip(ps, gs, ms, vs):
            _need(g_.numel() == p_.numel() == m_.numel() == v_.numel(), "adam_step: gradient / moment sizes differ from the parameter's")
        k = len(ps)
        mk = lambda ts: (ctypes.c_void_p * k)(*[t.data_ptr() for t in ts])
        numel = (ctypes.c_longlong * k)(*[t.numel() for t in ps])
        with torch.cuda.device(ps[0].device):
            _check(lib().nerf_adam_step(k, mk(ps), mk(gs), mk(ms), mk(vs), numel, int(step), float(lr), float(beta1),
                                        float(beta2), float(eps), float(weight_decay), _stream(ps[0])), "nerf_adam_step")
    WEIGHT_GENERATION[0] += 1
