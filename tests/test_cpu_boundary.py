"""CPU-side checks of the C-ABI boundary and of the host logic: the library loads without a
GPU and exports every symbol the header declares, the ctypes table agrees with the header,
host wrappers refuse CPU tensors instead of falling back, and the shipped package never
touches the oracle.  Nothing here launches a kernel."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

import nerf_loc_amd
from nerf_loc_amd import helpers, hip, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "nerf_hip.h")
PKG = os.path.join(ROOT, "nerf-loc_amd")


def _header_text():
    with open(HEADER) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return "\n".join(l for l in text.splitlines() if not l.lstrip().startswith("#"))


def _header_decls():
    """{name: (return type, [argument types])} parsed from the header's prototypes."""
    decls = {}
    for m in re.finditer(r"((?:const\s+)?(?:int|size_t|char)\s*\*?)\s*\b(nerf_\w+)\s*\(([^)]*)\)\s*;", _header_text()):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        arglist = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        decls[name] = (ret, arglist)
    return decls


def _ctype_of(decl):
    d = decl.strip()
    if "*" in d:
        return "ptr"
    d = re.sub(r"\b\w+$", "", d).strip() if len(d.split()) > 1 else d   # drop the parameter name
    return {"int": "int", "float": "float", "long long": "longlong", "size_t": "size_t"}[d]


_CT = {ctypes.c_int: "int", ctypes.c_float: "float", ctypes.c_longlong: "longlong", ctypes.c_size_t: "size_t",
       ctypes.c_void_p: "ptr", ctypes.c_char_p: "ptr", ctypes.POINTER(ctypes.c_void_p): "ptr"}


def test_header_declares_what_binding_binds():
    decls = _header_decls()
    assert len(decls) >= 11
    assert sorted(decls) == hip.exported_symbols()


def test_binding_signatures_match_header():
    for name, (ret, args) in _header_decls().items():
        res, argtypes = hip._SIGNATURES[name]
        kinds = ["ptr" if (isinstance(a, type) and issubclass(a, ctypes._Pointer)) else _CT[a] for a in argtypes]
        assert kinds == [_ctype_of(a) for a in args], name
        assert _CT[res] == ("ptr" if "*" in ret else _ctype_of(ret + " x")), name


def test_library_loads_and_exports_every_symbol():
    assert os.path.exists(hip.LIB_PATH), "build the library first: python -c 'import __graft_entry__ as g; g.build()'"
    L = ctypes.CDLL(hip.LIB_PATH)
    for name in _header_decls():
        assert hasattr(L, name), "libnerf_hip.so does not export %s" % name
    lib = hip.lib()
    assert lib.nerf_hip_abi_version() == hip.ABI_VERSION
    assert lib.nerf_last_error() is not None


def test_packed_sizes_host_only():
    lib = hip.lib()
    frag, bias = 1024, 128
    scales = 64          # pack-time scratch behind the bias block
    rowsums = 11 * 256 * 2 * 4          # pack-time scratch of the bf16 stream
    # bf16: the 32x32x16 stream the training kernels read, then the 16x16x32 stream of the inference kernel with its
    # own bias tiles (154 / 129 tiles of 16 floats, padded to 128 bytes)
    assert lib.nerf_mlp_packed_bytes(hip.PREC_BF16, 1) == 1184 * frag + 78 * bias + scales + rowsums + 1184 * frag + 9856
    assert lib.nerf_mlp_packed_bytes(hip.PREC_BF16, 0) == 976 * frag + 65 * bias + scales + rowsums + 976 * frag + 8320
    assert lib.nerf_mlp_packed_bytes(hip.PREC_F32, 1) == 2368 * frag + 78 * bias + scales
    assert lib.nerf_mlp_packed_bytes(hip.PREC_F32, 0) == 1952 * frag + 65 * bias + scales
    # split-operand streams: 3 (2) bf16 terms of each of the 1172 / 968 fragments, padded to whole 16-fragment stages,
    # then the 16-float bias tiles
    assert lib.nerf_mlp_packed_bytes(hip.PREC_BF16X3, 1) == 3520 * frag + 9856
    assert lib.nerf_mlp_packed_bytes(hip.PREC_BF16X3, 0) == 2912 * frag + 8320
    assert lib.nerf_mlp_packed_bytes(hip.PREC_BF16X2, 1) == 2352 * frag + 9856
    assert lib.nerf_mlp_packed_bytes(hip.PREC_BF16X2, 0) == 1936 * frag + 8320
    assert lib.nerf_mlp_packed_bytes(17, 1) == 0
    assert hip.PRECISIONS["bf16x3"] == hip.PREC_BF16X3 == 2 and hip.PRECISIONS["bf16x2"] == hip.PREC_BF16X2 == 3


def test_invalid_arguments_are_refused_before_any_launch():
    lib = hip.lib()
    # no pointer is ever dereferenced on the host: null outputs / bad sizes must come back as NERF_E_INVALID
    assert lib.nerf_embed(None, 4, 10, 1, None, None) == -1
    assert b"" != lib.nerf_last_error()
    assert lib.nerf_mlp_forward(None, 0, 1, 4, 0, None, None, 8, 1, 0, None, None) == -1
    assert lib.nerf_mlp_forward(ctypes.c_void_p(256), 7, 1, 4, 0, ctypes.c_void_p(256), None, 8, 1, 0,
                                ctypes.c_void_p(256), None) == -1
    assert lib.nerf_raw2outputs(ctypes.c_void_p(256), 3, ctypes.c_void_p(256), ctypes.c_void_p(256), 3, None, 4, 8, 0,
                                None, None, None, None, None, None) == -1
    assert lib.nerf_sample_pdf(ctypes.c_void_p(256), 8, 0, 1, ctypes.c_void_p(256), 8, ctypes.c_void_p(256), 4, 0, 4,
                               None, 0, None, None, None, None, None) == -1


def test_training_entry_points_check_precision_and_sizes():
    """Host-only checks of the training side of the ABI: tape sizes per precision, unknown precisions refused."""
    lib = hip.lib()
    f32, b16 = hip.PREC_F32, hip.PREC_BF16
    # float32 tapes: sample-major rows padded to 128 samples (pe 64 + view 32 + 8 x 256 + 256 + 128 floats, then
    # 8 dwords of ReLU bits per ReLU layer)
    assert lib.nerf_train_tape_bytes(1, 1, f32) == 128 * (64 + 32 + 8 * 256 + 256 + 128 + 9 * 8) * 4
    assert lib.nerf_train_tape_bytes(129, 0, f32) == 256 * (64 + 8 * 256 + 8 * 8) * 4
    assert lib.nerf_train_grad_tape_bytes(1, 1, f32) == 128 * (8 * 256 + 256 + 128) * 4
    # bf16 tapes: 2 KiB operand pieces per 64 samples, padded to 256 samples (167 / 156 with view directions)
    assert lib.nerf_train_tape_bytes(1, 1, b16) == 4 * 167 * 2048
    assert lib.nerf_train_tape_bytes(257, 0, b16) == 8 * 140 * 2048
    assert lib.nerf_train_grad_tape_bytes(256, 1, b16) == 4 * 156 * 2048
    assert lib.nerf_train_grad_tape_bytes(256, 0, b16) == 4 * 130 * 2048
    assert lib.nerf_mlp_backward_packed_bytes(1, f32) == 2192 * 1024 and lib.nerf_mlp_backward_packed_bytes(0, f32) == 1808 * 1024
    assert lib.nerf_mlp_backward_packed_bytes(1, b16) == 1104 * 1024 and lib.nerf_mlp_backward_packed_bytes(0, b16) == 912 * 1024
    # bf16x3 trains on the float32 tapes; its input-gradient chain has a stream of its own (3 terms x 1112 / 912
    # transposed 16 x 32 fragments, padded to whole stages of 16); bf16x2 is an inference mode only
    x3 = hip.PREC_BF16X3
    for M in (1, 129, 5000):
        for vd in (0, 1):
            assert lib.nerf_train_tape_bytes(M, vd, x3) == lib.nerf_train_tape_bytes(M, vd, f32)
            assert lib.nerf_train_grad_tape_bytes(M, vd, x3) == lib.nerf_train_grad_tape_bytes(M, vd, f32)
    assert lib.nerf_mlp_backward_packed_bytes(1, x3) == 3344 * 1024 and lib.nerf_mlp_backward_packed_bytes(0, x3) == 2736 * 1024
    assert lib.nerf_train_tape_bytes(100, 1, hip.PREC_BF16X2) == 0 and lib.nerf_mlp_backward_packed_bytes(1, hip.PREC_BF16X2) == 0
    for fn in (lib.nerf_train_tape_bytes, lib.nerf_train_grad_tape_bytes):
        assert fn(100, 1, 7) == 0 and fn(-1, 1, f32) == 0
    assert lib.nerf_mlp_backward_packed_bytes(1, 7) == 0
    p = ctypes.c_void_p(256)
    arr = (ctypes.c_void_p * 24)(*[256] * 24)
    assert lib.nerf_mlp_pack_backward(arr, 1, 4, 7, p, None) == -1
    assert b"precision" in lib.nerf_last_error()
    fp = ctypes.cast(p, ctypes.POINTER(ctypes.c_float))
    assert lib.nerf_mlp_forward_train(p, 7, 1, 4, 0, fp, None, 8, 1, 0, fp, p, None) == -1
    assert lib.nerf_mlp_backward(p, None, b16, 1, 4, fp, 8, p, p, fp, 1 << 30, arr, None) == -1      # no forward stream
    assert b"forward stream" in lib.nerf_last_error()
    with pytest.raises(ValueError):
        from nerf_loc_amd import train
        train.set_train_precision("fp8")


def test_missing_library_fails_loudly():
    code = ("import nerf_loc_amd\nfrom nerf_loc_amd import hip\n"
            "try:\n    hip.lib()\nexcept hip.NerfHipError as e:\n    print('LOUD', e)\n")
    env = dict(os.environ, NERF_HIP_LIB="/nonexistent/libnerf_hip.so", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert "LOUD" in out.stdout and "no CPU fallback" in out.stdout, out.stdout + out.stderr


def test_cpu_tensors_are_refused_not_computed():
    x = torch.zeros(4, 3)
    with pytest.raises(ValueError, match="GPU"):
        hip.embed(x, 10)
    with pytest.raises(ValueError, match="GPU"):
        helpers.get_embedder(10)[0](x)
    with pytest.raises(ValueError, match="GPU"):
        helpers.sample_pdf(torch.zeros(2, 9), torch.zeros(2, 8), 4, det=True)
    net = helpers.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, use_viewdirs=True)
    with torch.no_grad(), pytest.raises(ValueError, match="GPU"):
        net(torch.zeros(2, 90))


def test_product_never_imports_the_oracle():
    bad = []
    for base, _, files in os.walk(PKG):
        for fn in files:
            if fn.endswith((".py", ".hip", ".cpp", ".h")):
                with open(os.path.join(base, fn)) as f:
                    if re.search(r"\boracle\b", f.read()):
                        bad.append(fn)
    assert not bad, bad


# ---------------------------------------------------------------------------------- host logic
def test_nerf_module_matches_reference_checkpoint_layout():
    net = helpers.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = synth.make_state_dict(3, "A", use_viewdirs=True)
    assert list(shapes) == list(sd)                                   # names AND order (run_nerf_helpers.py:77-94)
    assert shapes == {k: tuple(v.shape) for k, v in sd.items()}
    assert shapes["pts_linears.5.weight"] == (256, 319) and shapes["views_linears.0.weight"] == (128, 283)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    novd = helpers.NeRF(D=8, W=256, input_ch=63, input_ch_views=0, output_ch=5, skips=[4], use_viewdirs=False)
    assert "output_linear.weight" in novd.state_dict() and novd.out_channels == 5 and novd.hip_supported()
    assert not helpers.NeRF(D=4, W=128, input_ch=63).hip_supported()
    # other shapes are evaluated with library GEMMs, on a GPU only: a CPU tensor is refused, not computed
    with torch.no_grad(), pytest.raises(ValueError, match="computes on GPUs only"):
        helpers.NeRF(D=4, W=128, input_ch=63)(torch.zeros(1, 63))
    with pytest.raises(NotImplementedError):
        helpers.NeRF(D=4, W=128, input_ch=63).packed_weights()


def test_get_embedder_interface():
    fn, dim = helpers.get_embedder(10, 0)
    assert dim == 63 and callable(fn) and fn.embedder.num_freqs == 10
    fn, dim = helpers.get_embedder(4, 0)
    assert dim == 27
    ident, dim = helpers.get_embedder(10, -1)
    assert dim == 3 and torch.equal(ident(torch.ones(2, 3)), torch.ones(2, 3))
    with pytest.raises(NotImplementedError):
        helpers.Embedder(include_input=True, input_dims=2, max_freq_log2=3, num_freqs=4, log_sampling=True,
                         periodic_fns=[torch.sin, torch.cos])


def test_misc_helpers_and_get_rays_np(golden):
    x, y = torch.tensor([0.0, 0.5, 1.0]), torch.tensor([0.0, 0.0, 0.0])
    mse = helpers.img2mse(x, y)
    assert abs(float(mse) - (0.25 + 1.0) / 3) < 1e-7
    assert abs(float(helpers.mse2psnr(mse)) + 10 * np.log10((0.25 + 1.0) / 3)) < 1e-5
    assert helpers.to8b(np.array([-1.0, 0.5, 2.0])).tolist() == [0, 127, 255]
    g = golden("g6_rays")
    H, W = [int(v) for v in g["lego40_hw"]]
    o, d = helpers.get_rays_np(H, W, g["lego40_K"], g["lego40_c2w"])
    assert np.array_equal(np.asarray(o).reshape(-1, 3), g["lego40_o_np"].reshape(-1, 3))
    assert np.array_equal(np.asarray(d).reshape(-1, 3), g["lego40_d_np"].reshape(-1, 3))
    assert np.asarray(d).dtype == g["lego40_d_np"].dtype


def test_precision_switch():
    assert helpers.get_default_precision() == "fp32"
    helpers.set_default_precision("bf16")
    try:
        net = helpers.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, use_viewdirs=True)
        assert net.precision_code() == hip.PREC_BF16
        net.precision = "fp32"
        assert net.precision_code() == hip.PREC_F32
    finally:
        helpers.set_default_precision("fp32")
    with pytest.raises(ValueError):
        helpers.set_default_precision("fp8")


def test_training_entry_points_say_what_is_missing():
    net = helpers.NeRF(D=8, W=256, input_ch=63, input_ch_views=27, use_viewdirs=True)
    with pytest.raises(NotImplementedError):
        net(torch.zeros(2, 90, requires_grad=True))


def test_dropin_alias_modules_resolve_to_the_engine(tmp_path):
    """dropin/ first on sys.path is the zero-edit swap.  The reference's `utils` is a namespace directory without an
    __init__.py holding many more modules (utils.dist, utils.misc, ...): dropin/utils must merge with it, not shadow it."""
    other = tmp_path / "ref_side" / "utils"
    other.mkdir(parents=True)
    (other / "misc.py").write_text("MARK = 'reference utils.misc'\n")
    (other / "nerf_utils.py").write_text("raise ImportError('the reference copy must lose to dropin/utils/nerf_utils.py')\n")
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(1, %r)\n"
            "import run_nerf, run_multi_nerf, run_multi_nerf_scannet, run_nerf_helpers\n"
            "from run_nerf_helpers import *\n"
            "from utils.nerf_utils import create_nerf_model, load_nerf_model, render_path\n"
            "import utils.misc\n"
            "assert utils.misc.MARK == 'reference utils.misc'\n"
            "assert create_nerf_model.__module__.startswith('nerf_loc_amd')\n"
            "assert NeRF.__module__.endswith('helpers') and run_multi_nerf.run_network is run_nerf.run_network\n"
            "assert run_multi_nerf_scannet.render_rays is run_nerf.render_rays\n"
            "for n in ('render', 'render_rays', 'batchify', 'batchify_rays', 'raw2outputs', 'create_nerf', 'render_path',\n"
            "          'sample_pdf', 'get_embedder', 'get_rays', 'get_rays_np', 'ndc_rays', 'img2mse', 'mse2psnr', 'to8b'):\n"
            "    assert callable(getattr(run_nerf, n)), n\n"
            "print('ALIAS OK')\n") % (os.path.join(ROOT, "dropin"), str(tmp_path / "ref_side"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd="/tmp")
    assert "ALIAS OK" in out.stdout, out.stdout + out.stderr
    assert not os.path.exists(os.path.join(ROOT, "dropin", "utils", "__init__.py"))


def test_graft_entry_build_from_a_clean_tree():
    """build() is the documented way to produce the library: it must succeed from `make clean` and leave a library
    whose ABI version is the binding's (a literal version in the entry point once went stale)."""
    import __graft_entry__ as g
    env = dict(os.environ, PYTHONPATH=ROOT)
    code = ("import subprocess, sys, os\n"
            "subprocess.run(['make', '-C', os.path.join(%r, 'nerf-loc_amd', 'csrc'), 'clean'], check=True, stdout=subprocess.DEVNULL)\n"
            "import __graft_entry__ as g\n"
            "g.build()\n"
            "from nerf_loc_amd import hip\n"
            "assert hip.lib().nerf_hip_abi_version() == hip.ABI_VERSION\n"
            "print('BUILD OK')\n") % ROOT
    if os.environ.get("NERF_SKIP_CLEAN_BUILD_TEST"):
        pytest.skip("NERF_SKIP_CLEAN_BUILD_TEST set")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert "BUILD OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
    assert callable(g.smoke)


def test_ray_bank_batching_mode_follows_the_reference_draws():
    """use_batching=True (run_nerf.py:688-737) is host + indexing logic: same numpy shuffle, same epoch handling."""
    from nerf_loc_amd import raybank
    H, W = 6, 5
    K = np.array([[7.0, 0, 2.5], [0, 7.0, 3.0], [0, 0, 1]])
    rs = np.random.RandomState(0)
    images = rs.uniform(size=(4, H, W, 3)).astype(np.float32)
    poses = np.stack([synth.pose_spherical(30.0 * i, -30.0, 4.0)[:3, :4] for i in range(4)]).astype(np.float32)
    i_train = [0, 2, 3]
    np.random.seed(5)
    bank = raybank.RayBank(images, poses, H, W, K, i_train, N_rand=32, use_batching=True, device="cpu")
    # restatement of the reference's preparation with the same seed
    np.random.seed(5)
    rays = np.stack([helpers.get_rays_np(H, W, K, p) for p in poses[:, :3, :4]], 0)
    ref = np.transpose(np.concatenate([rays, images[:, None]], 1), [0, 2, 3, 1, 4])
    ref = np.reshape(np.stack([ref[i] for i in i_train], 0), [-1, 3, 3]).astype(np.float32)
    np.random.shuffle(ref)
    assert len(bank) == 3 * H * W == ref.shape[0]
    b, t = bank.next(1)
    assert tuple(b.shape) == (2, 32, 3) and tuple(t.shape) == (32, 3)
    assert np.array_equal(b.numpy(), np.transpose(ref[:32], [1, 0, 2])[:2]) and np.array_equal(t.numpy(), ref[:32, 2])
    bank.next(2)
    torch.manual_seed(3)
    b3, _ = bank.next(3)            # 64..96 of 90 rows: short last batch, then the epoch reshuffle
    assert b3.shape[1] == 90 - 64 and bank.i_batch == 0
    torch.manual_seed(3)
    perm = torch.randperm(90)
    assert np.array_equal(bank.rays_rgb.numpy(), ref[perm.numpy()])


def test_c_program_is_built_and_links_the_library():
    """nerf-loc_amd/lib/render_c (tools/c_abi/render_c.cpp) is built with the library; without arguments it prints its
    usage and touches no GPU."""
    binary = os.path.join(ROOT, "nerf-loc_amd", "lib", "render_c")
    assert os.path.exists(binary), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    p = subprocess.run([binary], capture_output=True, text=True, timeout=60)
    assert p.returncode == 1 and "usage: render_c" in p.stderr


def test_lists_of_networks_are_checked_before_any_device_work():
    """One network per object: the whole ray batch is shared out in equal consecutive parts.  What would silently
    render rays with another object's weights is refused: a single fine network beside a list of coarse ones, a chunk
    that cuts the batch, pose lists of unequal length."""
    import torch
    from nerf_loc_amd import nerf_utils, render
    rays = torch.zeros(8, 11)
    q = render.NetworkQuery(None, None, 65536)
    with pytest.raises(ValueError, match="network_fine must be None or a list"):
        render.render_rays(rays, [object(), object()], q, 4, N_importance=4, network_fine=object())
    with pytest.raises(ValueError, match="network_fine must be None or a list"):
        render.render_rays(rays, [object(), object()], q, 4, N_importance=4, network_fine=[object()])
    with pytest.raises(ValueError, match="needs a list of coarse networks"):
        render.render_rays(rays, object(), q, 4, N_importance=4, network_fine=[object(), object()])
    with pytest.raises(ValueError, match="shares out the WHOLE ray batch"):
        render.batchify_rays(rays, 4, network_fn=[object(), object()], network_query_fn=q, N_samples=4)
    kws = [dict(network_fn=object(), near=0.1, far=9.0) for _ in range(3)]
    eye = torch.eye(4)[:3]
    with pytest.raises(ValueError, match="pose lists of equal length"):
        nerf_utils.render_path_device_objects([[eye, eye], [eye], [eye, eye, eye]], [4, 4, 3.0], None, kws)
    with pytest.raises(ValueError, match="pose lists of equal length"):
        nerf_utils.render_path_device_objects([[eye], [eye]], [4, 4, 3.0], None, kws)
