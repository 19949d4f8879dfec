#!/usr/bin/env python3
"""Generates the golden fixtures in this directory by running the reference
implementation (nerf-pytorch inside Jiankai-Sun/nerf-loc) on CPU.

Run only in the build container, where the reference is mounted read-only at
/root/reference:

    cd /tmp && python /root/repo/tests/golden/make_golden.py

Nothing of the reference is copied: it is imported, fed inputs generated from
numpy RandomState seeds (nerf_loc_amd.synth), and its outputs are stored as
.npz data.  Two modules the reference imports at top level but never touches on
this path (imageio, cv2) are not installed here and are registered as empty
placeholders so the import succeeds.

Fixtures (all float32 unless noted):
  g1_sample_pdf.npz   sample_pdf: inputs, samples, searchsorted indices
  g2_raw2outputs.npz  raw2outputs for S=64/192, white background on/off, noise
  g3_embed.npz        positional encodings L=10 and L=4
  g4_mlp.npz          NeRF.forward outputs for seeded weights
  g5_render_rays.npz  render_rays end to end (coarse, coarse+fine, fp64 reruns)
  g6_rays.npz         get_rays / get_rays_np / ndc_rays
  g7_linspace.npz     torch.linspace tables the sampling code depends on
  g8_train_step.npz   loss and gradients of one training step
  g9_render.npz       render() on a small image, both API flavours
  g10_checkpoint.npz  one reference training step (NeRF + torch.optim.Adam) and the checkpoint dict of
                      run_nerf.py:803-811 written and re-read: key names, optimizer-state layout, and the small
                      tensors after the step (the full checkpoint is 14 MB and is not stored)
  g11_poses.npz       load_llff.py pose pipeline (recentre, spiral, spherify) and load_blender.pose_spherical
  g12_view_rays.npz   4096 pixels of the benchmark's 800x800 view with the benchmark's networks, coarse + fine,
                      float32 and float64 reference renders (per-ray maps only)
  g13_loc_image.npz   NeRF-Loc's render() on one 240x180 Objectron view (every third pixel of rgb, disp, acc, depth)
  g15_depths_embed.npz stratified depths of render_rays for 1 ... 300 samples (linear / disparity, with / without the
                      seeded jitter) and positional encodings for 0, 1, 2, 6, 15 frequencies
  g16_render_branches.npz render() with the NDC warp, disparity-linear depths, a static camera, NDC + static camera, and a
                      network without view directions (12 x 16 / 16 x 16 pixels, coarse + fine, float64 mask data,
                      the reference's importance samples)
  g17_train_step_novd.npz loss and gradients of one training step of the architecture without view directions
  g14_shape_sweep.npz sample_pdf for 10 bin / sample counts (2 / 1 ... 200 / 300) and raw2outputs for 20 ray lengths
                      (1 ... 513 samples), a few rays each
"""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

for _m in ("imageio", "cv2"):
    if _m not in sys.modules:
        sys.modules[_m] = types.ModuleType(_m)
sys.path.insert(0, "/root/reference/nerf-pytorch")

import numpy as np
import torch

import run_nerf as ref            # the reference (read-only mount)
import run_nerf_helpers as refh

from nerf_loc_amd import synth

torch.manual_seed(0)


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def n(x):
    return x.detach().cpu().numpy()


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print("%-22s %8.1f KB  %d arrays" % (name, os.path.getsize(path) / 1024.0, len(arrs)))


class Recorder:
    """Wraps a function of a module to record its return values."""

    def __init__(self, mod, name):
        self.mod, self.name, self.orig, self.out = mod, name, getattr(mod, name), []

    def __enter__(self):
        def wrapped(*a, **k):
            r = self.orig(*a, **k)
            self.out.append(r)
            return r
        setattr(self.mod, self.name, wrapped)
        return self

    def __exit__(self, *exc):
        setattr(self.mod, self.name, self.orig)


def make_model(sd, use_viewdirs=True, output_ch=4, dtype=torch.float32):
    m = refh.NeRF(D=8, W=256, input_ch=63, output_ch=output_ch, skips=[4],
                  input_ch_views=27 if use_viewdirs else 0, use_viewdirs=use_viewdirs)
    m.load_state_dict({k: t(v) for k, v in sd.items()})
    return m.to(dtype)


def query_fn(use_viewdirs=True):
    embed_fn, _ = refh.get_embedder(10, 0)
    embeddirs_fn = refh.get_embedder(4, 0)[0] if use_viewdirs else None
    return lambda inputs, viewdirs, network_fn: ref.run_network(
        inputs, viewdirs, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn, netchunk=1024 * 64)


# ---------------------------------------------------------------- g1
def pdf_inputs(rs, R, nb):
    """bins [R,nb] sorted mid-points and weights [R,nb-1] of mixed character."""
    z = np.sort(rs.uniform(2.0, 6.0, size=(R, nb + 1)), -1).astype(np.float32)
    bins = (0.5 * (z[:, 1:] + z[:, :-1])).astype(np.float32)
    w = np.empty((R, nb - 1), np.float32)
    q = R // 8
    w[:2 * q] = rs.uniform(0, 1, size=(2 * q, nb - 1)) * rs.uniform(0.001, 0.2, size=(2 * q, 1))
    c = rs.uniform(0, nb - 1, size=(2 * q, 1))
    s = rs.uniform(0.5, 4.0, size=(2 * q, 1))
    w[2 * q:4 * q] = np.exp(-0.5 * ((np.arange(nb - 1)[None] - c) / s) ** 2) * rs.uniform(0.2, 1.0, size=(2 * q, 1))
    w[4 * q:5 * q] = 0.0
    w[5 * q:6 * q] = 0.0
    w[np.arange(5 * q, 6 * q), rs.randint(0, nb - 1, size=q)] = rs.uniform(0.1, 1.0, size=q)
    w[6 * q:] = 10.0 ** rs.uniform(-7, -2, size=(R - 6 * q, nb - 1))
    return bins, w.astype(np.float32)


def g1():
    rs = np.random.RandomState(101)
    out = {}
    for tag, R, nb, ns in (("a", 256, 63, 128), ("b", 64, 63, 64), ("c", 64, 31, 40)):
        bins, w = pdf_inputs(rs, R, nb)
        out[tag + "_bins"], out[tag + "_weights"] = bins, w
        for mode, kw in (("det", dict(det=True)), ("rnd", dict(det=False, pytest=True)),
                         ("detpy", dict(det=True, pytest=True))):
            with Recorder(torch, "searchsorted") as rec:
                s = refh.sample_pdf(t(bins), t(w), ns, **kw)
            out["%s_%s_samples" % (tag, mode)] = n(s)
            out["%s_%s_inds" % (tag, mode)] = n(rec.out[0]).astype(np.int16)
        out[tag + "_nsamples"] = np.int64(ns)
    save("g1_sample_pdf.npz", **out)


# ---------------------------------------------------------------- g2
def g2():
    rs = np.random.RandomState(202)
    out = {}
    for S in (64, 192, 5):
        R = 128
        raw = rs.normal(size=(R, S, 4)).astype(np.float32)
        raw[64:] *= np.float32(30.0)
        raw[120:, :, 3] = -np.abs(raw[120:, :, 3])      # no density at all: acc = 0, disp = NaN
        raw[112:120, :, 3] = np.abs(raw[112:120, :, 3]) + 5  # opaque at once
        z = np.sort(rs.uniform(2.0, 6.0, size=(R, S)), -1).astype(np.float32)
        d = (rs.normal(size=(R, 3)) * rs.uniform(0.5, 2.0, size=(R, 1))).astype(np.float32)
        out["s%d_raw" % S], out["s%d_z" % S], out["s%d_d" % S] = raw, z, d
        for tag, kw in (("k", dict(white_bkgd=False)), ("w", dict(white_bkgd=True)),
                        ("n", dict(white_bkgd=False, raw_noise_std=1.0, pytest=True))):
            r = ref.raw2outputs(t(raw), t(z), t(d), **kw)
            for name, v in zip(("rgb", "disp", "acc", "weights", "depth"), r):
                out["s%d_%s_%s" % (S, tag, name)] = n(v)
    save("g2_raw2outputs.npz", **out)


# ---------------------------------------------------------------- g3
def g3():
    rs = np.random.RandomState(303)
    x = rs.uniform(-6, 6, size=(256, 3)).astype(np.float32)
    x[:8] = rs.uniform(-1e-3, 1e-3, size=(8, 3))
    d = rs.normal(size=(256, 3))
    d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
    e10, c10 = refh.get_embedder(10, 0)
    e4, c4 = refh.get_embedder(4, 0)
    assert (c10, c4) == (63, 27)
    save("g3_embed.npz", x=x, d=d, embed10=n(e10(t(x))), embed4=n(e4(t(d))))


# ---------------------------------------------------------------- g4
def mlp_inputs(rs, N):
    x = rs.uniform(-4, 4, size=(N, 3)).astype(np.float32)
    d = rs.normal(size=(N, 3))
    d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
    e10 = refh.get_embedder(10, 0)[0]
    e4 = refh.get_embedder(4, 0)[0]
    return x, d, n(torch.cat([e10(t(x)), e4(t(d))], -1))


def g4():
    rs = np.random.RandomState(404)
    x, d, feat = mlp_inputs(rs, 512)
    out = dict(x=x, d=d, feat=feat)
    with torch.no_grad():
        for regime in "AB":
            m = make_model(synth.make_state_dict(40, regime))
            out["out_vd_" + regime] = n(m(t(feat)))
        m5 = make_model(synth.make_state_dict(41, "A", use_viewdirs=False, output_ch=5, input_ch_views=0), False, 5)
        out["out_novd5_A"] = n(m5(t(feat[:, :63])))
        m4 = make_model(synth.make_state_dict(42, "B", use_viewdirs=False, output_ch=4, input_ch_views=0), False, 4)
        out["out_novd4_B"] = n(m4(t(feat[:, :63])))
    save("g4_mlp.npz", **out)


# ---------------------------------------------------------------- g5
def run_render_rays(rays, sd_c, sd_f, dtype=torch.float32, use_viewdirs=True, output_ch=4, **kw):
    torch.set_default_dtype(dtype)
    try:
        net_c = make_model(sd_c, use_viewdirs, output_ch, dtype)
        net_f = make_model(sd_f, use_viewdirs, output_ch, dtype) if sd_f is not None else None
        with torch.no_grad(), Recorder(ref, "sample_pdf") as rec:
            r = ref.render_rays(t(rays).to(dtype), net_c, query_fn(use_viewdirs), network_fine=net_f, **kw)
        r = {k: n(v) for k, v in r.items()}
        if rec.out:
            r["z_samples"] = n(rec.out[0])
        return r
    finally:
        torch.set_default_dtype(torch.float32)


def g5():
    rays = synth.random_rays(256, seed=5)
    out = dict(rays=rays)

    def put(tag, r, keep_raw=0):
        for k, v in r.items():
            if k == "raw":
                if keep_raw:
                    out["%s_raw" % tag] = v[:keep_raw]
                continue
            out["%s_%s" % (tag, k)] = v

    for regime in "AB":
        sc, sf = synth.make_state_dict(50, regime), synth.make_state_dict(51, regime)
        kw = dict(N_samples=64, white_bkgd=True, perturb=0.0, raw_noise_std=0.0, retraw=True)
        put("c64_" + regime, run_render_rays(rays, sc, None, N_importance=0, **kw), keep_raw=32)
        put("cf_" + regime, run_render_rays(rays, sc, sf, N_importance=128, **kw), keep_raw=8)
        r64 = run_render_rays(rays, sc, sf, dtype=torch.float64, N_importance=128, **kw)
        put("cf64_" + regime, {k: v for k, v in r64.items() if k != "raw"})
        out["cf64_%s_rawlast" % regime] = r64["raw"][:, -1, 3]
        r64 = run_render_rays(rays, sc, None, dtype=torch.float64, N_importance=0, **kw)
        put("c6464_" + regime, {k: v for k, v in r64.items() if k != "raw"})
        out["c6464_%s_rawlast" % regime] = r64["raw"][:, -1, 3]
    sc, sf = synth.make_state_dict(50, "B"), synth.make_state_dict(51, "B")
    # inverse-depth sampling, black background, shared coarse net for the fine pass
    put("lindisp_B", run_render_rays(rays[:64], sc, None, N_samples=64, N_importance=64, lindisp=True,
                                     white_bkgd=False, perturb=0.0, retraw=False))
    # no view directions: 8-column rays, 5-channel head (run_nerf.py:193)
    s5c = synth.make_state_dict(52, "B", use_viewdirs=False, output_ch=5, input_ch_views=0)
    s5f = synth.make_state_dict(53, "B", use_viewdirs=False, output_ch=5, input_ch_views=0)
    put("novd_B", run_render_rays(rays[:64, :8], s5c, s5f, use_viewdirs=False, output_ch=5, N_samples=64,
                                  N_importance=128, white_bkgd=True, perturb=0.0, retraw=True), keep_raw=4)
    # stochastic branches through the numpy hooks (pytest=True)
    put("stoch_B", run_render_rays(rays[:64], sc, sf, N_samples=64, N_importance=128, white_bkgd=True,
                                   perturb=1.0, raw_noise_std=1.0, pytest=True, retraw=False))
    # Objectron-style bounds and sample counts (configs/objectron/objectron_1.txt)
    rays_o = synth.random_rays(64, seed=6, near=0.1, far=9.0)
    put("objectron_B", run_render_rays(rays_o, sc, sf, N_samples=64, N_importance=64, white_bkgd=False,
                                       perturb=0.0, retraw=False))
    save("g5_render_rays.npz", **out)


# ---------------------------------------------------------------- g6
def g6():
    out = {}
    cams = {"lego40": synth.lego_camera(40), "lego400": synth.lego_camera(400),
            "obj": synth.objectron_camera(0), "objc": synth.objectron_camera(1, coarse=True)}
    for tag, (H, W, K, c2w, _, _) in cams.items():
        o, d = refh.get_rays(H, W, K, t(c2w))
        o, d = n(o), n(d)
        on, dn = refh.get_rays_np(H, W, K, c2w)
        if H * W > 4096:
            idx = np.arange(0, H * W, 37)
            o, d = o.reshape(-1, 3)[idx], d.reshape(-1, 3)[idx]
            on, dn = np.asarray(on).reshape(-1, 3)[idx], np.asarray(dn).reshape(-1, 3)[idx]
            out[tag + "_idx"] = idx
        out[tag + "_K"], out[tag + "_c2w"], out[tag + "_hw"] = K, c2w, np.array([H, W])
        out[tag + "_o"], out[tag + "_d"] = o, d
        out[tag + "_o_np"], out[tag + "_d_np"] = np.asarray(on, np.float64), np.asarray(dn, np.float64)
    rs = np.random.RandomState(606)
    o = (rs.uniform(-1, 1, size=(256, 3)) + np.array([0, 0, 0.5])).astype(np.float32)
    d = rs.normal(size=(256, 3)).astype(np.float32)
    d[:, 2] = -np.abs(d[:, 2]) - 0.2
    no, nd = refh.ndc_rays(378, 504, 407.5, 1.0, t(o), t(d))
    out.update(ndc_o_in=o, ndc_d_in=d, ndc_o=n(no), ndc_d=n(nd), ndc_hwf=np.array([378, 504, 407.5]))
    save("g6_rays.npz", **out)


# ---------------------------------------------------------------- g7
def g7():
    out = {}
    for k in (2, 3, 5, 31, 32, 40, 63, 64, 65, 127, 128, 192, 256):
        out["lin%d" % k] = n(torch.linspace(0.0, 1.0, steps=k))
        out["nplin%d" % k] = n(torch.Tensor(np.linspace(0.0, 1.0, k)))
    save("g7_linspace.npz", **out)


# ---------------------------------------------------------------- g8
def g8():
    rays = synth.random_rays(128, seed=8)
    rs = np.random.RandomState(808)
    target = rs.uniform(0, 1, size=(128, 3)).astype(np.float32)
    out = dict(rays=rays, target=target)
    for tag, kw in (("det", dict(perturb=0.0, raw_noise_std=0.0)),
                    ("stoch", dict(perturb=1.0, raw_noise_std=1.0, pytest=True))):
        net_c = make_model(synth.make_state_dict(80, "B"))
        net_f = make_model(synth.make_state_dict(81, "B"))
        r = ref.render_rays(t(rays), net_c, query_fn(), N_samples=64, N_importance=128, network_fine=net_f,
                            white_bkgd=True, retraw=True, **kw)
        l1 = refh.img2mse(r["rgb_map"], t(target))
        l0 = refh.img2mse(r["rgb0"], t(target))
        loss = l1 + l0
        loss.backward()
        out[tag + "_loss"] = n(loss)
        out[tag + "_psnr"] = n(refh.mse2psnr(l1))
        out[tag + "_psnr0"] = n(refh.mse2psnr(l0))
        for pre, net in (("c", net_c), ("f", net_f)):
            names, norms = [], []
            for k, p in net.named_parameters():
                names.append(k)
                norms.append(float(p.grad.norm()))
                if k.endswith("bias") or k.startswith(("alpha_linear", "rgb_linear")):
                    out["%s_%s_grad_%s" % (tag, pre, k)] = n(p.grad)
            out["%s_%s_gradnorm" % (tag, pre)] = np.array(norms, np.float64)
            out["%s_names" % pre] = np.array(names)
            out["%s_%s_grad_pts0w" % (tag, pre)] = n(net.pts_linears[0].weight.grad)
            out["%s_%s_grad_pts5w_sub" % (tag, pre)] = n(net.pts_linears[5].weight.grad[::8, ::11])
            out["%s_%s_grad_viewsw_sub" % (tag, pre)] = n(net.views_linears[0].weight.grad[::4, ::9])
    save("g8_train_step.npz", **out)


# ---------------------------------------------------------------- g9
def g9():
    out = {}
    H, W, K, c2w, near, far = synth.lego_camera(20, theta=75.0)
    sc, sf = synth.make_state_dict(90, "B"), synth.make_state_dict(91, "B")
    net_c, net_f = make_model(sc), make_model(sf)
    kw = dict(network_query_fn=query_fn(), perturb=0.0, N_importance=128, network_fine=net_f, N_samples=64,
              network_fn=net_c, use_viewdirs=True, white_bkgd=True, raw_noise_std=0.0, ndc=False,
              lindisp=False, near=near, far=far)
    with torch.no_grad():
        rgb, disp, acc, extras = ref.render(H, W, K, chunk=150, c2w=t(c2w), **kw)
    out.update(lego_K=K, lego_c2w=c2w, lego_hw=np.array([H, W]), lego_rgb=n(rgb), lego_disp=n(disp),
               lego_acc=n(acc), **{"lego_" + k: n(v) for k, v in extras.items()})
    # rays= entry (training style): [2, N, 3]
    rays = synth.random_rays(96, seed=9)
    with torch.no_grad():
        rgb, disp, acc, extras = ref.render(H, W, K, chunk=1024, rays=t(np.stack([rays[:, :3], rays[:, 3:6]])), **kw)
    out.update(batch_rays=rays, batch_rgb=n(rgb), batch_disp=n(disp), batch_acc=n(acc),
               **{"batch_" + k: n(v) for k, v in extras.items()})

    # NeRF-Loc flavour (nerf_loc_coarse_fine/utils/nerf_utils.py): 4 maps + raw, coarse only
    sys.path.insert(0, "/root/reference/nerf_loc_coarse_fine")
    cwd = os.getcwd()
    os.chdir("/root/reference/nerf_loc_coarse_fine")   # it locates nerf-pytorch relative to the cwd
    try:
        from utils import nerf_utils as refloc
    finally:
        os.chdir(cwd)
    args = types.SimpleNamespace(nerf_render_rgb=True, nerf_render_depth=True)
    _, kw_test, _ = refloc.create_nerf_model(args, ckpt_path=None, device="cpu", near=0.1, far=9.0)
    kw_test["network_fn"].load_state_dict({k: t(v) for k, v in sc.items()})
    H, W, K, c2w, near, far = synth.objectron_camera(3)
    H, W = 24, 18
    K = K.copy()
    K[:2] /= 10.0
    with torch.no_grad():
        rgb, disp, acc, depth, extras = refloc.render(H, W, K, chunk=1024 * 32, c2w=t(c2w), **kw_test)
    out.update(loc_K=K, loc_c2w=c2w, loc_hw=np.array([H, W]), loc_rgb=n(rgb), loc_disp=n(disp), loc_acc=n(acc),
               loc_depth=n(depth), loc_raw=n(extras["raw"]))
    save("g9_render.npz", **out)


# ---------------------------------------------------------------- g10
def g10():
    """The reference's optimizer step and checkpoint dict (run_nerf.py:198-239, 770-811) on the g8 batch."""
    import io
    import json
    rays = synth.random_rays(128, seed=8)
    rs = np.random.RandomState(808)
    target = rs.uniform(0, 1, size=(128, 3)).astype(np.float32)
    net_c = make_model(synth.make_state_dict(80, "B"))
    net_f = make_model(synth.make_state_dict(81, "B"))
    grad_vars = list(net_c.parameters()) + list(net_f.parameters())
    optimizer = torch.optim.Adam(params=grad_vars, lr=5e-4, betas=(0.9, 0.999))          # run_nerf.py:213
    out = {}
    global_step = 0
    for it in range(2):
        r = ref.render_rays(t(rays), net_c, query_fn(), N_samples=64, N_importance=128, network_fine=net_f,
                            white_bkgd=True, retraw=True, perturb=0.0, raw_noise_std=0.0)
        optimizer.zero_grad()
        loss = refh.img2mse(r["rgb_map"], t(target)) + refh.img2mse(r["rgb0"], t(target))
        loss.backward()
        optimizer.step()
        global_step += 1
        new_lrate = 5e-4 * (0.1 ** (global_step / (250 * 1000)))                       # run_nerf.py:791-795
        for g in optimizer.param_groups:
            g["lr"] = new_lrate
        out["loss_%d" % it] = n(loss)
    buf = io.BytesIO()
    torch.save({"global_step": global_step, "network_fn_state_dict": net_c.state_dict(),
                "network_fine_state_dict": net_f.state_dict(),
                "optimizer_state_dict": optimizer.state_dict()}, buf)               # run_nerf.py:805-810
    out["ckpt_bytes"] = np.array(buf.tell())
    buf.seek(0)
    ck = torch.load(buf, weights_only=True)
    osd = ck["optimizer_state_dict"]
    layout = {"ckpt_keys": list(ck), "opt_keys": list(osd), "n_state": len(osd["state"]),
              "state_keys": sorted(osd["state"][0]), "step_dtype": str(osd["state"][0]["step"].dtype),
              "step_shape": list(osd["state"][0]["step"].shape), "step": float(osd["state"][0]["step"]),
              "group": {k: (list(v) if isinstance(v, (tuple, list)) else v) for k, v in osd["param_groups"][0].items()},
              "n_groups": len(osd["param_groups"]), "fn_keys": list(ck["network_fn_state_dict"])}
    out["layout_json"] = np.array(json.dumps(layout))
    names = [k for k, _ in net_c.named_parameters()]
    for pre, net, off in (("c", net_c, 0), ("f", net_f, len(names))):
        for i, (k, p) in enumerate(net.named_parameters()):
            if k.endswith("bias") or k.startswith(("alpha_linear", "rgb_linear")):
                st = osd["state"][off + i]
                out["%s_param_%s" % (pre, k)] = n(p)
                out["%s_m_%s" % (pre, k)] = n(st["exp_avg"])
                out["%s_v_%s" % (pre, k)] = n(st["exp_avg_sq"])
        out["%s_param_pts0w" % pre] = n(net.pts_linears[0].weight)
        out["%s_param_pts5w_sub" % pre] = n(net.pts_linears[5].weight[::8, ::11])
    out["rays"], out["target"] = rays, target
    save("g10_checkpoint.npz", **out)


# ---------------------------------------------------------------- g11
def g11():
    """Pose utilities of load_llff.py on a synthetic poses_bounds array: load_llff_data itself with its image reader
    replaced (no image files exist here), spiral / spherify variants, and pose_spherical of load_blender.py."""
    import load_llff
    import load_blender
    rs = np.random.RandomState(11)
    N = 14
    arr = np.zeros((N, 17))
    for i in range(N):
        c2w = synth.pose_spherical(rs.uniform(-40, 40), rs.uniform(-35, -5), rs.uniform(3.5, 4.5))[:3, :4].astype(np.float64)
        c2w[:, 3] += rs.normal(size=3) * 0.05
        # LLFF stores [down, right, back] columns and a hwf column; here simply a consistent synthetic set
        p = np.concatenate([c2w[:, 1:2], c2w[:, 0:1], -c2w[:, 2:3], c2w[:, 3:4], np.array([[1440.], [1920.], [1571.66]])], 1)
        arr[i, :15] = p.reshape(-1)
        arr[i, 15:] = (rs.uniform(1.0, 1.5), rs.uniform(8.0, 12.0))
    out = dict(poses_bounds=arr)
    H, W, pose_factor = 240, 180, 8

    def fake_load(basedir, factor=None, width=None, height=None, load_imgs=True, pose_factor=None):
        poses = arr[:, :-2].reshape([-1, 3, 5]).transpose([1, 2, 0]).copy()
        bds = arr[:, -2:].transpose([1, 0]).copy()
        poses[:2, 4, :] = np.array([H, W]).reshape([2, 1])
        poses[2, 4, :] = poses[2, 4, :] * 1. / pose_factor
        out["parsed_poses"], out["parsed_bds"] = poses.copy(), bds.copy()
        return poses, bds, np.zeros((H, W, 3, N))
    orig = load_llff._load_data
    load_llff._load_data = fake_load
    try:
        for tag, kw in (("spiral", {}), ("sphere", dict(spherify=True)),      # path_zflat: the reference's N_views/2 is a float that numpy's linspace refuses

                        ("norecenter", dict(recenter=False, bd_factor=None))):
            images, poses, bds, render_poses, i_test = load_llff.load_llff_data("none", factor=8, pose_factor=pose_factor, **kw)
            out[tag + "_poses"], out[tag + "_bds"], out[tag + "_render_poses"] = poses, bds, render_poses
            out[tag + "_i_test"] = np.array(i_test)
    finally:
        load_llff._load_data = orig
    out["pose_spherical"] = np.stack([n(load_blender.pose_spherical(a, -30.0, 4.0)) for a in np.linspace(-180, 180, 8 + 1)[:-1]])
    out["hw_pf"] = np.array([H, W, pose_factor])
    save("g11_poses.npz", **out)


# ---------------------------------------------------------------- g12
def g12():
    """The benchmark's own view and networks at statistical size: 4096 pixels drawn from the 800x800 view of
    bench.py (lego camera, theta 30), coarse + fine, rendered by the reference in float32 and float64."""
    H, W, K, c2w, near, far = synth.lego_camera(800, theta=30.0)
    o, d = refh.get_rays(H, W, K, t(c2w))
    rs = np.random.RandomState(12)
    idx = np.sort(rs.choice(H * W, 4096, replace=False))
    o, d = n(o).reshape(-1, 3)[idx], n(d).reshape(-1, 3)[idx]
    vd = d / np.linalg.norm(d, axis=-1, keepdims=True)
    rays = np.concatenate([o, d, np.full((4096, 1), near, np.float32), np.full((4096, 1), far, np.float32), vd], -1).astype(np.float32)
    out = dict(idx=idx, rays=rays)
    sc, sf = synth.make_state_dict(0, "B"), synth.make_state_dict(1, "B")
    kw = dict(N_samples=64, N_importance=128, white_bkgd=True, perturb=0.0, raw_noise_std=0.0, retraw=True)
    r = run_render_rays(rays, sc, sf, **kw)
    for k in ("rgb_map", "disp_map", "acc_map", "rgb0", "disp0", "acc0", "z_std"):
        out["f32_" + k] = r[k]
    r64 = run_render_rays(rays, sc, sf, dtype=torch.float64, **kw)
    for k in ("rgb_map", "acc_map", "rgb0", "acc0"):
        out["f64_" + k] = r64[k].astype(np.float32)
    out["f64_rawlast"] = r64["raw"][:, -1, 3].astype(np.float32)
    save("g12_view_rays.npz", **out)


# ---------------------------------------------------------------- g13
def g13():
    """NeRF-Loc's own render() (nerf_loc_coarse_fine/utils/nerf_utils.py) at its real image size: one 240x180 Objectron
    view, 64 samples, coarse network, near 0.1 / far 9; every third pixel of the four maps is kept."""
    sys.path.insert(0, "/root/reference/nerf_loc_coarse_fine")
    cwd = os.getcwd()
    os.chdir("/root/reference/nerf_loc_coarse_fine")   # it locates nerf-pytorch relative to the cwd
    try:
        from utils import nerf_utils as refloc
    finally:
        os.chdir(cwd)
    args = types.SimpleNamespace(nerf_render_rgb=True, nerf_render_depth=True)
    _, kw_test, _ = refloc.create_nerf_model(args, ckpt_path=None, device="cpu", near=0.1, far=9.0)
    sc = synth.make_state_dict(100, "B")
    kw_test["network_fn"].load_state_dict({k: t(v) for k, v in sc.items()})
    H, W, K, c2w, near, far = synth.objectron_camera(2)
    with torch.no_grad():
        rgb, disp, acc, depth, extras = refloc.render(H, W, K, chunk=1024 * 32, c2w=t(c2w), **kw_test)
    idx = np.arange(0, H * W, 3)
    pick = lambda v: n(v).reshape(H * W, -1)[idx]
    save("g13_loc_image.npz", K=K, c2w=c2w, hw=np.array([H, W]), idx=idx, rgb=pick(rgb), disp=pick(disp)[:, 0],
         acc=pick(acc)[:, 0], depth=pick(depth)[:, 0], raw_sigma_last=n(extras["raw"]).reshape(H * W, 64, 4)[idx, -1, 3])


# ---------------------------------------------------------------- g14
G14_PDF_SHAPES = ((2, 1), (3, 2), (5, 7), (9, 16), (17, 33), (33, 64), (64, 100), (100, 128), (129, 200), (200, 300))
G14_COMPOSITE_S = (1, 2, 3, 4, 7, 16, 31, 33, 63, 65, 100, 127, 129, 191, 193, 255, 256, 257, 300, 513)


def g14():
    """Shape sweep: sample_pdf for bin / sample counts from 2 / 1 to 200 / 300 (ATen's row sum takes different
    vectorised paths by length, and the searchsorted / gather edge cases move with the counts) and raw2outputs for
    1 to 513 samples per ray; a few rays each."""
    rs = np.random.RandomState(1414)
    out = {}
    for nb, ns in G14_PDF_SHAPES:
        R = 16
        bins, w = pdf_inputs(rs, R, nb)
        tag = "p%d_%d" % (nb, ns)
        out[tag + "_bins"], out[tag + "_weights"] = bins, w
        for mode, kw in (("det", dict(det=True)), ("rnd", dict(det=False, pytest=True))):
            with Recorder(torch, "searchsorted") as rec:
                smp = refh.sample_pdf(t(bins), t(w), ns, **kw)
            out["%s_%s_samples" % (tag, mode)] = n(smp)
            out["%s_%s_inds" % (tag, mode)] = n(rec.out[0]).astype(np.int16)
    for S in G14_COMPOSITE_S:
        R = 8
        raw = rs.normal(size=(R, S, 4)).astype(np.float32)
        raw[4:] *= np.float32(20.0)
        raw[7:, :, 3] = -np.abs(raw[7:, :, 3])
        z = np.sort(rs.uniform(0.1, 9.0, size=(R, S)), -1).astype(np.float32)
        d = (rs.normal(size=(R, 3)) * rs.uniform(0.5, 2.0, size=(R, 1))).astype(np.float32)
        out["c%d_raw" % S], out["c%d_z" % S], out["c%d_d" % S] = raw, z, d
        r = ref.raw2outputs(t(raw), t(z), t(d), white_bkgd=(S % 2 == 1))
        for name, v in zip(("rgb", "disp", "acc", "weights", "depth"), r):
            out["c%d_%s" % (S, name)] = n(v)
    save("g14_shape_sweep.npz", **out)


# ---------------------------------------------------------------- g15
G15_Z_SAMPLES = (1, 2, 3, 5, 63, 64, 65, 192, 300)
G15_EMBED_L = (0, 1, 2, 6, 15)


def g15():
    """Stratified depths as render_rays builds them (run_nerf.py:363-385), for 1 to 300 samples, linear in depth and
    in disparity, with and without the numpy-seeded jitter of the pytest hook: the depths are what render_rays hands
    to raw2outputs, recorded there (the network is a stand-in that returns zeros).  And the positional encoding for
    other frequency counts than the two the networks use."""
    rs = np.random.RandomState(1515)
    R = 40
    rays = synth.random_rays(R, seed=15, near=0.1, far=9.0)
    rays[:, 6] += rs.uniform(0, 1, R).astype(np.float32)
    rays[:, 7] -= rs.uniform(0, 2, R).astype(np.float32)
    out = {"rays": rays}
    zero_net = lambda pts, viewdirs, fn: torch.zeros(tuple(pts.shape[:-1]) + (4,))
    orig = ref.raw2outputs
    for S in G15_Z_SAMPLES:
        for lindisp in (False, True):
            for perturb in (0.0, 1.0):
                seen = []

                def spy(raw, z_vals, rays_d, *a, **k):
                    seen.append(n(z_vals).copy())
                    return orig(raw, z_vals, rays_d, *a, **k)
                ref.raw2outputs = spy
                try:
                    with torch.no_grad():
                        ref.render_rays(t(rays), None, zero_net, S, lindisp=lindisp, perturb=perturb, pytest=True)
                finally:
                    ref.raw2outputs = orig
                out["z%d_%s_%s" % (S, "disp" if lindisp else "lin", "jit" if perturb else "det")] = seen[0]
    x = rs.uniform(-6, 6, size=(64, 3)).astype(np.float32)
    out["x"] = x
    for L in G15_EMBED_L:
        fn, ch = refh.get_embedder(L, 0)
        assert ch == 3 + 6 * L
        out["embed%d" % L] = n(fn(t(x)))
    fn, ch = refh.get_embedder(10, -1)            # i = -1: identity, 3 channels
    assert ch == 3
    out["embed_identity"] = n(fn(t(x)))
    save("g15_depths_embed.npz", **out)


# ---------------------------------------------------------------- g16
def g16():
    """render() down its less travelled branches (run_nerf.py:95-123): the NDC warp of forward-facing scenes (near 0,
    far 1), depths linear in disparity, a static camera for the geometry with view directions from the moving
    one, both together, and a network without view directions on 8-column rays.  12 x 16 and 16 x 16 pixels, coarse + fine; beside
    every float32 render the density of the last fine sample from a float64 rerun (conditioning mask of the tests)."""
    H, W, focal = 12, 16, 14.0
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
    c2w = synth.pose_spherical(20.0, -20.0, 1.0)[:3, :4].copy()
    c2w_static = synth.pose_spherical(-35.0, -40.0, 1.3)[:3, :4].copy()
    lego = synth.lego_camera(16, theta=140.0)
    sc, sf = synth.make_state_dict(160, "B"), synth.make_state_dict(161, "B")
    sc5 = synth.make_state_dict(162, "B", use_viewdirs=False, output_ch=5, input_ch_views=0)
    sf5 = synth.make_state_dict(163, "B", use_viewdirs=False, output_ch=5, input_ch_views=0)
    out = dict(K=K, c2w=c2w, c2w_static=c2w_static, hw=np.array([H, W]), lego_K=lego[2], lego_c2w=lego[3],
               lego_hw=np.array([lego[0], lego[1]]))

    def run(tag, Hc, Wc, Kc, pose, vd, nets, **over):
        for dtype, suffix in ((torch.float32, ""), (torch.float64, "64")):
            net_c = make_model(nets[0], use_viewdirs=vd, output_ch=4 if vd else 5, dtype=dtype)
            net_f = make_model(nets[1], use_viewdirs=vd, output_ch=4 if vd else 5, dtype=dtype)
            kw = dict(network_query_fn=query_fn(vd), perturb=0.0, N_importance=128, network_fine=net_f, N_samples=64,
                      network_fn=net_c, use_viewdirs=vd, white_bkgd=True, raw_noise_std=0.0, ndc=False, lindisp=False,
                      near=2.0, far=6.0, retraw=True)
            kw.update(over)
            if "c2w_staticcam" in kw:
                kw["c2w_staticcam"] = t(kw["c2w_staticcam"]).to(dtype)
            if dtype == torch.float64:
                torch.set_default_dtype(torch.float64)
            try:
                with torch.no_grad(), Recorder(ref, "sample_pdf") as rec:
                    rgb, disp, acc, extras = ref.render(Hc, Wc, Kc, chunk=200, c2w=t(pose).to(dtype), **kw)
            finally:
                torch.set_default_dtype(torch.float32)
            if suffix:
                out[tag + "_rawlast64"] = n(extras["raw"])[..., -1, 3].astype(np.float32)
                out[tag + "_rgb64"], out[tag + "_acc64"] = n(rgb).astype(np.float32), n(acc).astype(np.float32)
            else:
                out[tag + "_z_samples"] = np.concatenate([n(z) for z in rec.out], 0)      # one entry per chunk
                out.update({tag + "_rgb": n(rgb), tag + "_disp": n(disp), tag + "_acc": n(acc),
                            tag + "_rgb0": n(extras["rgb0"]), tag + "_acc0": n(extras["acc0"]),
                            tag + "_z_std": n(extras["z_std"])})

    run("ndc", H, W, K, c2w, True, (sc, sf), ndc=True, near=0.0, far=1.0)
    run("lindisp", H, W, K, c2w, True, (sc, sf), near=0.5, far=3.0, lindisp=True)
    run("static", lego[0], lego[1], lego[2], lego[3], True, (sc, sf), c2w_staticcam=c2w_static)
    run("ndc_static", H, W, K, c2w, True, (sc, sf), ndc=True, near=0.0, far=1.0, c2w_staticcam=c2w_static)
    run("novd", lego[0], lego[1], lego[2], lego[3], False, (sc5, sf5))
    save("g16_render_branches.npz", **out)


# ---------------------------------------------------------------- g17
def g17():
    """One training step of the architecture without view directions (output_linear with 5 rows, as create_nerf builds
    it when N_importance > 0: run_nerf.py:193) on 8-column rays: loss and gradients from the reference's autograd."""
    rays = synth.random_rays(96, seed=17)[:, :8].copy()
    rs = np.random.RandomState(1717)
    target = rs.uniform(0, 1, size=(96, 3)).astype(np.float32)
    arch = dict(use_viewdirs=False, output_ch=5, input_ch_views=0)
    def step(dtype):
        net_c = make_model(synth.make_state_dict(170, "B", **arch), use_viewdirs=False, output_ch=5, dtype=dtype)
        net_f = make_model(synth.make_state_dict(171, "B", **arch), use_viewdirs=False, output_ch=5, dtype=dtype)
        torch.set_default_dtype(dtype)
        try:
            r = ref.render_rays(t(rays).to(dtype), net_c, query_fn(False), N_samples=64, N_importance=128,
                                network_fine=net_f, white_bkgd=True, retraw=True, perturb=0.0, raw_noise_std=0.0)
            l1 = refh.img2mse(r["rgb_map"], t(target).to(dtype))
            l0 = refh.img2mse(r["rgb0"], t(target).to(dtype))
            loss = l1 + l0
            loss.backward()
        finally:
            torch.set_default_dtype(torch.float32)
        return net_c, net_f, loss, l1

    net_c, net_f, loss, l1 = step(torch.float32)
    out = dict(rays=rays, target=target, loss=n(loss), psnr=n(refh.mse2psnr(l1)))
    # the same step in float64: how far the reference's own float32 gradients are from exact ones
    c64, f64, loss64, _ = step(torch.float64)
    out["loss64"] = n(loss64)
    for pre, net in (("c", c64), ("f", f64)):
        for k, p in net.named_parameters():
            if p.grad is not None and (k.endswith("bias") or k.startswith("output_linear")):
                out["%s_grad64_%s" % (pre, k)] = n(p.grad)
        out[pre + "_grad64_pts0w"] = n(net.pts_linears[0].weight.grad)
        out[pre + "_grad64_pts5w_sub"] = n(net.pts_linears[5].weight.grad[::8, ::11])
    for pre, net in (("c", net_c), ("f", net_f)):
        names, norms = [], []
        for k, p in net.named_parameters():
            if p.grad is None:                      # views_linears: constructed, never used without view directions
                continue
            names.append(k)
            norms.append(float(p.grad.norm()))
            if k.endswith("bias") or k.startswith("output_linear"):
                out["%s_grad_%s" % (pre, k)] = n(p.grad)
        out[pre + "_names"] = np.array(names)
        out[pre + "_gradnorm"] = np.array(norms, np.float64)
        out[pre + "_grad_pts0w"] = n(net.pts_linears[0].weight.grad)
        out[pre + "_grad_pts5w_sub"] = n(net.pts_linears[5].weight.grad[::8, ::11])
    save("g17_train_step_novd.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11", "g12", "g13", "g14", "g15",
                             "g16", "g17"]
    for g in which:
        globals()[g]()
