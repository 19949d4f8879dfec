#!/usr/bin/env python3
"""Throughput of the NeRF volume-rendering path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--precision bf16|fp32] [--res 800]

One step = one full render of the reference's headline workload on every rank: an 800 x 800 view
(640 000 rays), 64 coarse + 128 importance samples, D=8/W=256 networks with view directions, separate fine
network, random-init weights ("trained-like" scaling so the scene is opaque), white background.  Rays are
generated on the device from the pose; inputs and weights are resident in HBM before the timed region.  With
N > 1 every rank renders its own 640 000-ray shard of an N x 640 000-ray batch (weak scaling) and one RCCL
all-gather assembles the per-ray maps on every rank inside the timed region.

Rank 0 prints ONE JSON line: rays/s over all ranks, the MFMA roofline of the network kernel (HIP events
around every launch of it in the timed region) and the CPU oracle timed on a bounded sample.  The same line also
carries, measured after the headline region in the same process: "fp32" (the same view in the exact-float32 MFMA
mode, the precision the reference computes in) and "strong_4m" (BASELINE configs[2]: 4 194 304 rays split over
the ranks, one all-gather).

Launching: under torchrun (WORLD_SIZE set) every process is one rank and --gpus must equal WORLD_SIZE.  Without
it, `bench.py --gpus N` with N > 1 starts N rank processes itself (the parent never touches a GPU), relays rank
0's line and exits non-zero if any rank fails.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOP_PER_SAMPLE = 1186816          # 2 x 593 408 MAC (run_nerf_helpers.py:67-119 shapes), bias/activation/PE not counted
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3, "bf16x3": 2500.0, "bf16x2": 2500.0}   # dense MFMA peaks of MI355X
# bf16 products the split-operand modes issue per algorithmic multiply-add (operands as sums of 3 / 2 bf16 terms)
PRODUCTS = {"bf16": 1, "fp32": 1, "bf16x3": 6, "bf16x2": 3}
DTYPE = {"bf16": "bf16", "fp32": "f32", "bf16x3": "bf16x3", "bf16x2": "bf16x2"}
KERNEL = {"bf16": "mlp16_bf16_kernel", "fp32": "mlp_f32_kernel", "bf16x3": "mlp_split_kernel<3>", "bf16x2": "mlp_split_kernel<2>"}


def build_scene(device, precision, res, regime="B"):
    from nerf_loc_amd import synth, helpers, render
    H, W, K, c2w, near, far = synth.lego_camera(res, theta=30.0)

    def net(seed):
        n = helpers.NeRF(D=8, W=256, input_ch=63, output_ch=4, skips=[4], input_ch_views=27, use_viewdirs=True)
        n.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(seed, regime).items()})
        n = n.to(device).requires_grad_(False)
        n.precision = precision
        return n

    q = render.NetworkQuery(helpers.get_embedder(10, 0)[0], helpers.get_embedder(4, 0)[0], 1024 * 64)
    kwargs = dict(network_query_fn=q, perturb=0.0, N_importance=128, network_fine=net(1), N_samples=64,
                  network_fn=net(0), use_viewdirs=True, white_bkgd=True, raw_noise_std=0.0, ndc=False, lindisp=False,
                  near=near, far=far)
    return H, W, K, torch.from_numpy(c2w).to(device), kwargs


def cpu_baseline(full=True):
    """The numpy oracle (a port of the reference's CPU path, pinned to the reference by golden fixtures) on the host
    cores, on two bounded samples: a 32 768-ray batch of the headline view with 64 + 128 samples (`value`), and
    BASELINE configs[0] exactly: the whole 400 x 400 view, 160 000 rays, coarse pass only, random-init weights.
    full=False (quick runs) takes an eighth of each."""
    from nerf_loc_amd import synth
    from oracle import nerf_oracle as orc
    H, W, K, c2w, near, far = synth.lego_camera(800, theta=30.0)
    o, d = orc.get_rays(H, W, K, c2w)
    rays = orc.pack_rays(o, d, near, far, True)
    sc, sf = synth.make_state_dict(0, "B"), synth.make_state_dict(1, "B")
    n2 = 32768 if full else 4096
    first = (H // 2) * W                      # rows through the middle of the view: rays that hit the scene
    t0 = time.time()
    for i in range(0, n2, 8192):              # 8192-ray chunks bound the oracle's memory (the reference chunks at 32 768)
        orc.render_rays(rays[first + i:first + i + 8192], sc, 64, 128, sd_fine=sf, white_bkgd=True)
    dt = time.time() - t0
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        cores = os.cpu_count() or 1
    H0, W0, K0, c2w0, near0, far0 = synth.lego_camera(400, theta=30.0)
    o0, d0 = orc.get_rays(H0, W0, K0, c2w0)
    rays0 = orc.pack_rays(o0, d0, near0, far0, True)
    n3 = rays0.shape[0] if full else rays0.shape[0] // 8
    sa = synth.make_state_dict(0, "A")
    t0 = time.time()
    for i in range(0, n3, 32768):
        orc.render_rays(rays0[i:i + 32768], sa, 64, 0, white_bkgd=True)
    dt3 = time.time() - t0
    return {"value": n2 / dt, "unit": "rays/s", "cores": int(cores), "kind": "port", "cpu": cpu_model(),
            "seconds": dt + dt3,
            "sample": "%d rays from the middle rows of the same 800x800 view, 64+128 samples, numpy oracle (float32, %d BLAS "
                      "threads)" % (n2, cores),
            "coarse_only": {"value": n3 / dt3, "unit": "rays/s", "seconds": dt3,
                            "sample": "%s rays of BASELINE configs[0] (400x400 view, 64 coarse samples, no importance "
                                      "sampling, random-init weights)" % ("all 160 000" if n3 == 160000 else "the first %d" % n3)}}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def pmc_traffic(precision):
    """HBM-side bytes of one fine-pass launch of the network kernel, from the committed rocprofv3 PMC passes of
    this same command (tools/profile_bench.sh; counters cannot be read from inside the process).  None if the
    summary for this precision has not been collected."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic_%s.json" % precision)))
    try:
        with open(paths[-1]) as f:
            return float(json.load(f)["bytes_per_fine_launch"])
    except (IndexError, OSError, KeyError, ValueError):
        return None


def build_objects(device, precision, n_objects):
    """Per-object networks rendering the detector's inputs.  Each object: 8 poses x 2
    focal lengths (fine K and K/1.2) at 240 x 180, 64 samples, coarse network only, raw volume kept
    (model_3detr.py:619-724 through nerf_utils.render_path)."""
    import types
    from nerf_loc_amd import nerf_utils, synth
    args = types.SimpleNamespace(nerf_render_rgb=True, nerf_render_depth=True)
    objs = []
    for o in range(n_objects):
        _, kw, _ = nerf_utils.create_nerf_model(args, ckpt_path=None, device=device, near=0.1, far=9.0)
        net = kw["network_fn"]
        net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(100 + o, "B").items()})
        net.requires_grad_(False)
        net.precision = precision
        objs.append(kw)
    H, W, K, _, _, _ = synth.objectron_camera()
    Kc = np.array(K, dtype=np.float64).copy()
    Kc[0, 0] /= 1.2
    Kc[1, 1] /= 1.2
    poses = [torch.from_numpy(synth.pose_spherical(45.0 * i, -30.0, 4.0)[:3, :4].astype(np.float32)).to(device)
             for i in range(8)]
    return objs, H, W, K, Kc, poses


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "bf16x3", "bf16x2"])
    ap.add_argument("--train-precision", default="fp32", choices=["fp32", "bf16x3", "bf16"],
                    help="train workload only: float32 (what the reference computes) or bf16 MFMA operands")
    ap.add_argument("--res", type=int, default=800)
    ap.add_argument("--chunk", type=int, default=1 << 20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--quick-cpu-baseline", action="store_true", help="an eighth of the CPU baseline's two samples")
    ap.add_argument("--no-extras", action="store_true",
                    help="view workload: skip the fp32, 4M-ray and coarse-only legs that follow the headline region")
    ap.add_argument("--dry-run", action="store_true",
                    help="rendezvous only: every rank joins the process group, rank 0 prints the world it sees; no "
                         "GPU work (checks the launcher, also on a machine without GPUs)")
    ap.add_argument("--workload", default="view", choices=["view", "view_a", "rays4m", "objects", "train", "coarse400"],
                    help="view: BASELINE configs[1] (default, the headline line); coarse400: configs[0], one 400x400 view with the "
                         "coarse network only; rays4m: configs[2], a 4 194 304-ray "
                         "batch split over the ranks; objects: configs[3], per-object networks dealt to the ranks; "
                         "train: configs[4], 4096-ray training iteration, data parallel")
    return ap.parse_args(argv)


def resolve_world(gpus, environ):
    """('rank', world) when this process is one rank of a job somebody else started (WORLD_SIZE set: --gpus must
    agree with it), ('launch', N) when it has to start N ranks itself, ('rank', 1) for a plain single-GPU run."""
    if gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1, got %d" % gpus)
    ws = environ.get("WORLD_SIZE")
    if ws is not None:
        if int(ws) != gpus:
            raise SystemExit("bench.py: --gpus %d but the launcher set WORLD_SIZE=%s; they must agree" % (gpus, ws))
        return "rank", int(ws)
    return ("launch", gpus) if gpus > 1 else ("rank", 1)


def launch_ranks(n, argv, poll=0.2, grace=5.0):
    """Start n rank processes of this script (one per GPU, rendezvous on 127.0.0.1), relay rank 0's output and
    return non-zero if any rank failed.  All children are watched: when one exits non-zero the others, which would
    otherwise sit in the rendezvous or in a collective until its timeout, are terminated (killed after `grace`
    seconds) and the job fails at once.  Output of ranks other than 0 goes to stderr.  Nothing here initialises a GPU.
    The free port is found by binding and closing a socket: another process can take it before rank 0 binds it; that
    rank then fails, which ends the job through the same path."""
    import socket
    import subprocess
    import threading
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = bad
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(poll)
    if failed:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        deadline = time.time() + grace
        for p in procs:
            try:
                p.wait(timeout=max(0.0, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    reader.join(timeout=grace)
    sys.stdout.write(b"".join(out0).decode("utf-8", "replace"))
    sys.stdout.flush()
    if failed:
        sys.stderr.write("bench.py: ranks failed (rank, exit code): %s; the other ranks were stopped\n" % failed)
        return 1
    return 0


KEYS = ("rgb_map", "disp_map", "acc_map", "rgb0", "disp0", "acc0", "z_std")


def make_workload(name, precision, args, rank, world, device):
    """-> dict(step, n_rays (this rank), total_rays (all ranks, per step), desc, scaling, flop_per_ray, parallelism)."""
    from nerf_loc_amd import dist as ndist, render, synth
    w = {"scaling": "weak", "flop_per_ray": 256 * FLOP_PER_SAMPLE, "precision": precision}
    regime = "A" if name == "view_a" else "B"
    if name == "view_a":
        name = "view"
    if name in ("view", "rays4m"):
        H, W, K, c2w, kwargs = build_scene(device, precision, args.res, regime=regime)
        rk = {k: v for k, v in kwargs.items() if k not in ("near", "far", "use_viewdirs", "ndc")}
        w["parallelism"] = "ray-shard x%d + 1 all-gather" % world
    if name == "view":
        n_rays = H * W
        w.update(n_rays=n_rays, total_rays=n_rays * world,
                 desc=("BASELINE configs[1]: %dx%d view (%d rays/GPU), 64 coarse + 128 importance samples, D8/W256 L10/4 "
                       "viewdirs, separate fine net, seeded random weights (%s), %s-MFMA mode"
                       % (H, W, n_rays, "plain nn.Linear initialisation" if regime == "A" else
                          "density and colour heads scaled up: opaque scene", precision)))

        def step():
            rays_flat, _ = render._build_rays(H, W, K, None, c2w, False, kwargs["near"], kwargs["far"], True, None)
            rec = ndist.pack_record(render.batchify_rays(rays_flat, args.chunk, **rk), KEYS)
            return ndist.all_gather_rows(rec, n_rays * world, n_rays) if world > 1 else rec
    elif name == "coarse400":
        # BASELINE configs[0], the reference's own CPU-runnable case, on the GPU: coarse network only
        H, W, K, c2w, kwargs = build_scene(device, precision, 400, regime="A")
        rk = {k: v for k, v in kwargs.items() if k not in ("near", "far", "use_viewdirs", "ndc")}
        rk.update(N_importance=0, network_fine=None)
        n_rays = H * W
        w.update(n_rays=n_rays, total_rays=n_rays * world, flop_per_ray=64 * FLOP_PER_SAMPLE,
                 parallelism="ray-shard x%d + 1 all-gather" % world,
                 desc=("BASELINE configs[0]: one 400x400 view (%d rays/GPU), 64 coarse samples, no importance sampling, "
                       "D8/W256 L10/4 viewdirs, random-init weights, %s-MFMA mode" % (n_rays, precision)))
        keys0 = tuple(k for k in KEYS if k in ("rgb_map", "disp_map", "acc_map"))

        def step():
            rays_flat, _ = render._build_rays(H, W, K, None, c2w, False, kwargs["near"], kwargs["far"], True, None)
            rec = ndist.pack_record(render.batchify_rays(rays_flat, args.chunk, **rk), keys0)
            return ndist.all_gather_rows(rec, n_rays * world, n_rays) if world > 1 else rec
    elif name == "rays4m":
        total_rays = 4194304
        all_rays = torch.from_numpy(synth.random_rays(total_rays, seed=0)).to(device)
        lo, hi, per = ndist.shard_bounds(total_rays, rank, world)
        w.update(scaling="strong", n_rays=hi - lo, total_rays=total_rays,
                 desc=("BASELINE configs[2]: 4 194 304 synthetic rays split over %d GPU(s), 64 + 128 samples, %s-MFMA "
                       "mode, one all-gather of the per-ray record" % (world, precision)))

        def step():
            rec = ndist.pack_record(render.batchify_rays(all_rays[lo:hi], args.chunk, **rk), KEYS)
            return ndist.all_gather_rows(rec, total_rays, per) if world > 1 else rec
    elif name == "objects":
        from nerf_loc_amd import nerf_utils
        n_obj = 8                           # per rank (weak scaling: 64 objects on 8 GPUs)
        objs, H, W, K, Kc, poses = build_objects(device, precision, n_obj)
        n_rays = n_obj * 2 * 8 * H * W
        w.update(n_rays=n_rays, total_rays=n_rays * world, flop_per_ray=64 * FLOP_PER_SAMPLE,
                 parallelism="objects round-robin x%d, no collective" % world,
                 desc=("BASELINE configs[3]: %d objects/GPU x 2 focal lengths x 8 poses x %dx%d, 64 samples, coarse net, "
                       "rgb + depth + raw volume kept on device, one network launch per field of view, %s-MFMA mode" % (n_obj, H, W, precision)))

        def step():
            out = None
            for Kx in (K, Kc):       # both fields of view; all objects of the rank as ONE network launch each
                rgbs, disps, raws, depths = nerf_utils.render_path_device_objects([poses] * n_obj, [H, W, float(Kx[0][0])],
                                                                                  Kx, objs)
                out = rgbs.reshape(-1, 3)
            return out
    else:
        from nerf_loc_amd import helpers, train
        total = 4096
        H, W, K, c2w, kwargs = build_scene(device, "fp32", 64)
        for k in ("network_fn", "network_fine"):
            kwargs[k].requires_grad_(True)
        rk = {k: v for k, v in kwargs.items() if k not in ("near", "far", "use_viewdirs", "ndc")}
        rk["perturb"] = 1.0
        opt = train.FusedAdam(list(kwargs["network_fn"].parameters()) + list(kwargs["network_fine"].parameters()),
                              lr=5e-4, betas=(0.9, 0.999))
        rays_all = torch.from_numpy(synth.random_rays(total, seed=0)).to(device)
        tgt_all = torch.rand(total, 3, generator=torch.Generator().manual_seed(0)).to(device)
        lo, hi, _ = ndist.shard_bounds(total, rank, world)
        rays_l, tgt_l = rays_all[lo:hi], tgt_all[lo:hi]
        train.set_train_precision(args.train_precision)
        ndist.seed_ranks(0)            # every rank draws its own stratified jitter
        ex = ndist.GradientExchange([kwargs["network_fn"], kwargs["network_fine"]])
        w.update(scaling="strong", n_rays=hi - lo, total_rays=total, flop_per_ray=3 * 256 * FLOP_PER_SAMPLE,
                 precision=args.train_precision, parallelism="data parallel x%d, in-place gradient all-reduce per network, the fine network's under the "
                                                      "coarse network's backward" % world,
                 desc=("BASELINE configs[4]: training iteration, 4096 rays over %d GPU(s), 64 + 128 samples, stratified "
                       "jitter, forward + backward of both networks + gradient all-reduce + Adam, %s"
                       % (world, {"fp32": "float32", "bf16x3": "float32 (forward pass in three bf16 terms on the bf16 matrix cores)",
                                  "bf16": "bf16 operands, float32 accumulation and parameters"}[args.train_precision])))
        it = [0]

        def step():
            ret = render.batchify_rays(rays_l, 1 << 20, **rk)
            ex.begin(n_local=hi - lo, n_global=total)          # in place of optimizer.zero_grad(): one fill
            loss = helpers.img2mse(ret["rgb_map"], tgt_l) + helpers.img2mse(ret["rgb0"], tgt_l)
            loss.backward()                                    # writes into the exchange's buffer; the fine network's
            ex.finish()                                        # all-reduce runs under the coarse network's backward
            opt.step()
            it[0] += 1
            for g in opt.param_groups:
                g["lr"] = 5e-4 * (0.1 ** (it[0] / 500000.0))
            return ret["rgb_map"].detach()
    w["step"] = step
    return w


def timed(step, steps, warmup, world, device):
    """`warmup` untimed steps, then exactly `steps` steps between barrier + synchronize on both sides; the time is
    the maximum over ranks.  Returns (seconds, HIP-event records of the network launches, last output)."""
    import torch.distributed as tdist
    from nerf_loc_amd import hip

    def sync():
        if world > 1:
            tdist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    sync()
    hip.MLP_EVENTS = []
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    sync()
    dt = time.perf_counter() - t0
    events, hip.MLP_EVENTS = hip.MLP_EVENTS, None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device if tdist.get_backend() != "gloo" else "cpu")
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(out[:, :3]).all()
    return dt, events, out


def mlp_roofline(events, precision, dt, n_rays, traffic):
    """MFMA roofline of the network kernel from the HIP events of its launches: the largest launches (the fine pass)
    carry `achieved`; algorithmic FLOPs = samples x 1 186 816."""
    peak = PEAK_TFLOPS[precision]
    kname = KERNEL[precision]
    ms = [e0.elapsed_time(e1) for e0, e1, _ in events]
    flops = [n * FLOP_PER_SAMPLE for _, _, n in events]
    big = [(m, f) for m, f in zip(ms, flops) if f == max(flops)]
    ach = sum(f for _, f in big) / (sum(m for m, _ in big) * 1e-3) / 1e12
    issued = {}
    if PRODUCTS[precision] > 1:      # algorithmic FLOPs stay the numerator; what the matrix cores execute is stated beside it
        issued = {"issued": {"products_per_multiply_add": PRODUCTS[precision], "achieved": ach * PRODUCTS[precision],
                             "frac": ach * PRODUCTS[precision] / peak,
                             "note": "bf16 MFMA FLOPs executed, zero padding not counted; peak = dense bf16 MFMA"}}
    return {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic, **issued,
            "traffic_unit": "HBM-side bytes per largest launch (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE, separate "
                            "PMC passes of the default command)",
            "algorithmic_bytes": big[0][1] // FLOP_PER_SAMPLE * 20 + n_rays * 44,
            "kernel": "%s (largest launches: %d samples each, avg %.3f ms)"
                      % (kname, big[0][1] // FLOP_PER_SAMPLE, sum(m for m, _ in big) / len(big)),
            "all_launches_tflops": sum(flops) / (sum(ms) * 1e-3) / 1e12,
            "mlp_time_share": sum(ms) * 1e-3 / dt}


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    mode, n = resolve_world(args.gpus, os.environ)
    if mode == "launch":
        raise SystemExit(launch_ranks(n, argv))

    if os.environ.get("NERF_BENCH_FAIL_RANK") == os.environ.get("RANK", "0"):     # launcher test: this rank dies at once
        raise SystemExit("bench.py: rank %s told to fail (NERF_BENCH_FAIL_RANK)" % os.environ.get("RANK", "0"))
    from nerf_loc_amd import dist as ndist, hip
    import torch.distributed as tdist
    if args.dry_run:
        rank, world, _ = ndist.init_from_env(None if os.environ.get("NERF_DIST_BACKEND") else "gloo")
        if world > 1:
            t = torch.ones(1)
            tdist.all_reduce(t)
            assert int(t.item()) == world
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": tdist.get_world_size() if world > 1 else 1,
                              "backend": tdist.get_backend() if world > 1 else "none", "gpus_arg": args.gpus}))
        if world > 1:
            tdist.barrier()
            tdist.destroy_process_group()
        return
    rank, world, device = ndist.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU path)")
    hip.lib()
    backend = tdist.get_backend() if world > 1 else "none"
    precision = args.precision
    w = make_workload(args.workload, precision, args, rank, world, device)
    precision = w["precision"]
    dt, events, _ = timed(w["step"], args.steps, args.warmup, world, device)
    n_rays, total_rays = w["n_rays"], w["total_rays"]
    meta = {k: v for k, v in w.items() if k != "step"}

    extras = {}
    if args.workload in ("view", "view_a") and not args.no_extras:
        # the same view in the reference's own precision, and the 4M-ray strong-scaling batch, in this same run
        w = None      # drops the closure and with it the headline workload's device tensors
        torch.cuda.empty_cache()
        for tag, wl, prec, st, wu in (("fp32", "view", "fp32", 2, 1), ("bf16x3", "view", "bf16x3", 2, 1),
                                      ("strong_4m", "rays4m", "bf16", 2, 1), ("coarse_only_400", "coarse400", "bf16", 10, 2),
                                      ("view_random_init", "view_a", "bf16", 3, 1)):
            if tag == precision:
                continue
            x = make_workload(wl, prec, args, rank, world, device)
            dtx, evx, _ = timed(x["step"], st, wu, world, device)
            extras[tag] = {"value": x["total_rays"] * st / dtx, "unit": "rays/s", "ms_per_step": dtx / st * 1e3,
                           "steps": st, "warmup": wu, "dtype": DTYPE[prec],
                           "scaling": x["scaling"], "rays_per_gpu": x["n_rays"], "workload": x["desc"],
                           "roofline": mlp_roofline(evx, prec, dtx, x["n_rays"],
                                                    pmc_traffic(prec) if wl == "view" else None)}
            del x
            torch.cuda.empty_cache()
        if world == 1:
            # BASELINE configs[4] on this one GPU (the whole 4096-ray batch), in the reference's float32 and with forward
            # and input gradients in three bf16 terms; `--workload train` is the data-parallel run
            import copy
            for tp in ("fp32", "bf16x3"):
                a2 = copy.copy(args)
                a2.train_precision = tp
                x = make_workload("train", "fp32", a2, rank, world, device)
                dtx, _, _ = timed(x["step"], 5, 2, world, device)
                extras["train_4096_" + tp] = {"value": x["total_rays"] * 5 / dtx, "unit": "rays/s", "ms_per_step": dtx / 5 * 1e3,
                                              "steps": 5, "warmup": 2, "dtype": DTYPE[tp], "workload": x["desc"]}
                del x
                torch.cuda.empty_cache()
            from nerf_loc_amd import train as _train
            _train.set_train_precision("fp32")

    if rank == 0:
        peak = PEAK_TFLOPS[precision]
        if events:
            roof = mlp_roofline(events, precision, dt, n_rays, pmc_traffic(precision) if args.workload in ("view", "view_a") else None)
        else:   # training: forward-with-tape, input-gradient and weight-gradient kernels, timed as a whole step
            ach = n_rays * meta["flop_per_ray"] * args.steps / dt / 1e12
            if precision == "bf16":
                # tape traffic per sample: forward writes 167 pieces of 32 B (9 of them ReLU bits), the input-gradient
                # chain reads those 9 and writes 156, the weight-gradient launches read 156 (gradients) + 178 (inputs)
                tape_bytes = (167 + 9 + 156 + 156 + 178) * 32
                gbs = n_rays * 192 * tape_bytes * args.steps / dt / 1e9
                roof = {"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0, "traffic": None,
                        "kernel": "whole iteration (mlp_bf16_kernel<TAPE> + mlp_dgrad_bf16_kernel + wgrad16_kernel): "
                                  "%d B of tape traffic per sample / step time; %.0f TFLOP/s at 3 x forward FLOPs" % (tape_bytes, ach)}
            elif precision == "bf16x3":
                # two pipes: forward and input-gradient chain issue six bf16 products per multiply-add (bound: the bf16
                # peak), the weight gradients are exact float32 MFMA (bound: the float32 peak).  frac = the time both
                # would take at their peaks / the measured step.
                fwd, dgrad = FLOP_PER_SAMPLE, 2 * 16 * 32 * 1112      # per sample; 1112 transposed 16 x 32 fragments per term
                ideal = n_rays * 192 * (6 * (fwd + dgrad) / (PEAK_TFLOPS["bf16"] * 1e12) + fwd / (PEAK_TFLOPS["fp32"] * 1e12))
                roof = {"bound": "mfma", "achieved": ach, "peak": None, "unit": "TFLOP/s", "frac": ideal * args.steps / dt,
                        "traffic": None,
                        "kernel": "whole iteration (mlp_split_train_kernel + mlp_split_dgrad_kernel: six bf16 products per "
                                  "multiply-add against %.0f TFLOP/s; wgrad_kernel: float32 MFMA against %.1f TFLOP/s); achieved = "
                                  "3 x forward FLOPs per sample / step time, frac = time at those peaks / step time"
                                  % (PEAK_TFLOPS["bf16"], PEAK_TFLOPS["fp32"])}
            else:
                roof = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": None,
                        "kernel": "whole iteration (mlp_f32_train_kernel + mlp_dgrad_f32_kernel + wgrad_kernel), "
                                  "3 x forward FLOPs per sample / step time"}
        metric = {"view": "rendered rays/sec (64+128 samples)", "view_a": "rendered rays/sec (64+128 samples)", "rays4m": "rendered rays/sec (64+128 samples)",
                  "coarse400": "rendered rays/sec (64 samples, coarse only)",
                  "objects": "rendered rays/sec (64 samples, coarse only)",
                  "train": "training rays/sec (64+128 samples, fwd+bwd+Adam)"}[args.workload]
        line = {
            "metric": metric,
            "value": total_rays * args.steps / dt, "unit": "rays/s",
            "n_gpus": world, "backend": backend, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": meta["scaling"], "vs_baseline": None,
            "dtype": DTYPE[precision], "data": "synthetic",
            "config": {"workload": meta["desc"], "rays_per_gpu": n_rays, "chunk": args.chunk,
                       "parallelism": meta["parallelism"]},
            "roofline": roof,
        }
        line.update(extras)
        # the CPU figure is reported once, by the single-GPU run
        line["cpu_baseline"] = None if (args.no_cpu_baseline or args.workload not in ("view", "view_a") or world > 1) else cpu_baseline(full=not args.quick_cpu_baseline)
        print(json.dumps(line))
    if world > 1:
        tdist.barrier()
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
