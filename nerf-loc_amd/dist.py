"""Multi-GPU layer: one process per GPU, torch.distributed over RCCL (backend
"nccl" on ROCm) or gloo for CPU rehearsal.

The reference renders on a single device (run_nerf.py:22, run_multi_nerf.py:572
just skips finished objects); nothing on its render path communicates.  Rays are
independent, so a ray batch is cut into contiguous shards of ceil(N/G) rays, every
rank renders its shard with the single-GPU path, and ONE all-gather of the packed
per-ray record assembles the maps on every rank.  The raw volume stays sharded.
Per-object networks (multi-NeRF) are dealt round-robin with no collective at all.
Data-parallel training (not in the reference, which trains on one device) has the one
exchange step it needs: a single all-reduce of all gradients of both networks.
"""
import os

import torch
import torch.distributed as dist

RECORD_KEYS = ("rgb_map", "disp_map", "acc_map", "depth_map", "rgb0", "disp0", "acc0", "z_std")
_WIDTH = {"rgb_map": 3, "rgb0": 3}


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* and
    bind this process to cuda:LOCAL_RANK.  Returns (rank, world_size, device)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available()
    if use_cuda:
        local %= torch.cuda.device_count()      # rehearsals may run several ranks on one GPU (gloo only)
    backend = backend or os.environ.get("NERF_DIST_BACKEND") or None
    device = torch.device("cuda", local) if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend or ("nccl" if use_cuda else "gloo"), rank=rank, world_size=world)
    return rank, world, device


def world():
    return dist.get_world_size() if dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


def shard_bounds(n, r, g):
    """Contiguous shard [lo, hi) of rank r among g ranks: ceil(n/g) items each,
    the tail ranks may get fewer or none."""
    per = (n + g - 1) // g
    lo = min(n, r * per)
    return lo, min(n, lo + per), per


def objects_of_rank(n_objects, r, g):
    """Round-robin object ids of rank r (independent per-object networks)."""
    return list(range(r, n_objects, g))


def pack_record(ret, keys):
    """[n, width] float32 record of the per-ray maps named in keys."""
    cols = []
    for k in keys:
        v = ret[k]
        cols.append(v.reshape(v.shape[0], -1))
    return torch.cat(cols, -1).contiguous()


def unpack_record(rec, keys):
    out, c = {}, 0
    for k in keys:
        w = _WIDTH.get(k, 1)
        v = rec[:, c:c + w]
        out[k] = v if w > 1 else v[:, 0]
        c += w
    return out


def all_gather_rows(local, n_total, per):
    """One all-gather of equally padded row blocks; returns the first n_total rows."""
    g = world()
    if g == 1:
        return local
    width = local.shape[1]
    if local.shape[0] < per:
        pad = torch.zeros((per - local.shape[0], width), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], 0)
    if local.is_cuda and dist.get_backend() == "gloo":
        # rehearsal on a machine without one GPU per rank: gloo moves host memory
        host = torch.empty((g * per, width), dtype=local.dtype)
        dist.all_gather_into_tensor(host, local.cpu().contiguous())
        return host[:n_total].to(local.device)
    out = torch.empty((g * per, width), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    return out[:n_total]


def render_rays_sharded(rays_flat, render_fn, keys=None):
    """rays_flat [N, 8|11] present on every rank; render_fn(rays) -> dict of per-ray
    tensors (e.g. lambda r: render.batchify_rays(r, chunk, **kwargs)).  Returns the
    dict of gathered maps [N, ...], identical on all ranks."""
    n = rays_flat.shape[0]
    lo, hi, per = shard_bounds(n, rank(), world())
    local = render_fn(rays_flat[lo:hi]) if hi > lo else None
    if keys is None:
        probe = local if local is not None else render_fn(rays_flat[:1])
        keys = [k for k in RECORD_KEYS if k in probe]
    width = sum(_WIDTH.get(k, 1) for k in keys)
    if local is not None:
        rec = pack_record(local, keys)
    else:
        rec = torch.zeros((0, width), dtype=torch.float32, device=rays_flat.device)
    return unpack_record(all_gather_rows(rec, n, per), keys)


def average_gradients(modules, n_local=None, n_global=None):
    """Data-parallel step between backward() and optimizer.step(): ONE all-reduce (sum) over the
    concatenated gradients of all given modules (coarse + fine network: 1 191 688 floats, 4.8 MB).
    Every rank computes its loss as a mean over ITS n_local rays; the gradient of the mean over the global batch
    of n_global rays is sum_r (n_local_r / n_global) grad_r, so each rank scales by its own weight before the sum.
    Without the counts the shards are taken to be equal (weight 1 / world size).  The buffer covers every
    parameter that requires grad, in module order, with zeros where a rank has no gradient (an empty shard), so
    all ranks reduce the same length; afterwards every such parameter carries the global gradient, and identical
    Adam states stay identical on every rank."""
    g = world()
    params = [p for m in modules for p in m.parameters() if p.requires_grad]
    if g == 1 or not params:
        return 0
    if (n_local is None) != (n_global is None):
        raise ValueError("average_gradients: pass both n_local and n_global, or neither")
    weight = 1.0 / g if n_local is None else float(n_local) / float(n_global)
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    if weight != 1.0:
        flat = flat * weight
    if flat.is_cuda and dist.get_backend() == "gloo":
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        flat = host.to(flat.device)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    o = 0
    for p in params:
        n = p.numel()
        if p.grad is None:
            p.grad = flat[o:o + n].view_as(p).clone()
        else:
            p.grad.copy_(flat[o:o + n].view_as(p.grad))
        o += n
    return o


class _ArenaSlot:
    """What a module knows about its share of a GradientExchange: its gradient views (parameter order) and whether
    they still wait for this step's gradient."""

    def __init__(self, exchange, index, params, views):
        self.exchange, self.index, self.params, self.views = exchange, index, params, views
        self.fresh = True

    def view_of(self, p):
        for q, v in zip(self.params, self.views):
            if q is p:
                return v
        return None

    def attach(self):
        for p, v in zip(self.params, self.views):
            if p.grad is not v:
                p.grad = v

    def written(self):
        """The module's gradient of this step is complete in its views (written in place by the backward kernels)."""
        self.fresh = False
        self.exchange.module_ready(self.index)


class GradientExchange:
    """Data-parallel gradient exchange without copies, overlapped with the backward pass.

    All gradients of the given modules (coarse + fine network: 1 191 688 floats) live in ONE flat float32 buffer;
    every parameter's .grad is a view of it, so the all-reduce runs in place: no concatenation before, no copies
    after.  The buffer is cut per module: as soon as a module's gradient is complete (the HIP backward writes it
    straight into the views and reports it; loss.backward() runs the fine network's backward first, run_nerf.py:785)
    its slice is scaled by this rank's weight n_local / n_global and all-reduced asynchronously, while the other
    network's backward is still running on the compute stream; finish() exchanges whatever is left and waits.
    Per step:  ex.begin(n_local, n_global); loss.backward(); ex.finish(); optimizer.step().
    A world of one exchanges nothing and still saves the per-step gradient allocations."""

    def __init__(self, modules):
        self.modules = list(modules)
        params = [[p for p in m.parameters() if p.requires_grad] for m in self.modules]
        first = next((p for ps in params for p in ps), None)
        if first is None:
            raise ValueError("GradientExchange: no parameter requires grad")
        total = sum(p.numel() for ps in params for p in ps)
        self.flat = torch.zeros(total, dtype=torch.float32, device=first.device)
        self.bounds, self.slots = [], []
        o = 0
        for i, (m, ps) in enumerate(zip(self.modules, params)):
            start, views = o, []
            for p in ps:
                if p.dtype != torch.float32 or p.device != first.device:
                    raise ValueError("GradientExchange: parameters must be float32 on one device")
                views.append(self.flat[o:o + p.numel()].view_as(p))
                o += p.numel()
            self.bounds.append((start, o))
            slot = _ArenaSlot(self, i, ps, views)
            slot.attach()
            m._grad_arena = slot
            self.slots.append(slot)
        self.weight = 1.0
        self._pending = []
        self._sent = [False] * len(self.modules)

    def begin(self, n_local=None, n_global=None):
        """Start of a step (in place of optimizer.zero_grad()): clears the buffer with one fill and fixes this rank's
        weight, n_local / n_global rays of the global batch (equal shards when omitted)."""
        if (n_local is None) != (n_global is None):
            raise ValueError("GradientExchange.begin: pass both n_local and n_global, or neither")
        g = world()
        self.weight = (1.0 / g if n_local is None else float(n_local) / float(n_global)) if g > 1 else 1.0
        self.flat.zero_()
        self._pending = []
        self._sent = [False] * len(self.modules)
        for s in self.slots:
            s.attach()
            s.fresh = True

    zero_grad = begin

    def module_ready(self, index):
        """Module `index` has its complete local gradient in its slice: exchange it now, asynchronously."""
        if world() == 1 or self._sent[index]:
            return
        self._sent[index] = True
        lo, hi = self.bounds[index]
        part = self.flat[lo:hi]
        if self.weight != 1.0:
            part.mul_(self.weight)
        if part.is_cuda and dist.get_backend() == "gloo":     # CPU rehearsal of a GPU job: through host memory
            host = part.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            part.copy_(host)
            return
        # RCCL: the collective is ordered after everything enqueued on the current stream so far and runs on the
        # communicator's own stream; the compute stream goes on with the other network's backward
        self._pending.append(dist.all_reduce(part, op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        """Exchanges the slices nobody reported, then makes the current stream wait for all of them.  Returns the
        number of gradient elements this rank holds (all of them global sums afterwards)."""
        for i in range(len(self.modules)):
            self.module_ready(i)
        for w in self._pending:
            w.wait()
        self._pending = []
        return self.flat.numel()


def seed_ranks(seed):
    """Seeds torch's generators (CPU and every GPU) with seed + rank: the stratified jitter and the density noise of
    a data-parallel step (torch.rand / torch.randn in render_rays, run_nerf.py:377, 291) must differ between the
    shards of one batch; a world of one keeps `seed`."""
    torch.manual_seed(int(seed) + rank())
    return int(seed) + rank()


def shard_batch(rays, target):
    """This rank's contiguous share of a training batch (rays [N, C], target [N, 3])."""
    lo, hi, _ = shard_bounds(rays.shape[0], rank(), world())
    return rays[lo:hi], target[lo:hi]
