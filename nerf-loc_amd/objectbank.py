"""Many per-object networks resident on one GPU, streamed in from nerf-pytorch checkpoints.

NeRF-Loc renders the scene of every sample with that object's own network and reloads the single model whenever the
object changes (model_3detr.py:619-624 -> utils/nerf_utils.py:88-105: torch.load of the 14 MB `.tar`, load_state_dict).
Here the networks STAY on the device: an object costs 2.4 MB of parameters plus 2.5 MB in the kernels' packed layout,
so a 288 GB GPU holds tens of thousands of them, and a batch of objects renders in one network launch
(nerf_utils.render_path_device_objects / detector_inputs_batch take the kwargs list this bank hands out).

Loading is a pipeline: worker threads read and unpickle (`weights_only=True`) the checkpoints and gather the 24
tensors of `network_fn_state_dict` into ONE pinned host buffer each; the calling thread then enqueues one host-to-device
copy and the packing kernel per object on a side stream and makes the render stream wait for the event.  Nothing blocks
the host except a checkpoint that has not been read yet.
"""
import collections
import threading
from concurrent.futures import ThreadPoolExecutor

import torch

from .helpers import NeRF


def state_dict_layout(net):
    """[(name, shape, offset, numel)] of a network's parameters in state_dict order, and the total element count."""
    out, off = [], 0
    for name, p in net.state_dict().items():
        out.append((name, tuple(p.shape), off, p.numel()))
        off += p.numel()
    return out, off


def read_flat(path, layout, total, out=None, key='network_fn_state_dict'):
    """One checkpoint file -> (flat float32 host tensor in `layout` order, global_step).  The file is memory-mapped, so
    only the pages of the tensors that are copied get read (a nerf-pytorch checkpoint is 5/6 fine network and Adam
    moments, which this path never looks at).  `out`: a buffer to fill (e.g. pinned).  Refuses a checkpoint whose
    tensors do not have the network's names and shapes."""
    try:
        ckpt = torch.load(path, map_location='cpu', weights_only=True, mmap=True)
    except (RuntimeError, ValueError):
        ckpt = torch.load(path, map_location='cpu', weights_only=True)      # pre-zipfile format: cannot be mapped
    sd = ckpt[key]
    if set(sd.keys()) != {n for n, _, _, _ in layout}:
        raise ValueError("%s: %s has other tensors than the network (%d against %d)" % (path, key, len(sd), len(layout)))
    flat = out if out is not None else torch.empty(total, dtype=torch.float32)
    for name, shape, off, n in layout:
        t = sd[name]
        if tuple(t.shape) != shape:
            raise ValueError("%s: %s is %s, the network wants %s" % (path, name, tuple(t.shape), shape))
        flat[off:off + n].copy_(t.reshape(-1))
    return flat, int(ckpt.get('global_step', 0))


class _Entry:
    __slots__ = ('net', 'flat', 'event', 'step', 'path', 'dev')


class ObjectBank:
    """bank = ObjectBank(render_kwargs_test)            # the kwargs create_nerf_model returns: the template
    bank.request(key, ckpt_paths)                      # optional: start reading ahead (non-blocking)
    kw = bank.kwargs(key, ckpt_paths)                  # render kwargs with this object's network
    kws = bank.kwargs_list([(key, ckpt_paths), ...])   # a batch, read in parallel

    `ckpt_paths` follows load_nerf_model: a list of files of which the last is used, or one path.  `capacity` bounds
    the number of resident objects (least recently used ones are dropped); None keeps everything.  `workers`: reader
    threads.  torch's restricted unpickler (`weights_only=True`) is Python code that holds the interpreter lock for
    nearly all of a read (about 200 tensor records per nerf-pytorch checkpoint; the bytes themselves are memory-mapped),
    so readers do not run side by side: one reader, which overlaps with the caller's copy and packing launches, is as
    fast as it gets (measured 11.6 ms per 12 MB checkpoint with 1 reader, 23 ms with 4: the threads take turns on the
    lock and pay for the hand-overs).  More than one reader is therefore refused unless `allow_slow_readers` is set."""

    def __init__(self, render_kwargs_test, capacity=None, workers=1, precision=None, allow_slow_readers=False):
        self.template = dict(render_kwargs_test)
        net = self.template['network_fn']
        if not isinstance(net, NeRF):
            raise ValueError("ObjectBank needs the template's network_fn to be a nerf_loc_amd NeRF")
        self.device = next(net.parameters()).device
        if self.device.type != 'cuda':
            raise ValueError("ObjectBank keeps networks on a GPU; the template network is on %s" % self.device)
        self._ctor = dict(D=net.D, W=net.W, input_ch=net.input_ch, input_ch_views=net.input_ch_views,
                          output_ch=net.output_ch, skips=list(net.skips), use_viewdirs=net.use_viewdirs)
        self._precision = precision if precision is not None else net.precision
        self.layout, self.total = state_dict_layout(net)
        self.capacity = capacity
        workers = max(1, int(workers))
        if workers > 1 and not allow_slow_readers:
            workers = 1                                      # see the class docstring: more readers are slower
        self.workers = workers
        self._pool = ThreadPoolExecutor(max_workers=workers, thread_name_prefix="nerf-objectbank")
        self._pending = {}                                   # key -> Future of (flat, step)
        self._resident = collections.OrderedDict()           # key -> _Entry, least recently used first
        self._lock = threading.Lock()
        self._miss_lock = threading.Lock()
        self._stream = torch.cuda.Stream(device=self.device)
        self.stats = collections.Counter()
        # pinned staging buffers, recycled: page-locking a fresh 2.4 MB buffer per object costs more than reading it
        self._staging = []                                   # free buffers
        self._staging_busy = collections.deque()             # (event after the copy, buffer)

    def _staging_get(self):
        with self._lock:
            while self._staging_busy and self._staging_busy[0][0].query():
                self._staging.append(self._staging_busy.popleft()[1])
            if self._staging:
                return self._staging.pop()
        return torch.empty(self.total, dtype=torch.float32, pin_memory=True)

    def _read(self, path):
        buf = self._staging_get()
        try:
            return read_flat(path, self.layout, self.total, out=buf)
        except BaseException:
            with self._lock:
                self._staging.append(buf)                     # a failed read gives its staging buffer back
            raise

    # ---- reading ahead
    @staticmethod
    def _last(ckpt_paths):
        if isinstance(ckpt_paths, (list, tuple)):
            if not ckpt_paths:
                raise ValueError("ObjectBank: empty checkpoint list")
            return ckpt_paths[-1]
        return ckpt_paths

    def request(self, key, ckpt_paths):
        """Starts reading the object's checkpoint on a worker thread unless it is resident or already on its way."""
        path = self._last(ckpt_paths)
        with self._lock:
            e = self._resident.get(key)
            if e is not None and e.path == path:
                return
            fut = self._pending.get(key)
            if fut is not None and fut[0] == path:
                return
            self._pending[key] = (path, self._pool.submit(self._read, path))
            self.stats['reads'] += 1

    # ---- handing out
    def _materialise(self, key, path, flat, step):
        with torch.device('meta'):
            net = NeRF(**self._ctor)                          # names and shapes only: no initialisation kernels
        ev = torch.cuda.Event()
        with torch.cuda.stream(self._stream):
            dev_flat = torch.empty(self.total, dtype=torch.float32, device=self.device)
            dev_flat.copy_(flat, non_blocking=True)
            mods = dict(net.named_modules())
            for name, shape, off, n in self.layout:
                mod_name, _, pname = name.rpartition('.')
                mods[mod_name]._parameters[pname] = torch.nn.Parameter(dev_flat[off:off + n].view(shape), requires_grad=False)
            net.precision = self._precision
            net.packed_weights()                              # packing kernel on the side stream, cached on the network
            ev.record(self._stream)
        with self._lock:
            self._staging_busy.append((ev, flat))             # back to the pool once the copy has run
        e = _Entry()
        e.net, e.flat, e.event, e.step, e.path = net, None, ev, step, path
        e.dev = [dev_flat] + [packed for _, packed in net._pack_cache.values()]
        return e

    def network(self, key, ckpt_paths=None):
        """The object's network, weights on their way to (or on) the GPU; the current stream waits for them."""
        with self._lock:
            e = self._resident.get(key)
            path = self._last(ckpt_paths) if ckpt_paths is not None else (e.path if e is not None else None)
            if e is not None and e.path == path:
                self._resident.move_to_end(key)
                self.stats['hits'] += 1
            else:
                e = None
        if e is None:
            if path is None:
                raise KeyError("ObjectBank: %r is not resident and no checkpoint was named" % (key,))
            with self._miss_lock:                             # one caller at a time brings objects in
                with self._lock:
                    e = self._resident.get(key)               # another caller may have brought it in meanwhile
                    if e is not None and e.path != path:
                        e = None
                if e is None:
                    self.request(key, ckpt_paths)
                    with self._lock:
                        _, fut = self._pending[key]           # stays listed until the object is resident: a request()
                    try:                                      # from another thread meanwhile must not read it again
                        flat, step = fut.result()
                    except BaseException:
                        with self._lock:
                            self._pending.pop(key, None)
                        raise
                    e = self._materialise(key, path, flat, step)
                    with self._lock:
                        self._resident[key] = e
                        self._pending.pop(key, None)
                        self._resident.move_to_end(key)
                        self.stats['loads'] += 1
                        while self.capacity is not None and len(self._resident) > self.capacity:
                            self._resident.popitem(last=False)
                            self.stats['evictions'] += 1
        cur = torch.cuda.current_stream(self.device)
        for t in e.dev:
            t.record_stream(cur)                              # allocated on the side stream, used on this one
        ev = e.event                                          # read once: another caller may clear it meanwhile
        if ev is not None:
            cur.wait_event(ev)
            if ev.query():
                e.event = None                                # landed
        return e.net

    def kwargs(self, key, ckpt_paths=None):
        kw = dict(self.template)
        kw['network_fn'] = self.network(key, ckpt_paths)
        return kw

    def kwargs_list(self, items):
        """items: [(key, ckpt_paths), ...] -> one kwargs dict per item; all missing checkpoints are read in parallel."""
        items = list(items)
        for key, paths in items:
            if paths is not None:
                self.request(key, paths)
        return [self.kwargs(key, paths) for key, paths in items]

    def global_step(self, key):
        return self._resident[key].step

    def __len__(self):
        return len(self._resident)

    def __contains__(self, key):
        return key in self._resident

    def resident_bytes(self):
        per = self.total * 4
        for e in self._resident.values():
            for _, packed in e.net._pack_cache.values():
                per_packed = packed.numel() * packed.element_size()
                return len(self._resident) * (per + per_packed)
        return len(self._resident) * per

    def evict(self, key):
        with self._lock:
            return self._resident.pop(key, None) is not None

    def close(self):
        self._pool.shutdown(wait=True)
