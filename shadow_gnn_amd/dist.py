"""Data-parallel training of the hot path: one process per GPU, minibatches of
disjoint subgraphs sharded over the ranks (shadow_gnn_amd.minibatch.plan_epoch
cuts the shared root permutation), and the gradient exchange over one flat fp32
buffer (RCCL over xGMI through torch.distributed backend "nccl"; "gloo" on CPU
for the tests).  The reference has no multi-GPU code at all (SURVEY.md section
8(e)); this is new design, not a translation.

Exchange contract: every rank scales its loss by its share B_r / B of the
step's global batch (``OneBatchSubgraph.loss_weight``) and the buffers are
SUMMED -- one collective kernel per bucket, no division pass -- which gives the
gradient of the mean loss over the global batch also when the shares differ or
a rank's share is empty (it contributes zeros, but still enters the collective)."""
import os
from typing import Iterable, List, Optional

import numpy as np
import torch
import torch.distributed as dist


def force_init() -> bool:
    """SHADOW_DIST_FORCE_INIT=1: build the process group and issue every collective even with ONE rank -- the RCCL path
    (communicator set-up, GPU-side broadcast, the async bucket all-reduces behind the backward hooks) then runs on a
    single GPU exactly as it does on eight (tests/test_dist_rccl_gpu.py, `bench.py --gpus 1` under torch.distributed.run)."""
    return os.environ.get("SHADOW_DIST_FORCE_INIT", "0") == "1"


def collectives_on() -> bool:
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force_init())


def init_from_env(backend: Optional[str] = None):
    """torchrun-style environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).
    Returns (rank, local_rank, world_size); a no-op for single-process runs (unless SHADOW_DIST_FORCE_INIT=1)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force_init()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # SHADOW_DIST_BACKEND=gloo lets several ranks share one GPU (functional tests only)
            backend = os.environ.get("SHADOW_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            dev = torch.device("cuda", local % max(1, torch.cuda.device_count()))
            torch.cuda.set_device(dev)
            kw["device_id"] = dev            # (eager communicator set-up on this rank's GPU; barrier() needs no device guess)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local, world


def pin_host_threads(local_rank: int, local_world: int, threads_per_rank: int = 0) -> dict:
    """One Python process per GPU on a shared host: give every rank its own contiguous slice of the host's hardware
    threads (sched_setaffinity) and cap torch's intra-op pool to it.  Without it N ranks each start a pool of ALL cores
    and their ~2 ms of launch work per step (the autograd walk, ctypes entries) migrate across the sockets and contend --
    the weak-scaling risk SURVEY 8(e) names.  Single-rank runs are left alone (bench.py's CPU baselines want the cores)."""
    info = dict(pinned=False)
    if local_world <= 1:
        return info
    try:
        avail = sorted(os.sched_getaffinity(0))
    except AttributeError:                              # (not Linux)
        return info
    per = max(1, len(avail) // local_world)
    mine = avail[local_rank * per:(local_rank + 1) * per] or avail
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return info
    nthreads = max(1, min(threads_per_rank or 8, len(mine)))
    torch.set_num_threads(nthreads)
    info.update(pinned=True, cpus=(mine[0], mine[-1]), n_cpus=len(mine), torch_threads=nthreads)
    return info


FLAT_ALIGN = 32          # floats: tensors of the flat gradient / parameter buffers start on 128-byte lines


LAZY_GRAD_PACK = True      # gradients packed into the flat buffer when a slice is complete (GradSync)


class GradSync:
    """All gradients end up in one flat fp32 buffer: the data-parallel exchange is a few SUM all-reduces over contiguous
    slices of it, clip-by-global-norm and Adam (optim.FlatAdam) run over it in two launches.

    Packing: during backward ``param.grad`` is None, so autograd ASSIGNS each gradient (with ``.grad`` pre-set to a view of
    the flat buffer it would launch one tiny add kernel per parameter -- 36 per step for SAGE-5); a slice's gradients are
    copied into the buffer with ONE multi-tensor copy when the slice is complete, and ``param.grad`` becomes the view.
    Parameters that received no gradient leave zeros (the buffer is cleared at the start of the step).

    Overlap with backward: the buffer is cut into ``num_buckets`` slices at parameter boundaries, in REVERSE parameter
    order (the order backward produces gradients in).  A post-accumulate hook per parameter counts a slice down; when it
    is complete it is packed and its all-reduce is issued asynchronously (RCCL runs it on its own stream, ordered behind
    the kernels queued so far) while backward keeps producing the earlier layers' gradients.  Slices are always issued in
    slice order, so every rank issues the same sequence of collectives whatever order its hooks fire in -- including a
    rank that had nothing to back-propagate this step."""

    def __init__(self, params: Iterable[torch.nn.Parameter], world_size: Optional[int] = None, group=None,
                 num_buckets: int = 2, overlap: bool = True):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.world_size = world_size if world_size is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.group = group
        sizes = [p.numel() for p in self.params]
        dev = self.params[0].device
        # every tensor starts on a 128-byte line of the flat buffer (the kernels take 16-byte aligned rows: a 47-class
        # classifier's offset / scale vectors would leave the weight matrix behind them on an 8-byte boundary); the pad is zero
        # and stays zero (zero gradient, zero moments)
        self._off = np.zeros(len(sizes) + 1, dtype=np.int64)
        for i, sz in enumerate(sizes):
            self._off[i + 1] = (self._off[i] + sz + FLAT_ALIGN - 1) // FLAT_ALIGN * FLAT_ALIGN
        self._sizes = sizes
        n = int(self._off[-1])
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self._views = []
        for i, p in enumerate(self.params):
            assert p.dtype == torch.float32
            self._views.append(self.flat[self._off[i]:self._off[i] + sizes[i]].view_as(p))
            p.grad = self._views[i]
        # slices (lo_param, hi_param) over the parameter list; slice 0 holds the LAST parameters
        self._slices = []
        nb = max(1, min(int(num_buckets), len(self.params)))
        hi = len(self.params)
        for b in range(nb):
            want = n * (nb - 1 - b) // nb                 # flat offset this slice should reach down to
            lo = hi - 1
            while lo > 0 and self._off[lo] > want:
                lo -= 1
            if b == nb - 1:
                lo = 0
            if lo < hi:
                self._slices.append((lo, hi))
            hi = lo
        self._slice_of = {}
        for s, (lo, hi) in enumerate(self._slices):
            for i in range(lo, hi):
                self._slice_of[i] = s
        self._armed = False
        self._left = [0] * len(self._slices)
        self._next = 0
        self._works = []
        # (the collectives also run on a single rank under SHADOW_DIST_FORCE_INIT=1: the RCCL smoke path)
        self.collective = self.world_size > 1 or (force_init() and dist.is_available() and dist.is_initialized())
        self.overlap = bool(overlap) and self.collective
        self.wait_s = 0.0                # host time spent in Work.wait() (nccl: the enqueue of a stream wait; gloo: the exchange itself)
        self.issued = 0                  # collectives issued over the object's lifetime
        if self.overlap:
            for i, p in enumerate(self.params):
                p.register_post_accumulate_grad_hook(self._make_hook(i))

    def _make_hook(self, i):
        def hook(_param):
            if not self._armed:
                return
            s = self._slice_of[i]
            self._left[s] -= 1
            self._issue_ready()
        return hook

    def _pack(self, s):
        """Gradients of slice s into the flat buffer (one multi-tensor copy); ``.grad`` becomes the buffer's view."""
        lo, hi = self._slices[s]
        from . import ops
        dst, src = [], []
        views, params = self._views, self.params
        for i in range(lo, hi):
            g = params[i].grad
            v = views[i]
            if g is v:
                continue
            params[i].grad = v                                 # (also for a parameter without a gradient this step: zeros)
            if g is None or g.data_ptr() == v.data_ptr():      # (a foreign view of the same storage: already in place)
                continue
            dst.append(v); src.append(g if g.shape == v.shape else g.reshape(v.shape))
        if dst:
            torch._foreach_copy_(dst, src)

    def _issue(self, s):
        self._pack(s)
        if self.collective:
            lo, hi = self._slices[s]
            self._works.append(dist.all_reduce(self.flat[self._off[lo]:self._off[hi]], op=dist.ReduceOp.SUM,
                                               group=self.group, async_op=True))
            self.issued += 1

    def _issue_ready(self):
        while self._next < len(self._slices) and self._left[self._next] <= 0:
            self._issue(self._next)
            self._next += 1

    def zero(self):
        """Start of a step: clear the buffer, detach the ``.grad`` views (autograd then assigns instead of adding), arm
        the hooks."""
        self.flat.zero_()
        lazy = LAZY_GRAD_PACK      # (False: .grad stay views, autograd adds into them)
        for i, p in enumerate(self.params):
            p.grad = None if lazy else self._views[i]
        self._left = [hi - lo for lo, hi in self._slices]
        self._next = 0
        self._works = []
        self._armed = self.overlap

    def all_reduce(self, _params=None):
        """End of backward: pack and issue whatever has not gone out yet (parameters without a gradient this step, a
        rank that skipped backward, overlap off, a single process) and wait for the sums.  The result is the global-batch
        gradient when every rank scaled its loss by its share of the batch; ``param.grad`` are views of the buffer."""
        self._armed = False
        while self._next < len(self._slices):
            self._issue(self._next)
            self._next += 1
        if self._works:
            import time
            t0 = time.perf_counter()
            for w in self._works:
                w.wait()
            self.wait_s += time.perf_counter() - t0
        self._works = []


def broadcast_parameters(module: torch.nn.Module, src: int = 0):
    if collectives_on():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src)


def broadcast_array(arr: np.ndarray, src: int = 0, device=None) -> np.ndarray:
    """The same int64 array on every rank (e.g. the epoch's root permutation: the ranks must cut the SAME
    permutation, which their private numpy generators do not guarantee)."""
    if not collectives_on():
        return arr
    on_gpu = dist.get_backend() == "nccl"
    t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.int64))
    if on_gpu:
        t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    dist.broadcast(t, src=src)
    return t.cpu().numpy()
