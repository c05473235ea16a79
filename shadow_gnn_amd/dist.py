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


def init_from_env(backend: Optional[str] = None):
    """torchrun-style environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).
    Returns (rank, local_rank, world_size); a no-op for single-process runs."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # SHADOW_DIST_BACKEND=gloo lets several ranks share one GPU (functional tests only)
            backend = os.environ.get("SHADOW_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


class GradSync:
    """All gradients live in one flat fp32 buffer (``param.grad`` are views of it), so zeroing them is one
    memset and the data-parallel exchange is a few SUM all-reduces over contiguous slices of it.

    Overlap with backward: the buffer is cut into ``num_buckets`` slices at parameter boundaries, in REVERSE
    parameter order (the order backward produces gradients in).  A post-accumulate hook per parameter counts a
    slice down; when it is complete its all-reduce is issued asynchronously (RCCL runs it on its own stream,
    ordered behind the kernels queued so far) while backward keeps producing the earlier layers' gradients.
    Slices are always issued in slice order, so every rank issues the same sequence of collectives whatever
    order its hooks fire in -- including a rank that had nothing to back-propagate this step."""

    def __init__(self, params: Iterable[torch.nn.Parameter], world_size: Optional[int] = None, group=None,
                 num_buckets: int = 2, overlap: bool = True):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.world_size = world_size if world_size is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.group = group
        sizes = [p.numel() for p in self.params]
        n = sum(sizes)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self._off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        for i, p in enumerate(self.params):
            assert p.dtype == torch.float32
            p.grad = self.flat[self._off[i]:self._off[i + 1]].view_as(p)
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
        self.overlap = bool(overlap) and self.world_size > 1
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

    def _issue(self, s):
        lo, hi = self._slices[s]
        self._works.append(dist.all_reduce(self.flat[self._off[lo]:self._off[hi]], op=dist.ReduceOp.SUM,
                                           group=self.group, async_op=True))

    def _issue_ready(self):
        while self._next < len(self._slices) and self._left[self._next] <= 0:
            self._issue(self._next)
            self._next += 1

    def zero(self):
        """Start of a step: clear the buffer and arm the hooks."""
        self.flat.zero_()
        # something may have dropped a .grad (zero_grad(set_to_none=True)): re-attach the views.  (A .grad that is set
        # stays the view: autograd accumulates into it in place.)
        if any(p.grad is None for p in self.params):
            for i, p in enumerate(self.params):
                if p.grad is None:
                    p.grad = self.flat[self._off[i]:self._off[i + 1]].view_as(p)
        self._left = [hi - lo for lo, hi in self._slices]
        self._next = 0
        self._works = []
        self._armed = self.overlap

    def all_reduce(self, _params=None):
        """End of backward: issue whatever has not gone out yet (parameters without a gradient this step, a
        rank that skipped backward, overlap off) and wait for the sums.  The result is the global-batch
        gradient when every rank scaled its loss by its share of the batch."""
        self._armed = False
        if self.world_size <= 1:
            return
        while self._next < len(self._slices):
            self._issue(self._next)
            self._next += 1
        for w in self._works:
            w.wait()
        self._works = []


def broadcast_parameters(module: torch.nn.Module, src: int = 0):
    if dist.is_initialized() and dist.get_world_size() > 1:
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src)


def broadcast_array(arr: np.ndarray, src: int = 0, device=None) -> np.ndarray:
    """The same int64 array on every rank (e.g. the epoch's root permutation: the ranks must cut the SAME
    permutation, which their private numpy generators do not guarantee)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return arr
    on_gpu = dist.get_backend() == "nccl"
    t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.int64))
    if on_gpu:
        t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    dist.broadcast(t, src=src)
    return t.cpu().numpy()
