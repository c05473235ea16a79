"""Data-parallel training of the hot path: one process per GPU, minibatches of
disjoint subgraphs sharded over the ranks (shadow_gnn_amd.minibatch slices the
shared root permutation), and ONE gradient all-reduce per step over a single
flattened fp32 bucket (RCCL over xGMI through torch.distributed backend "nccl";
"gloo" on CPU for the tests).  The reference has no multi-GPU code at all
(SURVEY.md section 8(e)); this is new design, not a translation."""
import os
from typing import Iterable, List, Optional

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
    """All gradients live in one flat fp32 buffer (``param.grad`` are views of it),
    so the data-parallel exchange is a single all-reduce of ~2-12 MB per step and
    zeroing the gradients is one memset."""

    def __init__(self, params: Iterable[torch.nn.Parameter], world_size: Optional[int] = None,
                 group=None):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.world_size = world_size if world_size is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.group = group
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            assert p.dtype == torch.float32
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero(self):
        self.flat.zero_()
        # autograd may have replaced .grad (e.g. after zero_grad(set_to_none=True)); re-attach
        off = 0
        for p in self.params:
            view = self.flat[off:off + p.numel()].view_as(p)
            if p.grad is None or p.grad.data_ptr() != view.data_ptr():
                p.grad = view
            off += p.numel()

    def all_reduce(self, _params=None):
        """Average the gradients over the ranks (each rank holds the mean over
        its equal share of the global batch, so the result is the global mean)."""
        if self.world_size > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.div_(self.world_size)


def broadcast_parameters(module: torch.nn.Module, src: int = 0):
    if dist.is_initialized() and dist.get_world_size() > 1:
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src)
