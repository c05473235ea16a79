"""Clip-by-global-norm + Adam over ONE flat fp32 buffer.

``dist.GradSync`` already keeps every gradient as a view of one flat buffer; ``FlatAdam`` does the same for the
parameters and the two Adam moments, so that the optimiser part of a training step (shaDow/models.py:225-226:
``clip_grad_norm_(parameters, 5)`` and ``optimizer.step()``) is a fixed handful of kernels on ~0.6 M floats instead
of two foreach passes over ~30 tensors -- what is left of the step's host time at the reference's small batch sizes
sits in exactly those passes.  Arithmetic: torch.optim.Adam's defaults (betas (0.9, 0.999), eps 1e-8, no weight
decay, no amsgrad), statement for statement, and torch.nn.utils.clip_grad_norm_'s rule."""
import math
import os
from typing import Tuple

import torch


class FlatAdam:
    def __init__(self, grad_sync, lr: float, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8):
        self.sync = grad_sync
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        params = grad_sync.params
        self.flat_param = torch.empty_like(grad_sync.flat)
        off = 0
        with torch.no_grad():
            for p in params:
                n = p.numel()
                self.flat_param[off:off + n].copy_(p.detach().reshape(-1))
                p.data = self.flat_param[off:off + n].view_as(p)          # the module's parameters ARE the flat buffer
                off += n
        self.exp_avg = torch.zeros_like(self.flat_param)
        self.exp_avg_sq = torch.zeros_like(self.flat_param)
        self.step_count = 0
        self._scratch = None
        self.param_groups = [dict(lr=self.lr, betas=self.betas, eps=self.eps, params=list(params))]   # (torch-like, read-only)

    # torch.nn.utils.clip_grad_norm_(params, max_norm): no host synchronisation
    def clip_(self, max_norm: float) -> torch.Tensor:
        g = self.sync.flat
        total = torch.linalg.vector_norm(g, 2)
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
        g.mul_(coef)
        return total

    @torch.no_grad()
    def step(self):
        self.step_count += 1
        b1, b2 = self.betas
        g = self.sync.flat
        self.exp_avg.lerp_(g, 1 - b1)
        self.exp_avg_sq.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1 ** self.step_count
        bc2 = 1 - b2 ** self.step_count
        denom = (self.exp_avg_sq.sqrt() / math.sqrt(bc2)).add_(self.eps)
        self.flat_param.addcdiv_(self.exp_avg, denom, value=-self.lr / bc1)

    @torch.no_grad()
    def clip_step_(self, max_norm: float) -> torch.Tensor:
        """clip_(max_norm) + step() in two HIP launches (sl_clip_adam) instead of a dozen element-wise torch kernels; falls
        back to the torch statements off the GPU.  Returns the gradient norm before clipping (a device scalar)."""
        g = self.sync.flat
        if not g.is_cuda or os.environ.get("SHADOW_FUSED_ADAM", "1") == "0":
            total = self.clip_(max_norm)
            self.step()
            return total
        from . import _lib, ops
        lib = _lib.load()
        if self._scratch is None:
            self._scratch = torch.empty(int(lib.sl_clip_adam_scratch_floats()), dtype=torch.float32, device=g.device)
        self.step_count += 1
        _lib.check(lib.sl_clip_adam(self.flat_param.data_ptr(), g.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                    g.numel(), self.lr, self.betas[0], self.betas[1], self.eps, self.step_count, float(max_norm),
                                    self._scratch.data_ptr(), ops._stream(g)))
        return self._scratch[-1]

    def zero_grad(self, set_to_none: bool = False):
        self.sync.zero()

    def state_dict(self):
        return dict(step=self.step_count, exp_avg=self.exp_avg.clone(), exp_avg_sq=self.exp_avg_sq.clone(),
                    lr=self.lr, betas=self.betas, eps=self.eps)

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.lr, self.betas, self.eps = float(sd["lr"]), tuple(sd["betas"]), float(sd["eps"])
