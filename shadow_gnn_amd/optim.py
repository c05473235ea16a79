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
        self.flat_param = torch.zeros_like(grad_sync.flat)
        with torch.no_grad():
            for i, p in enumerate(params):
                n, off = p.numel(), self._offset(i)                       # (the gradient buffer's layout: line-aligned tensors, zero pads)
                self.flat_param[off:off + n].copy_(p.detach().reshape(-1))
                p.data = self.flat_param[off:off + n].view_as(p)          # the module's parameters ARE the flat buffer
        self.exp_avg = torch.zeros_like(self.flat_param)
        self.exp_avg_sq = torch.zeros_like(self.flat_param)
        self.step_count = 0
        self._scratch = None
        self.param_groups = [dict(lr=self.lr, betas=self.betas, eps=self.eps, params=list(params))]   # (torch-like, read-only)

    def _offset(self, i: int) -> int:
        """Start of parameter i in the flat buffers: dist.GradSync's table, or back to back for a plain holder of a flat buffer."""
        off = getattr(self.sync, "_off", None)
        if off is not None:
            return int(off[i])
        return sum(p.numel() for p in self.sync.params[:i])

    def _gather_grads(self):
        """The gradients must be IN the flat buffer: GradSync packs them lazily (param.grad is None during backward and
        the buffer is filled inside GradSync.all_reduce).  DeepGNN._finish_update has normally done that already; a caller
        that runs backward and then steps this optimiser directly would otherwise update from an all-zero buffer.
        (Idempotent: a second call finds nothing left to pack or to reduce.)"""
        gather = getattr(self.sync, "all_reduce", None)      # (a plain holder of a flat buffer has nothing to gather)
        if gather is not None:
            gather()

    # torch.nn.utils.clip_grad_norm_(params, max_norm): no host synchronisation
    def clip_(self, max_norm: float) -> torch.Tensor:
        self._gather_grads()
        g = self.sync.flat
        total = torch.linalg.vector_norm(g, 2)
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
        g.mul_(coef)
        return total

    @torch.no_grad()
    def step(self):
        self._gather_grads()
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
        self._gather_grads()
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

    def _param_slices(self):
        for i, p in enumerate(self.sync.params):
            off = self._offset(i)
            yield off, off + p.numel(), p

    def state_dict(self):
        """torch.optim.Adam's layout -- {'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [...]} -- so that
        the reference's optimizer checkpoints (main.py:130-132, saved_optimizer_*.pkl) and this class's are interchangeable:
        the flat moment buffers are cut per parameter."""
        state = {}
        for i, (lo, hi, p) in enumerate(self._param_slices()):
            state[i] = dict(step=torch.tensor(float(self.step_count)), exp_avg=self.exp_avg[lo:hi].view_as(p).clone(),
                            exp_avg_sq=self.exp_avg_sq[lo:hi].view_as(p).clone())
        group = dict(lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=0, amsgrad=False, maximize=False, foreach=None,
                     capturable=False, differentiable=False, fused=None, params=list(range(len(self.sync.params))))
        return dict(state=state, param_groups=[group])

    def load_state_dict(self, sd):
        if "state" not in sd:                         # the flat layout of round 2
            # (that layout packed the parameters back to back; the buffers now start every tensor on a dist.FLAT_ALIGN line:
            #  each parameter's slice is copied to its own offset -- ADVICE r4)
            total = sum(p.numel() for p in self.sync.params)
            ea, es = sd["exp_avg"].reshape(-1), sd["exp_avg_sq"].reshape(-1)
            if ea.numel() != total or es.numel() != total:
                raise ValueError(f"flat optimizer state holds {ea.numel()} / {es.numel()} moments, this model has {total} parameters")
            self.step_count = int(sd["step"])
            with torch.no_grad():
                self.exp_avg.zero_(); self.exp_avg_sq.zero_()
                src = 0
                for lo, hi, p in self._param_slices():
                    self.exp_avg[lo:hi].copy_(ea[src:src + p.numel()]); self.exp_avg_sq[lo:hi].copy_(es[src:src + p.numel()])
                    src += p.numel()
            self.lr, self.betas, self.eps = float(sd["lr"]), tuple(sd["betas"]), float(sd["eps"])
            return
        g = sd["param_groups"][0]
        if len(sd["param_groups"]) != 1 or len(g["params"]) != len(self.sync.params):
            raise ValueError("optimizer state does not match this model's parameter list")
        if g.get("weight_decay", 0) or g.get("amsgrad", False):
            raise ValueError("FlatAdam implements Adam without weight decay / amsgrad")
        self.lr, self.betas, self.eps = float(g["lr"]), (float(g["betas"][0]), float(g["betas"][1])), float(g["eps"])
        steps = set()
        with torch.no_grad():
            for i, (lo, hi, p) in enumerate(self._param_slices()):
                st = sd["state"].get(g["params"][i], sd["state"].get(i))
                if st is None:                        # a parameter that never received a gradient has no entry in torch
                    self.exp_avg[lo:hi].zero_(); self.exp_avg_sq[lo:hi].zero_()
                    continue
                self.exp_avg[lo:hi].copy_(st["exp_avg"].reshape(-1)); self.exp_avg_sq[lo:hi].copy_(st["exp_avg_sq"].reshape(-1))
                steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): FlatAdam keeps one")
        self.step_count = steps.pop() if steps else 0
