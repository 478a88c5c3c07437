"""Drop-in for the reference's optimizer module (app/utils/optimizer.py): ``create_optimizer_or_freeze_model``
and ``Adam`` (per-attribute learning rates, ``name2pg``, ``set_pervoxel_lr``) with the update done by ONE fused
HIP kernel per parameter tensor (``esr_adam_step``) instead of ~10 dense torch passes.

Same state layout as the reference (``state[p] = {"step", "exp_avg", "exp_avg_sq"}``) so checkpoints of the
optimizer interchange.  amsgrad is not provided (the reference never enables it).  Parameters stored
channels-last (colour grids) are updated in their storage order -- Adam is elementwise, any consistent
order gives the same values.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib


def create_optimizer_or_freeze_model(model: nn.Module, **lrates: float):
    """optimizer.py:11-60: one param group per named attribute with lr > 0, freeze the others."""
    groups = []
    for k, lr in lrates.items():
        if not hasattr(model, k):
            continue
        param = getattr(model, k)
        if param is None:
            print(f"create_optimizer_or_freeze_model: param {k} not exist")
            continue
        if lr > 0:
            params = list(param.parameters()) if isinstance(param, nn.Module) else [param]
            groups.append({"params": params, "lr": lr, "name": k})
        else:
            for p in (param.parameters() if isinstance(param, nn.Module) else [param]):
                p.requires_grad = False
    return Adam(groups, betas=(0.9, 0.99))


def _flat_storage(t: torch.Tensor) -> torch.Tensor:
    """1-D view over the tensor's memory (dense in SOME dimension order: contiguous or channels-last)."""
    if t.is_contiguous():
        return t.view(-1)
    if t.dim() == 5 and t.is_contiguous(memory_format=torch.channels_last_3d):
        return t.permute(0, 2, 3, 4, 1).view(-1)
    raise RuntimeError("fused Adam needs densely stored parameters")


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if amsgrad:
            raise NotImplementedError("amsgrad is never enabled by the reference and is not on the HIP path")
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 \
                or not 0.0 <= weight_decay:
            raise ValueError("invalid Adam hyper-parameter")
        self.per_lr = None
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False))
        self.name2pg = {pg["name"]: pg for pg in self.param_groups if "name" in pg}

    def set_pervoxel_lr(self, count):
        assert self.param_groups[0]["params"][0].shape == count.shape
        self.per_lr = count.float() / count.max()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("the fused Adam runs on libesr_hip.so and needs GPU parameters (no CPU fallback)")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                g = p.grad
                if g.stride() != p.stride():            # bring the gradient into the parameter's storage order
                    g = torch.empty_like(p, memory_format=torch.preserve_format).copy_(g)
                plr = None
                if self.per_lr is not None and p.shape == self.per_lr.shape:
                    plr = torch.empty_like(p, memory_format=torch.preserve_format).copy_(self.per_lr)
                pf, gf, mf, vf = _flat_storage(p), _flat_storage(g), _flat_storage(st["exp_avg"]), _flat_storage(st["exp_avg_sq"])
                rc = L.esr_adam_step(_lib.ptr(pf), _lib.ptr(gf), _lib.ptr(mf), _lib.ptr(vf),
                                     _lib.ptr(_flat_storage(plr)) if plr is not None else None,
                                     C.c_int64(pf.numel()), C.c_float(group["lr"]), C.c_float(beta1), C.c_float(beta2),
                                     C.c_float(group["eps"]), C.c_float(group["weight_decay"]), int(st["step"]),
                                     _lib.stream_ptr(p.device))
                _lib.check(rc, "esr_adam_step")
        return loss
