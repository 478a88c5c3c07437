"""Drop-in for the reference's optimizer module (app/utils/optimizer.py): ``create_optimizer_or_freeze_model``
and ``Adam`` (per-attribute learning rates, ``name2pg``, ``set_pervoxel_lr``) with the update done by ONE fused
HIP kernel per parameter tensor (``esr_adam_step``) instead of ~10 dense torch passes.

Same state layout as the reference (``state[p] = {"step", "exp_avg", "exp_avg_sq"}``) so checkpoints of the
optimizer interchange.  amsgrad is not provided (the reference never enables it).  Parameters stored
channels-last (colour grids) are updated in their storage order -- Adam is elementwise, any consistent
order gives the same values; moments restored from a reference checkpoint arrive contiguous
(``Optimizer.load_state_dict`` keeps the loaded strides) and are brought into the parameter's order first.

``CosineLR`` is the trainers' learning-rate schedule (optimizer.py:231-275): warm-up then half cosine, handed
out as the multiplicative step-to-step factor the trainers apply to every group's lr.
"""
from __future__ import annotations

import ctypes as C
import math

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


def _same_layout(a: torch.Tensor, b: torch.Tensor) -> bool:
    """Equal strides over the dimensions that have more than one element (a size-1 dimension's stride is free)."""
    return a.shape == b.shape and all(sa == sb for n, sa, sb in zip(a.shape, a.stride(), b.stride()) if n > 1)


def _like_param(p: torch.Tensor, src: torch.Tensor) -> torch.Tensor:
    """``src`` re-stored in the memory order of ``p`` (no copy when it already is)."""
    if _same_layout(src, p) and src.dtype == p.dtype and src.device == p.device:
        return src
    return torch.empty_like(p, memory_format=torch.preserve_format).copy_(src)


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
        self._per_lr_for = {}            # id(param) -> per_lr in that parameter's storage order (built once)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False))
        self.name2pg = {pg["name"]: pg for pg in self.param_groups if "name" in pg}

    def __setstate__(self, state):
        super().__setstate__(state)
        for group in self.param_groups:
            group.setdefault("amsgrad", False)
        self.__dict__.setdefault("per_lr", None)
        self._per_lr_for = {}

    def set_pervoxel_lr(self, count):
        assert self.param_groups[0]["params"][0].shape == count.shape
        self.per_lr = count.float() / count.max()
        self._per_lr_for = {}

    def load_state_dict(self, state_dict):
        """torch keeps the strides of the LOADED moments; a checkpoint written by the reference (fine.py:254-255)
        holds them contiguous while the colour grids live channels-last here.  Bring both moments into each
        parameter's storage order so the flat views of step() pair element i of p, g, m and v."""
        super().load_state_dict(state_dict)
        for group in self.param_groups:
            for p in group["params"]:
                st = self.state.get(p)
                if st:
                    for k in ("exp_avg", "exp_avg_sq"):
                        if k in st:
                            st[k] = _like_param(p, st[k])

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
                else:                                   # state injected without load_state_dict (tests, hand-made resumes)
                    for k in ("exp_avg", "exp_avg_sq"):
                        if not _same_layout(st[k], p):
                            st[k] = _like_param(p, st[k])
                st["step"] += 1
                g = _like_param(p, p.grad)              # the gradient in the parameter's storage order (usually a no-op)
                plr = None
                if self.per_lr is not None and p.shape == self.per_lr.shape:
                    plr = self._per_lr_for.get(id(p))
                    if plr is None:
                        plr = self._per_lr_for[id(p)] = _like_param(p, self.per_lr.to(device=p.device, dtype=p.dtype))
                pf, gf, mf, vf = _flat_storage(p), _flat_storage(g), _flat_storage(st["exp_avg"]), _flat_storage(st["exp_avg_sq"])
                rc = L.esr_adam_step(_lib.ptr(pf), _lib.ptr(gf), _lib.ptr(mf), _lib.ptr(vf),
                                     _lib.ptr(_flat_storage(plr)) if plr is not None else None,
                                     C.c_int64(pf.numel()), C.c_float(group["lr"]), C.c_float(beta1), C.c_float(beta2),
                                     C.c_float(group["eps"]), C.c_float(group["weight_decay"]), int(st["step"]),
                                     _lib.stream_ptr(p.device))
                _lib.check(rc, "esr_adam_step")
        return loss


def _hip_adam(p, g, m, v, lr, beta1, beta2, eps, step):
    rc = _lib.lib().esr_adam_step(_lib.ptr(p), _lib.ptr(g), _lib.ptr(m), _lib.ptr(v), None, C.c_int64(p.numel()),
                                  C.c_float(lr), C.c_float(beta1), C.c_float(beta2), C.c_float(eps), C.c_float(0.0),
                                  int(step), _lib.stream_ptr(p.device))
    _lib.check(rc, "esr_adam_step")


class ShardedGridAdam:
    """Adam over the dense grids with the state and the update sharded over the data-parallel ranks (SURVEY 8(e)
    option 2; ZeRO-1 style).  ``grids``: grad_sync.ShardedGrids (the grid parameters as one flat buffer).  Every step:
    ``grids.reduce_scatter(flat_grad)`` (started by the trainer step as soon as the grid gradients are complete) ->
    the fused Adam on this rank's shard only -> ``all_gather`` of the updated parameters.  The moments exist for the
    owned shard only (1/G of the dense optimizer's memory and update traffic).  Same arithmetic as ``Adam`` above
    (``esr_adam_step``), per-attribute learning rates by position in the flat buffer; a per-voxel lr is not supported
    (only the alphamask pre-stage of the reference uses one).  ``adam_fn``: test double for CPU rehearsals."""

    def __init__(self, grids, lrs, betas=(0.9, 0.99), eps=1e-8, adam_fn=None):
        self.grids = grids
        self.lr = {n: float(lrs[n]) for n in grids.names}
        self.betas, self.eps = betas, eps
        self.step_count = 0
        self.exp_avg = torch.zeros_like(grids.grad_shard)
        self.exp_avg_sq = torch.zeros_like(grids.grad_shard)
        self._adam = adam_fn or _hip_adam

    def scale_lr(self, factor: float):
        """what the trainers do to every param group after a step (fine.py:410-415)"""
        for n in self.lr:
            self.lr[n] *= factor

    @torch.no_grad()
    def step(self):
        G = self.grids
        G.wait()                                          # the reduce-scatter of this step's grid gradients
        self.step_count += 1
        lo, hi = G.my_range()
        for name, b, e in G.bounds:                       # the shard may straddle two grids (different lr each)
            a, z = max(lo, b), min(hi, e)
            if a < z:
                self._adam(G.flat[a:z], G.grad_shard[a - lo:z - lo], self.exp_avg[a - lo:z - lo],
                           self.exp_avg_sq[a - lo:z - lo], self.lr[name], self.betas[0], self.betas[1], self.eps,
                           self.step_count)
        G.all_gather()

    def state_dict(self):
        """This rank's shard of the state (resume with the SAME world size: ``load_state_dict``).  For a checkpoint in
        the reference's per-parameter format -- readable by the dense ``Adam``, by the reference, or by a run with
        another world size -- use ``full_state`` / ``load_full_state``."""
        return dict(step=self.step_count, exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq, lr=dict(self.lr),
                    shard=self.grids.my_range(), world=self.grids.world, names=list(self.grids.names))

    def load_state_dict(self, sd):
        G = self.grids
        if tuple(sd["shard"]) != tuple(G.my_range()) or int(sd.get("world", G.world)) != G.world:
            raise ValueError(f"shard state {sd['shard']} (world {sd.get('world')}) does not fit this rank's shard "
                             f"{G.my_range()} (world {G.world}); convert through full_state / load_full_state")
        if set(sd["lr"]) != set(self.lr):
            raise ValueError(f"learning-rate keys {sorted(sd['lr'])} do not match the grids {sorted(self.lr)}")
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"].to(self.exp_avg.device))
        self.exp_avg_sq.copy_(sd["exp_avg_sq"].to(self.exp_avg_sq.device))
        self.lr = {k: float(v) for k, v in sd["lr"].items()}

    @torch.no_grad()
    def full_state(self):
        """COLLECTIVE.  The whole state in the reference's format on every rank: {name: {"step", "exp_avg",
        "exp_avg_sq"}} with contiguous [1,C,X,Y,Z] moments (app/utils/optimizer.py:104-121), plus the lrs."""
        from .grad_sync import _all_gather
        G = self.grids
        lo, hi = G.my_range()
        out = {n: {"step": self.step_count} for n in G.names}
        for key, shard in (("exp_avg", self.exp_avg), ("exp_avg_sq", self.exp_avg_sq)):
            flat = torch.zeros(G.padded, dtype=torch.float32, device=shard.device)
            flat[lo:hi].copy_(shard)
            _all_gather(flat, flat[lo:hi], G.pg)
            for n, t in G.to_reference_layout(flat).items():
                out[n][key] = t
        return dict(state=out, lr=dict(self.lr))

    @torch.no_grad()
    def load_full_state(self, full):
        """Inverse of ``full_state`` (no communication: every rank cuts its own shard out of the full moments); accepts a
        state written under any world size, by the dense ``Adam`` or by the reference."""
        G = self.grids
        lo, hi = G.my_range()
        st = full["state"]
        steps = {int(st[n]["step"]) for n in G.names}
        if len(steps) != 1:
            raise ValueError(f"the grids' step counts differ: {steps}")
        self.step_count = steps.pop()
        for key, shard in (("exp_avg", self.exp_avg), ("exp_avg_sq", self.exp_avg_sq)):
            flat = G.from_reference_layout({n: st[n][key] for n in G.names}, device=shard.device)
            shard.copy_(flat[lo:hi])
        if "lr" in full:
            self.lr = {n: float(full["lr"][n]) for n in G.names}


class CosineLR:
    """optimizer.py:231-275.  ``ratio(i)``: linear warm-up from ``warm_up_min_ratio`` to 1 over ``warm_up_iters``
    steps (held at the minimum with ``const_warm_up``), then a half cosine from 1 down to ``cos_min_ratio`` at
    ``n_iters``; ``warm_up_iters == -1`` means "warm up for the whole run".  The trainers multiply every group's lr
    by ``decay_factor`` once per step (fine.py:410-415): the ratio of this step's schedule value to the previous
    step's, which is what reading the property returns (and advances).  A run resumed at ``cur_step`` continues the
    same sequence of factors."""

    def __init__(self, cfg, cur_step: int = 0):
        tr = cfg.app.trainer
        self.cfg = cfg
        self.cur_step = cur_step
        self.n_iters = tr.n_iters
        self.warm_up_iters = tr.n_iters if tr.warm_up_iters == -1 else tr.warm_up_iters
        self.warm_up_min_ratio = tr.warm_up_min_ratio
        self.const_warm_up = tr.const_warm_up
        self.cos_min_ratio = tr.cos_min_ratio
        self.pre_decay_factor = 1.0 if cur_step == 0 else self.cosine_lr_func(cur_step - 1)
        self.pos_decay_factor = self.cosine_lr_func(cur_step)

    def cosine_lr_func(self, iter: int) -> float:
        w = self.warm_up_iters
        if iter < w:
            if self.const_warm_up:
                return self.warm_up_min_ratio
            return self.warm_up_min_ratio + (1 - self.warm_up_min_ratio) * (iter / w)
        phase = (iter - w) / (self.n_iters - w) * math.pi
        return (1 + math.cos(phase)) * 0.5 * (1 - self.cos_min_ratio) + self.cos_min_ratio

    @property
    def decay_factor(self) -> float:
        now = self.cosine_lr_func(self.cur_step)
        factor = now / self.pre_decay_factor
        self.pre_decay_factor = now
        self.cur_step += 1
        return factor
