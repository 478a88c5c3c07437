"""One training step of the fine stage on the HIP path, without autograd in the
loop: forward kernels -> fused loss+gradient kernel -> backward kernels.

It reproduces the body of the reference trainer's hot loop
(app/fine/fine.py:346-395: renderer call, white-background add + clamps, MSE on
sRGB, MSE on gamma-encoded linear colour, entropy of alphainv_last, backward)
for ``bench.py`` and for data-parallel runs; ``VoxurfF.forward`` + a torch loss +
``loss.backward()`` (the drop-in route used by the reference's ``Fine.learn``)
enqueues exactly the same renderer kernels and is tested to give the same
numbers.  The optimizer step is outside the named path and not
part of this class.

Data parallelism: rays are independent, so every rank runs the
same step on its contiguous shard of the global batch and the parameter
gradients live in ONE flat buffer that is summed with two all-reduces (RCCL over
xGMI when the process group is NCCL): the dense-grid part as soon as the grid scatters are
done -- overlapped with the weight-gradient kernels -- and the small MLP part at the end.  Losses are normalised by the GLOBAL ray count so the reduced
gradient equals the single-process gradient of the full batch.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from .fine_engine import KIND_RADIANCE, KIND_TONEMAP


def shard_batch(batch: Dict[str, torch.Tensor], rank: int, world: int) -> Dict[str, torch.Tensor]:
    """Contiguous ray shard of a global batch (last rank takes the remainder)."""
    n = next(iter(batch.values())).shape[0]
    per = n // world
    lo = rank * per
    hi = n if rank == world - 1 else lo + per
    return {k: v[lo:hi].contiguous() for k, v in batch.items()}


def dp_loss_weights(n_local: int, global_rays, entropy_owner: bool, weight_entropy_last: float):
    """(scale, w_ent) for one rank's shard.  The two MSE terms are means over the batch, so a
    shard's mean-normalised loss/gradients are multiplied by ``scale = n_local / n_global``; the
    entropy term (last ray only, fine.py:378) is not a mean: it is added by the owning rank only
    and pre-divided by ``scale`` so that the common rescale leaves it at ``weight_entropy_last``."""
    scale = 1.0 if global_rays is None else n_local / float(global_rays)
    w_ent = (weight_entropy_last / scale) if entropy_owner else 0.0
    return scale, w_ent


class FineStep:
    def __init__(self, model, white_bg: bool = True, weight_linear: float = 0.1,
                 weight_entropy_last: float = 0.001, process_group=None):
        self.model = model
        self.white_bg = white_bg
        self.weight_linear = weight_linear
        self.weight_entropy_last = weight_entropy_last
        self.pg = process_group
        self._names = None
        self._flat = None

    # names follow state_dict / named_parameters of the renderer
    def _param_names(self):
        if self._names is None:
            names = []
            for net in ("off_rgbnet", "emo_rgbnet", "tonemapper"):
                seq = "linear" if net != "tonemapper" else "srgb"
                mod = getattr(getattr(self.model, net), seq)
                for key, sub in mod.named_modules():
                    if isinstance(sub, torch.nn.Linear):
                        names += [f"{net}.{seq}.{key}.weight", f"{net}.{seq}.{key}.bias"]
            self._names = names
        return self._names

    def _alloc_grads(self, dev):
        """One flat zero buffer holding every gradient (a single memset, a single all-reduce)."""
        m = self.model
        X, Y, Z = [int(v) for v in m.world_size]
        shapes = [("sdf.grid", (1, 1, X, Y, Z)), ("off_color.grid", (1, X, Y, Z, 6)),
                  ("emo_color.grid", (1, X, Y, Z, 6))]
        shapes += [(n, tuple(p.shape)) for n, p in zip(self._param_names(), m._mlp_params())]
        total = sum(int(torch.Size(s).numel()) for _, s in shapes)
        if self._flat is None or self._flat.numel() != total:
            self._flat = torch.empty(total, dtype=torch.float32, device=dev)
        self._flat.zero_()
        out, o = {}, 0
        for i, (n, s) in enumerate(shapes):
            k = int(torch.Size(s).numel())
            out[n] = self._flat[o:o + k].view(s)
            o += k
            if i == 2:
                self._n_grid = o          # [0, _n_grid): the three dense grids; the rest: MLP tensors
        return out

    @torch.no_grad()
    def forward_loss_backward(self, batch: Dict[str, torch.Tensor], s_val: float,
                              global_rays: Optional[int] = None,
                              entropy_owner: bool = True) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        """``global_rays``: size of the whole batch when ``batch`` is one rank's shard.
        ``entropy_owner``: the reference's entropy term looks at the LAST ray of the batch only
        (fine.py:378), so under sharding exactly one rank -- the one holding the global last ray --
        must add it."""
        m = self.model
        eng = m.engine
        m.s_val = s_val
        ps = m._mlp_params()
        eng.pack("off", KIND_RADIANCE, list(ps[0:8:2]), list(ps[1:8:2]))
        eng.pack("emo", KIND_RADIANCE, list(ps[8:16:2]), list(ps[9:16:2]))
        eng.pack("tone", KIND_TONEMAP, list(ps[16:20:2]), list(ps[17:20:2]))
        ctx, last, srgb, lin = eng.forward(
            m.scene_struct(), batch["rays_o"], batch["rays_d"], batch["viewdirs"], batch["em_modes"],
            m.mask_cache.density.view(*m.mask_cache.density.shape[2:]),
            m.sdf.device_view(), m.off_color.device_view(), m.emo_color.device_view())
        m.last_counts = ctx.counts
        scale, w_ent = dp_loss_weights(last.shape[0], global_rays, entropy_owner, self.weight_entropy_last)
        loss, g_last, g_srgb, g_lin = eng.loss_fwd_bwd(last, srgb, lin, batch["rgbs"], self.white_bg,
                                                       self.weight_linear, w_ent)
        if scale != 1.0:
            loss, g_last, g_srgb, g_lin = loss * scale, g_last * scale, g_srgb * scale, g_lin * scale
        g = self._alloc_grads(last.device)
        names = self._param_names()
        grads = dict(sdf=g["sdf.grid"], off_color=g["off_color.grid"], emo_color=g["emo_color.grid"],
                     off_w=[g[n] for n in names[0:8:2]], off_b=[g[n] for n in names[1:8:2]],
                     emo_w=[g[n] for n in names[8:16:2]], emo_b=[g[n] for n in names[9:16:2]],
                     tone_w=[g[n] for n in names[16:20:2]], tone_b=[g[n] for n in names[17:20:2]])
        works = []
        if self.pg is not None:
            import torch.distributed as dist

            def after_grids():
                # grid gradients (218 MB at C2, >99 % of the payload) are final here: their all-reduce
                # runs on RCCL's stream underneath the wgrad kernels that the engine enqueues next
                works.append(dist.all_reduce(self._flat[: self._n_grid], group=self.pg, async_op=True))
        else:
            after_grids = None
        eng.backward(ctx, g_last, g_srgb, g_lin, grads, after_grids=after_grids)
        if self.pg is not None:
            works.append(dist.all_reduce(self._flat[self._n_grid:], group=self.pg, async_op=True))
            works.append(dist.all_reduce(loss, group=self.pg, async_op=True))
            for w in works:
                w.wait()                  # stream-level wait: the caller's stream sees reduced gradients
        g["off_color.grid"] = g["off_color.grid"].permute(0, 4, 1, 2, 3)     # logical [1,6,X,Y,Z]
        g["emo_color.grid"] = g["emo_color.grid"].permute(0, 4, 1, 2, 3)
        return loss, g

    def assign_grads(self, grads: Dict[str, torch.Tensor]):
        """Expose the step's gradients as ``param.grad`` (for a torch optimizer)."""
        for n, p in self.model.named_parameters():
            if n in grads:
                p.grad = grads[n]
