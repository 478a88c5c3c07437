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
gradients live in ONE flat buffer.  Its dense-grid part (> 99 % of the bytes) is exchanged
brick-sparsely over RCCL / xGMI as soon as the grid scatters are done (grad_sync.GridGradSync),
underneath the weight-gradient kernels that run on a second HIP stream; the small MLP part and
the loss follow as plain all-reduces.  Losses are normalised by the GLOBAL ray count so the
reduced gradient equals the single-process gradient of the full batch.
"""
from __future__ import annotations

import os

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


def lts_point_share(num_ltspts: int, world: int, rank: int):
    """(points, weight) of one rank when ``LtsStep(split_points=True)`` splits the reference's ``num_ltspts`` surface points
    over ``world`` ranks: the remainder goes to the first ranks one point each (the counts add up to ``num_ltspts``), and a
    rank's per-point loss terms -- means over ITS points -- enter the global mean with ``points / num_ltspts``."""
    base, rem = divmod(int(num_ltspts), int(world))
    n = base + (1 if rank < rem else 0)
    return n, n / float(num_ltspts)


def default_grid_sync(world: int) -> str:
    """The exchange form when ``ESR_GRAD_SYNC`` does not force one: ``sparse`` for 2-4 ranks (few usable xGMI links: the wire
    time dominates), ``dense`` otherwise (link arithmetic, to be replaced by the driver's multi-GPU measurements)."""
    return "sparse" if 2 <= int(world) <= 4 else "dense"


def _grid_sync(step, eng):
    """after_grids callback of a data-parallel step: how the dense-grid gradients are summed over ranks.

    * ``sparse`` -- grad_sync.GridGradSync: brick flags, union, ONE all-reduce of the packed union (C2: 35 MB per
      rank instead of 218 MB).  No host wait inside the exchange (fixed-capacity brick list sized from the previous
      step, verified at the end of the step); two more collective launches than the dense form, so it pays when the
      wire time dominates: few ranks = few usable xGMI links (one at N = 2).
    * ``dense`` -- one asynchronous all-reduce of the whole grid part, no host sync; underneath the weight-gradient
      kernels.  With all 7 links per GPU in play (N = 8) a ring moves 218 MB in about a millisecond, which the
      ~1.3 ms of wgrad work hides.
    * ``shard`` -- option 2 of SURVEY 8(e): reduce-scatter of the grid part into this rank's shard
      (``step.sharded``: grad_sync.ShardedGrids, attached by the caller together with optimizer.ShardedGridAdam, which
      updates the shard and all-gathers the parameters).  The grid part of the returned gradient buffer then holds
      this rank's LOCAL gradients only.
    ``ESR_GRAD_SYNC=sparse|dense|shard`` forces one; the default picks sparse for 2-4 ranks (link arithmetic, to be
    replaced by the driver's multi-GPU measurements)."""
    import torch.distributed as dist
    works = []
    mode = step._sync_mode                    # ESR_GRAD_SYNC, read once when the step object was built
    if getattr(step, "sharded", None) is not None:
        mode = "shard"
    elif mode == "shard":
        raise RuntimeError("ESR_GRAD_SYNC=shard needs step.sharded = grad_sync.ShardedGrids(model, names, group) "
                           "(and optimizer.ShardedGridAdam for the update)")
    if mode == "auto":
        mode = default_grid_sync(dist.get_world_size(step.pg))
    step.sync_mode_used = mode            # what bench.py reports (grad_exchange.mode)
    if mode == "shard":
        def after_grids():
            step.sharded.reduce_scatter(step._flat[: step._n_grid_pad])      # padded in place: no copy
    elif mode == "sparse":
        from .grad_sync import GridGradSync
        if step._sync is None:
            step._sync = GridGradSync(step.pg)

        def after_grids():
            step._sync.reduce(step._flat[: step._n_grid])      # no host wait inside; closed by _sync.verify() below
    else:
        def after_grids():
            works.append(dist.all_reduce(step._flat[: step._n_grid], group=step.pg, async_op=True))
    return after_grids, works


def _reduce_loss_and_overflow(step, eng, loss, works):
    """Data-parallel end of step: the loss and the march-overflow flag of this rank summed over the ranks in ONE small
    all-reduce; the flag goes to pinned host memory and is examined at the start of the NEXT step (by then it has landed:
    no host wait), where every rank raises together."""
    import torch.distributed as dist
    lf = getattr(eng, "loss_pair", None)      # [loss, 0] of this step (fine_engine.loss_fwd_bwd): nothing to build
    if lf is None or lf.data_ptr() != loss.data_ptr():
        lf = torch.cat([loss.reshape(1), torch.zeros(1, device=loss.device)])
    if eng.overflow_seen:
        lf[1:2].fill_(1.0)
    eng.overflow_seen = False
    w = dist.all_reduce(lf, group=step.pg, async_op=True)
    works.append(w)
    return lf


def _publish_overflow(step, lf):
    if getattr(step, "_ovf_host", None) is None:
        step._ovf_host = torch.zeros(1, pin_memory=lf.is_cuda)
    step._ovf_host.copy_(lf[1:2], non_blocking=True)
    step._ovf_event = None
    if lf.is_cuda:
        step._ovf_event = torch.cuda.Event()
        step._ovf_event.record()


def _check_overflow(step):
    """Raise (on every rank together) if a rank flagged an overflow in the PREVIOUS step.  Called right after this
    step's forward has waited for its plan header: the flag's copy was enqueued before that on the same stream, so the
    event below is already complete.  (Checked at the very start of the step it was a full host wait for the previous
    step's backward: the host could not run ahead, and every launch up to the plan wait -- march, plan, the prelude's
    packs and fill -- was exposed: 0.25 ms of GPU idle per step, tools/trace_step.py on `bench.py --force-dist`.)"""
    ev = getattr(step, "_ovf_event", None)
    if getattr(step, "_ovf_host", None) is None:
        return
    if ev is not None:
        ev.synchronize()
    if float(step._ovf_host[0]) > 0:
        step._ovf_host.zero_()
        raise RuntimeError("on at least one rank in the previous step: a ray exceeded scene.max_steps (the LDS bound of the march "
                           "kernel is wrong, its rays were skipped) or, with ESR_SPLIT_STRICT=1, an MLP operand left fp16's range "
                           "in a split-fp16 kernel (that step's gradients are not those of the f32 engine)")


def _grid_pad(step, n_grid: int) -> int:
    """Size of the grid part of a step's flat gradient buffer: with ``step.sharded`` (grad_sync.ShardedGrids) attached it
    is padded to the shard quantum, so that reduce_scatter reads the buffer in place (no per-step padded copy)."""
    sh = getattr(step, "sharded", None)
    if sh is not None and sh.n == n_grid:
        return int(sh.padded)
    return n_grid


def _warn_hw_queues(process_group):
    """Data-parallel steps need >= 8 HIP hardware queues (see esr_nerf_amd/__init__.py); warn once when the process
    was started with fewer -- the step still runs, with its side stream serialised behind the main one."""
    if process_group is None:
        return
    try:
        q = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    except ValueError:
        q = 4
    if q < 8:
        import warnings
        warnings.warn(f"GPU_MAX_HW_QUEUES={q}: with an RCCL communicator alive the engine's side stream shares a hardware "
                      "queue with the main stream (weight gradients and packing serialise); export GPU_MAX_HW_QUEUES=8 "
                      "before the process initialises HIP", RuntimeWarning, stacklevel=3)


class _OverflowScope:
    """``eng.defer_overflow`` for the duration of ONE data-parallel step (an overflow must not raise on one rank while
    the others wait in the exchange); restored afterwards so that later non-DP uses of the same engine -- the autograd
    route, single-process stepping -- raise at once again."""

    def __init__(self, eng, on):
        self.eng, self.on = eng, on

    def __enter__(self):
        self.prev = self.eng.defer_overflow
        if self.on:
            self.eng.defer_overflow = True

    def __exit__(self, *exc):
        self.eng.defer_overflow = self.prev
        return False


class _RangeGuard:
    """The ``after_grids`` callback of a trainer step, with the split-fp16 kernels' range fallback in front of it.  The engine
    calls it when the input-gradient chain and the grid scatters are enqueued and BEFORE anything leaves the rank (the
    data-parallel exchange starts inside the wrapped callback): the best place for the one host wait the check needs -- the
    device has ~0.7 ms of queued work behind the forward the host waits for, so it never idles.  ``hit``: a split launch of
    this step's forward raised the flag; the exchange was not started and the step object runs the step again."""

    def __init__(self, eng, after_grids):
        self.eng, self.after_grids, self.hit = eng, after_grids, False

    def __call__(self):
        if self.eng.range_hit():
            self.hit = True
            return
        if self.after_grids is not None:
            self.after_grids()


def _add_regularisers(m, loss, grads, n_rays_global, weight_tv_density, tvs, dense_mode):
    """The trainers' ``do_tv`` lines (fine.py:383-400, lts.py:381-398, pdra.py:459-476: the same block in all three)."""
    from . import render_utils
    w = weight_tv_density * tvs["smooth_grad"]
    loss1 = loss.reshape(1)
    m.smooth_grad_tv_fwd(w, loss1)
    m.smooth_grad_tv_bwd(w, grads["sdf.grid"])
    wt = weight_tv_density * tvs["sdf"] / n_rays_global * max(m._world_size_l) / 128
    render_utils.total_variation_add_grad(m.sdf.grid.detach(), grads["sdf.grid"], wt, wt, wt, dense_mode)
    return loss


def _finish(step):
    """After the LAST step: the march-overflow flag of a data-parallel step is examined by the next step's
    ``_check_overflow``; this examines the final one (data parallel: every rank raises together).  (The split-fp16 kernels'
    range flag needs nothing here: every step examines and heals its own, _RangeGuard.)"""
    if step.pg is not None:
        _check_overflow(step)


def _cached_views(step, key):
    """The gradient views of the previous step when nothing they depend on changed (grid size, the flat buffer): the buffer is
    zeroed and a COPY of the dict handed out (callers replace entries).  Building ~45 views and asking the model for its
    parameters again cost the host ~0.1 ms per step, between the plan launch and the read-back it waits for."""
    v = getattr(step, "_views", None)
    if v is None or v[0] != key or v[1] is not step._flat or step._flat is None:
        return None
    step._flat.zero_()
    return dict(v[2])


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
        self._sync = None
        self._sync_mode = os.environ.get("ESR_GRAD_SYNC", "auto")
        _warn_hw_queues(process_group)

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
        """One flat zero buffer holding every gradient (a single memset; grids first, then the MLP tensors)."""
        m = self.model
        X, Y, Z = m._world_size_l                   # host copy: int(device scalar) is a sync each
        vkey = (X, Y, Z, id(getattr(self, "sharded", None)))      # (a ShardedGrids attached later changes the padding)
        hit = _cached_views(self, vkey)
        if hit is not None:
            return hit
        shapes = [("sdf.grid", (1, 1, X, Y, Z)), ("off_color.grid", (1, X, Y, Z, 6)),
                  ("emo_color.grid", (1, X, Y, Z, 6))]
        shapes += [(n, tuple(p.shape)) for n, p in zip(self._param_names(), m._mlp_params())]
        n_grid = sum(int(torch.Size(s).numel()) for _, s in shapes[:3])
        pad = _grid_pad(self, n_grid) - n_grid
        total = sum(int(torch.Size(s).numel()) for _, s in shapes) + pad
        if self._flat is None or self._flat.numel() != total:
            self._flat = torch.empty(total, dtype=torch.float32, device=dev)
        self._flat.zero_()
        out, o = {}, 0
        for i, (n, s) in enumerate(shapes):
            k = int(torch.Size(s).numel())
            out[n] = self._flat[o:o + k].view(s)
            o += k
            if i == 2:
                self._n_grid = o          # [0, _n_grid): the three dense grids; [_n_grid, _n_grid_pad): zero padding
                o += pad                  # (shard mode); the rest: MLP tensors
                self._n_grid_pad = o
        self._views = (vkey, self._flat, dict(out))
        return out

    @torch.no_grad()
    def forward_loss_backward(self, batch: Dict[str, torch.Tensor], s_val: float,
                              global_rays: Optional[int] = None,
                              entropy_owner: bool = True,
                              regularisers: Optional[dict] = None) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        """``global_rays``: size of the whole batch when ``batch`` is one rank's shard.
        ``entropy_owner``: the reference's entropy term looks at the LAST ray of the batch only
        (fine.py:378), so under sharding exactly one rank -- the one holding the global last ray --
        must add it.
        ``regularisers``: the arguments of ``add_regularisers`` (dict n_rays_global, weight_tv_density, tvs, dense_mode) on an
        iteration whose ``do_tv`` lines run (fine.py:383-400).  The step then adds them itself, in the reference's order
        (after the backward's sums into ``sdf.grid``'s gradient), but as soon as the grid scatters are enqueued: the three
        dense-grid launches (0.13 ms at C2, no LDS, few registers) run BESIDE the weight-gradient kernels of the second
        stream instead of behind them.  Same launches, same order of additions: results are those of calling
        ``add_regularisers`` on the returned gradients, bit for bit.  (Data parallel: after the exchange, as before.)"""
        m = self.model
        eng = m.engine
        m.s_val = s_val
        ps = m._mlp_params()
        g = None
        with _OverflowScope(eng, self.pg is not None):
            return self._step(batch, s_val, global_rays, entropy_owner, m, eng, ps, regularisers)

    def close(self):
        """Call once after the final step (data parallel): reports a march overflow flagged by that step."""
        _finish(self)

    finish = close

    # `with FineStep(...) as step:` -- close() runs at exit, so the last step's deferred overflow flag cannot be dropped
    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        if et is None:
            self.close()
        return False

    def _step(self, batch, s_val, global_rays, entropy_owner, m, eng, ps, regularisers=None):
        res = self._attempt(batch, s_val, global_rays, entropy_owner, m, eng, ps, regularisers)
        if res is None:
            # a split-fp16 kernel of the forward raised the range flag (fine_engine.py): nothing has left the step -- no
            # exchange was started, no gradient handed out -- so the whole step runs again on the f32 MFMA kernels.  The
            # second attempt's launches are ordered behind the first's on every stream they share, and its prelude zeroes
            # the gradient buffer again.
            with eng.f32_only():
                res = self._attempt(batch, s_val, global_rays, entropy_owner, m, eng, ps, regularisers)
        return res

    def _attempt(self, batch, s_val, global_rays, entropy_owner, m, eng, ps, regularisers=None):
        g = None

        def prelude():        # independent of the march: runs on the device while the host waits for the plan header
            nonlocal g
            with eng.packing():       # the three nets' packs: one launch
                eng.pack("off", KIND_RADIANCE, list(ps[0:8:2]), list(ps[1:8:2]))
                eng.pack("emo", KIND_RADIANCE, list(ps[8:16:2]), list(ps[9:16:2]))
                eng.pack("tone", KIND_TONEMAP, list(ps[16:20:2]), list(ps[17:20:2]))
            g = self._alloc_grads(batch["rays_o"].device)

        ctx, last, srgb, lin = eng.forward(
            m.scene_struct(), batch["rays_o"], batch["rays_d"], batch["viewdirs"], batch["em_modes"],
            m.mask_cache.density.view(*m.mask_cache.density.shape[2:]),
            m.sdf.device_view(), m.off_color.device_view(), m.emo_color.device_view(), prelude=prelude, heal=False)
        m.last_counts = ctx.counts
        if self.pg is not None:
            _check_overflow(self)         # (after the forward's host wait: see _check_overflow)
        scale, w_ent = dp_loss_weights(last.shape[0], global_rays, entropy_owner, self.weight_entropy_last)
        loss, g_last, g_srgb, g_lin = eng.loss_fwd_bwd(last, srgb, lin, batch["rgbs"], self.white_bg,
                                                       self.weight_linear, w_ent, scale=scale)      # scaled in the kernel
        names = self._param_names()
        grads = dict(sdf=g["sdf.grid"], off_color=g["off_color.grid"], emo_color=g["emo_color.grid"],
                     off_w=[g[n] for n in names[0:8:2]], off_b=[g[n] for n in names[1:8:2]],
                     emo_w=[g[n] for n in names[8:16:2]], emo_b=[g[n] for n in names[9:16:2]],
                     tone_w=[g[n] for n in names[16:20:2]], tone_b=[g[n] for n in names[17:20:2]])
        works = []
        if self.pg is not None:
            import torch.distributed as dist
            # grid gradients (218 MB at C2, >99 % of the payload) are final when the engine calls this:
            # their exchange runs underneath the wgrad kernels
            after_grids, works = _grid_sync(self, eng)
        elif regularisers is not None:
            # the do_tv lines, behind the grid scatters on the main stream and beside the weight gradients of the second one
            # (a range fallback returns before this is called: _RangeGuard)
            after_grids = lambda: self.add_regularisers(loss, g, **regularisers)
        else:
            after_grids = None
        guard = _RangeGuard(eng, after_grids)
        eng.backward(ctx, g_last, g_srgb, g_lin, grads, after_grids=guard)
        if guard.hit:
            return None
        if self.pg is not None:
            works.append(dist.all_reduce(self._flat[self._n_grid_pad:], group=self.pg, async_op=True))
            lf = _reduce_loss_and_overflow(self, eng, loss, works)
            for w in works:
                w.wait()                  # stream-level wait: the caller's stream sees reduced gradients
            loss = lf[0:1].reshape(loss.shape)
            _publish_overflow(self, lf)
            if self._sync is not None:
                self._sync.verify()       # everything of the step is enqueued: close the brick exchange
            if regularisers is not None:
                self.add_regularisers(loss, g, **regularisers)
        g["off_color.grid"] = g["off_color.grid"].permute(0, 4, 1, 2, 3)     # logical [1,6,X,Y,Z]
        g["emo_color.grid"] = g["emo_color.grid"].permute(0, 4, 1, 2, 3)
        return loss, g

    @torch.no_grad()
    def add_regularisers(self, loss: torch.Tensor, grads: Dict[str, torch.Tensor], n_rays_global: int,
                         weight_tv_density: float, tvs: Dict[str, float], dense_mode: bool):
        """The ``do_tv`` lines of the trainer (fine.py:383-400, every ``tv_every``-th iteration) without autograd:
        ``loss += w * smoothed-gradient TV`` with its SDF gradient (csrc/tv.hip, fused forward + backward), then the
        in-place 6-neighbour TV gradient (``sdf_total_variation_add_grad``).  Dense-grid work on replicated
        parameters: under data parallelism call it AFTER the exchange, identically on every rank."""
        return _add_regularisers(self.model, loss, grads, n_rays_global, weight_tv_density, tvs, dense_mode)

    def assign_grads(self, grads: Dict[str, torch.Tensor]):
        """Expose the step's gradients as ``param.grad`` (for a torch optimizer)."""
        for n, p in self.model.named_parameters():
            if n in grads:
                p.grad = grads[n]


class LtsStep:
    """One training step of the LTS stage (``stage="lts"``, app/fine/lts.py:327-397) or the PDRA stage
    (``stage="pdra"``, app/fine/pdra.py:374-475) on the HIP path, without autograd in the loop:
    ``LtsEngine.lts_forward`` -> loss kernels (the fine-stage image loss plus one
    ``esr_pair_loss_fwd_bwd`` launch per LTS/PDRA term) -> ``LtsEngine.lts_backward``.

    Data parallelism as in ``FineStep``: every rank renders its own ray shard and its own
    ``num_ltspts`` surface points (per process, as in the reference), every loss term is scaled by
    ``n_local / n_global`` and ONE flat gradient buffer is summed over the ranks -- the dense-grid part
    asynchronously as soon as the scatters are done, underneath the weight-gradient kernels."""

    NETS = (("off_rgbnet", "linear"), ("emo_rgbnet", "linear"), ("tonemapper", "srgb"), ("brdfnet", "brdfnet"),
            ("emitnet", "brdfnet"))

    def __init__(self, model, trainer_cfg, stage: str = "lts", white_bg: bool = True, process_group=None,
                 split_points: bool = False):
        """``split_points``: under data parallelism split the reference's ``num_ltspts`` surface points over the ranks
        (``lts_point_share``: the remainder one point each to the first ranks, so the counts add up to ``num_ltspts``; the
        per-point loss terms are weighted by a rank's actual share) instead of drawing ``num_ltspts`` on each (the
        reference's setting is per process): the GLOBAL light-transport estimate then uses the reference's number of points
        and secondary rays, and the per-rank secondary work shrinks with G."""
        if stage not in ("lts", "pdra"):
            raise ValueError("stage must be 'lts' or 'pdra'")
        self.model, self.t, self.stage, self.white_bg, self.pg = model, trainer_cfg, stage, white_bg, process_group
        self.ltspts = int(model.num_ltspts)
        self.pt_scale = None                 # weight of this rank's per-point terms in the global mean (None: the ray share)
        if split_points and process_group is not None:
            import torch.distributed as dist
            self.ltspts, self.pt_scale = lts_point_share(self.ltspts, dist.get_world_size(process_group),
                                                         dist.get_rank(process_group))
        self._names = None
        self._flat = None
        self._sync = None
        self._sync_mode = os.environ.get("ESR_GRAD_SYNC", "auto")
        self._pair_jobs = []
        _warn_hw_queues(process_group)

    def close(self):
        """Call once after the final step (data parallel): reports a march overflow flagged by that step."""
        _finish(self)

    finish = close

    # `with FineStep(...) as step:` -- close() runs at exit, so the last step's deferred overflow flag cannot be dropped
    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        if et is None:
            self.close()
        return False

    def _param_names(self):
        if self._names is None:
            names = []
            for net, seq in self.NETS:
                for key, sub in getattr(getattr(self.model, net), seq).named_modules():
                    if isinstance(sub, torch.nn.Linear):
                        names += [f"{net}.{seq}.{key}.weight", f"{net}.{seq}.{key}.bias"]
            self._names = names
        return self._names

    def _alloc_grads(self, dev):
        m = self.model
        X, Y, Z = m._world_size_l                   # host copy: int(device scalar) is a sync each
        vkey = (X, Y, Z, id(getattr(self, "sharded", None)))      # (a ShardedGrids attached later changes the padding)
        hit = _cached_views(self, vkey)
        if hit is not None:
            return hit
        J = m.envmap.mus.shape[0]
        shapes = [("sdf.grid", (1, 1, X, Y, Z)), ("off_color.grid", (1, X, Y, Z, 6)), ("emo_color.grid", (1, X, Y, Z, 6)),
                  ("brdf.grid", (1, X, Y, Z, 6))]
        shapes += [(n, tuple(p.shape)) for n, p in zip(self._param_names(), m._mlp_params())]
        shapes += [("envmap.mus", (J, 3)), ("envmap.lambdas", (J, 1)), ("envmap.lobes", (J, 3))]
        n_grid = sum(int(torch.Size(s).numel()) for _, s in shapes[:4])
        pad = _grid_pad(self, n_grid) - n_grid
        total = sum(int(torch.Size(s).numel()) for _, s in shapes) + pad
        if self._flat is None or self._flat.numel() != total:
            self._flat = torch.empty(total, dtype=torch.float32, device=dev)
        self._flat.zero_()
        out, o = {}, 0
        for i, (n, s) in enumerate(shapes):
            k = int(torch.Size(s).numel())
            out[n] = self._flat[o:o + k].view(s)
            o += k
            if i == 3:
                self._n_grid = o          # [0, _n_grid): the four dense grids; [_n_grid, _n_grid_pad): zero padding
                o += pad                  # (shard mode); the rest: MLP + env-map tensors
                self._n_grid_pad = o
        self._views = (vkey, self._flat, dict(out))
        return out

    def _pair(self, eng, loss, a, b, kind, w_value, w_a, w_b, scale, want_gb=True, row_mask=None, mask_value=0,
              count=None):
        a = a.contiguous()
        ga = torch.empty_like(a)
        gb = torch.empty_like(a) if (b is not None and want_gb) else None
        rows = a.shape[0]
        cols = a.numel() // max(rows, 1)
        if rows:
            # collected; ONE launch for all terms of the step (_pair_flush -> esr_pair_loss_batch)
            b = b.contiguous() if b is not None else None
            self._pair_jobs.append((a, b, rows, cols, row_mask, mask_value, count, kind, w_value * scale, w_a * scale,
                                    w_b * scale, ga, gb))
        return ga, gb

    def _pair_flush(self, eng, loss):
        import ctypes as C
        from . import _lib
        jobs, self._pair_jobs = self._pair_jobs, []
        if not jobs:
            return
        arr = (_lib.EsrPairJob * len(jobs))()
        for jb, (a, b, rows, cols, row_mask, mask_value, count, kind, wv, wa, wb, ga, gb) in zip(arr, jobs):
            jb.a, jb.b = a.data_ptr(), (b.data_ptr() if b is not None else None)
            jb.rows, jb.cols = rows, cols
            jb.row_mask = row_mask.data_ptr() if row_mask is not None else None
            jb.mask_value = mask_value
            jb.count_dev = count.data_ptr() if count is not None else None
            jb.kind, jb.w_value, jb.w_a, jb.w_b = kind, wv, wa, wb
            jb.ga, jb.gb = ga.data_ptr(), (gb.data_ptr() if gb is not None else None)
        eng._run("pair_loss", eng.L.esr_pair_loss_batch, arr, len(jobs), _lib.ptr(loss), eng._s())

    @torch.no_grad()
    def forward_loss_backward(self, batch: Dict[str, torch.Tensor], s_val: float, global_rays: Optional[int] = None,
                              entropy_owner: bool = True, draws=None, regularisers: Optional[dict] = None):
        """``regularisers``: as ``FineStep.forward_loss_backward`` -- the arguments of ``add_regularisers`` on an iteration whose
        ``do_tv`` lines run (lts.py:381-398, pdra.py:459-476); launched behind the grid scatters, beside the last
        weight-gradient jobs."""
        m, t = self.model, self.t
        eng = m.engine
        m.s_val = s_val
        ps = m._mlp_params()
        with _OverflowScope(eng, self.pg is not None):
            return self._step(batch, s_val, global_rays, entropy_owner, draws, m, t, eng, ps, regularisers)

    def add_regularisers(self, loss: torch.Tensor, grads: Dict[str, torch.Tensor], n_rays_global: int,
                         weight_tv_density: float, tvs: Dict[str, float], dense_mode: bool):
        """The ``do_tv`` lines of the LTS / PDRA trainers (lts.py:381-398, pdra.py:459-476): ``FineStep.add_regularisers``."""
        return _add_regularisers(self.model, loss, grads, n_rays_global, weight_tv_density, tvs, dense_mode)

    def _step(self, batch, s_val, global_rays, entropy_owner, draws, m, t, eng, ps, regularisers=None):
        res = self._attempt(batch, s_val, global_rays, entropy_owner, draws, m, t, eng, ps, regularisers)
        if res is None:
            # the range fallback (FineStep._step): again on the f32 MFMA kernels, with the SAME random draws
            with eng.f32_only():
                res = self._attempt(batch, s_val, global_rays, entropy_owner, eng.last_draws, m, t, eng, ps, regularisers)
        return res

    def _attempt(self, batch, s_val, global_rays, entropy_owner, draws, m, t, eng, ps, regularisers=None):
        from .fine_engine import KIND_RADIANCE as KR, KIND_TONEMAP as KT
        from .lts_engine import KIND_BRDF as KB, KIND_EMIT as KE
        G = None

        def prelude():        # independent of the march: on a side stream while the host waits for the plan header
            nonlocal G
            o = 0
            with eng.packing():       # the five nets' packs: one launch
                for name, kind, n in (("off", KR, 8), ("emo", KR, 8), ("tone", KT, 4), ("brdf", KB, 8), ("emit", KE, 8)):
                    eng.pack(name, kind, list(ps[o:o + n:2]), list(ps[o + 1:o + n:2]))
                    o += n
            G = self._alloc_grads(batch["rays_o"].device)
        grids = dict(sdf=m.sdf.device_view(), off=m.off_color.device_view(), emo=m.emo_color.device_view(),
                     brdf=m.brdf.device_view(), mask=m.mask_cache.density.view(*m.mask_cache.density.shape[2:]))
        env = dict(mus=m.envmap.mus.detach(), lambdas=m.envmap.lambdas.detach(), lobes=m.envmap.lobes.detach())
        pdra = self.stage == "pdra"
        cfg = dict(num_2ndrays=m.num_2ndrays, num_ltspts=self.ltspts, normal_eps=t.normal_eps, emit_eps=t.emit_eps,
                   pdra=m.pdra_mode, eps_grads=pdra)
        ctx, out = eng.lts_forward(m.scene_struct(), m.scene_struct(near=m.lts_near), batch, grids, env, cfg, draws,
                                   prelude=prelude)
        m.last_counts = dict(eng.prim.counts)
        if self.pg is not None:
            _check_overflow(self)         # (after the forward's host waits: see _check_overflow)
        last = out["etc/alphainv_cum"]
        scale, w_ent = dp_loss_weights(last.shape[0], global_rays, entropy_owner, t.weight_entropy_last)
        loss, g_last, g_srgb, g_lin = eng.loss_fwd_bwd(last, out["srgb/rgb"], out["lin/rgb"], batch["rgbs"],
                                                       self.white_bg, t.weight_linear, w_ent, scale=scale)
        g = {"etc/alphainv_cum": g_last, "srgb/rgb": g_srgb, "lin/rgb": g_lin}
        wl = t.weight_lts
        pscale = scale if self.pt_scale is None else self.pt_scale       # the terms that are means over the surface POINTS
        if not pdra:
            g["lin/pbr/off"], g["lin/pbr/off_hat"] = self._pair(eng, loss, out["lin/pbr/off"], out["lin/pbr/off_hat"],
                                                                0, wl, wl, wl, pscale)
            g["lin/pbr/emo"], g["lin/pbr/emo_hat"] = self._pair(eng, loss, out["lin/pbr/emo"], out["lin/pbr/emo_hat"],
                                                                0, wl, wl, wl, pscale)
        else:
            g["lin/pbr/off"], g["lin/pbr/off_hat"] = self._pair(eng, loss, out["lin/pbr/off"], out["lin/pbr/off_hat"],
                                                                1, wl, wl, wl, pscale)
            wL, wR = t.weight_lts_l, t.weight_lts_r
            g["lin/pbr/emo"], g["lin/pbr/emo_hat"] = self._pair(eng, loss, out["lin/pbr/emo"], out["lin/pbr/emo_hat"],
                                                                1, wl * (wL + wR), wl * wR, wl * wL, pscale)
            um8 = batch["uncert_masks"].view(torch.uint8)
            n_cert = (um8 == 0).sum(dtype=torch.int32).view(1)
            g["emit_marched"], _ = self._pair(eng, loss, out["emit_marched"], None, 0, t.weight_emit_supp,
                                              t.weight_emit_supp, 0.0, scale, row_mask=um8, mask_value=0, count=n_cert)
            ws = t.weight_emit_smooth
            g["etc/emit"], g["etc/emit_eps"] = self._pair(eng, loss, out["etc/emit"], out["etc/emit_eps"], 1, ws, ws, ws,
                                                          scale)
        wn = t.weight_normal_smooth
        g["etc/normal"], g["etc/normal_eps"] = self._pair(eng, loss, out["etc/normal"], out["etc/normal_eps"], 1, wn, wn,
                                                          wn, scale)
        self._pair_flush(eng, loss)       # the step's two-operand loss terms: one launch
        if G is None:                     # (degenerate: the engine did not run the prelude)
            G = self._alloc_grads(last.device)
        names = self._param_names()
        pick = lambda lo, hi: ([G[n] for n in names[lo:hi:2]], [G[n] for n in names[lo + 1:hi:2]])
        (ow, ob), (ew, eb), (tw, tb), (bw, bb), (mw, mb) = pick(0, 8), pick(8, 16), pick(16, 20), pick(20, 28), pick(28, 36)
        grads = dict(sdf=G["sdf.grid"], off=G["off_color.grid"], emo=G["emo_color.grid"], brdf=G["brdf.grid"],
                     off_w=ow, off_b=ob, emo_w=ew, emo_b=eb, tone_w=tw, tone_b=tb, brdf_w=bw, brdf_b=bb,
                     emit_w=mw, emit_b=mb, mus=G["envmap.mus"], lambdas=G["envmap.lambdas"], lobes=G["envmap.lobes"])
        works = []
        if self.pg is not None:
            import torch.distributed as dist
            # the four grid gradients (> 99 % of the payload) are final when the engine calls this: their exchange
            # runs underneath the weight-gradient kernels
            after_grids, works = _grid_sync(self, eng)
        elif regularisers is not None:
            after_grids = lambda: self.add_regularisers(loss, G, **regularisers)      # (FineStep._attempt)
        else:
            after_grids = None
        guard = _RangeGuard(eng, after_grids)
        eng.lts_backward(ctx, g, grads, after_grids=guard)
        if guard.hit:
            self._pair_jobs = []
            return None
        if self.pg is not None:
            works.append(dist.all_reduce(self._flat[self._n_grid_pad:], group=self.pg, async_op=True))
            lf = _reduce_loss_and_overflow(self, eng, loss, works)
            for w in works:
                w.wait()
            loss = lf[0:1].reshape(loss.shape)
            _publish_overflow(self, lf)
            if self._sync is not None:
                self._sync.verify()       # everything of the step is enqueued: close the brick exchange
            if regularisers is not None:
                self.add_regularisers(loss, G, **regularisers)
        for k in ("off_color.grid", "emo_color.grid", "brdf.grid"):
            G[k] = G[k].permute(0, 4, 1, 2, 3)
        return loss, G, out

    def assign_grads(self, grads: Dict[str, torch.Tensor]):
        for n, p in self.model.named_parameters():
            if n in grads:
                p.grad = grads[n]


class FinetuneStep:
    """One step of the re-lighting fine-tune (app/fine/pdra.py:1047-1109: ``forward_finetune`` + ``0.5 * mse(emo, emo_hat)`` +
    backward) on the HIP path without autograd in the loop: ``LtsEngine.finetune_forward`` -> one ``esr_pair_loss_batch``
    launch -> ``LtsEngine.finetune_backward``.  Only ``emo_color.grid`` and the emo radiance net receive gradients (the
    reference freezes everything else, pdra.py:1060-1066); the model must be in fine-tune mode (``model.train(True,
    finetune=True)``: the frozen ``emit_color`` copy exists).  The drop-in route -- ``ESRNeRF.forward_finetune`` + a torch loss +
    ``loss.backward()`` -- enqueues the same kernels and is tested to give the same numbers; this driver spares the step
    autograd's bookkeeping and a dozen torch launches (it was the most host-bound step of the path)."""

    def __init__(self, model, weight: float = 0.5):
        if not hasattr(model, "emit_color"):
            raise RuntimeError("FinetuneStep needs the model in fine-tune mode: model.train(True, finetune=True)")
        self.model, self.weight = model, float(weight)
        self._flat = None
        self._names = [f"emo_rgbnet.linear.{k}.{p}" for k, sub in model.emo_rgbnet.linear.named_modules()
                       if isinstance(sub, torch.nn.Linear) for p in ("weight", "bias")]
        self._pair_jobs = []

    # the two-operand loss helpers of LtsStep, as methods of this class (class attributes, not bound methods stored on the
    # instance: that would make every step object a reference cycle holding its gradient buffer -- modules.ForwardSwitch)
    _pair = LtsStep._pair
    _pair_flush = LtsStep._pair_flush

    def _alloc_grads(self, dev):
        m = self.model
        X, Y, Z = m._world_size_l
        ps = [t for lin in m.emo_rgbnet.layers() for t in (lin.weight, lin.bias)]
        shapes = [("emo_color.grid", (1, X, Y, Z, 6))] + [(n, tuple(p.shape)) for n, p in zip(self._names, ps)]
        total = sum(int(torch.Size(s).numel()) for _, s in shapes)
        if self._flat is None or self._flat.numel() != total:
            self._flat = torch.empty(total, dtype=torch.float32, device=dev)
        self._flat.zero_()
        out, o = {}, 0
        for n, s in shapes:
            k = int(torch.Size(s).numel())
            out[n] = self._flat[o:o + k].view(s)
            o += k
        return out

    @torch.no_grad()
    def forward_loss_backward(self, batch: Dict[str, torch.Tensor], s_val: float, draws=None):
        """batch: rays_o, rays_d, viewdirs, em_modes, em_intensities, em_colors.  -> (loss [1], gradients by parameter name)."""
        m = self.model
        eng = m.engine
        m.s_val = s_val
        res = self._attempt(batch, draws, m, eng)
        if res is None:                   # the split-fp16 kernels' range fallback (FineStep._step): again, same draws
            with eng.f32_only():
                res = self._attempt(batch, eng.last_draws, m, eng)
        return res

    def _attempt(self, batch, draws, m, eng):
        from .fine_engine import KIND_RADIANCE as KR
        from .lts_engine import KIND_BRDF as KB, KIND_EMIT as KE
        dev = batch["rays_o"].device
        emo = m.emo_rgbnet.layers()
        with eng.packing():
            eng.pack("emo", KR, [l.weight.detach() for l in emo], [l.bias.detach() for l in emo])
            for name, kind, net in (("brdf", KB, m.brdfnet), ("emit", KE, m.emitnet)):
                lins = net.layers()
                eng.pack(name, kind, [l.weight.detach() for l in lins], [l.bias.detach() for l in lins])
        G = self._alloc_grads(dev)
        grids = dict(sdf=m.sdf.device_view(), emo=m.emo_color.device_view(), brdf=m.brdf.device_view(),
                     emit=m.emit_color.device_view(), mask=m.mask_cache.density.view(*m.mask_cache.density.shape[2:]))
        cfg = dict(num_2ndrays=m.num_2ndrays, num_ltspts=m.num_ltspts)
        b = {k: batch[k].contiguous() for k in ("rays_o", "rays_d", "viewdirs", "em_modes", "em_intensities", "em_colors")}
        ctx, out = eng.finetune_forward(m.scene_struct(), m.scene_struct(near=m.lts_near), b, grids, cfg, draws)
        m.last_counts = dict(eng.prim.counts)
        loss = torch.zeros(1, dtype=torch.float32, device=dev)
        g_emo, _ = self._pair(eng, loss, out["lin/pbr/emo"], out["lin/pbr/emo_hat"], 0, self.weight, self.weight, 0.0, 1.0,
                              want_gb=False)
        self._pair_flush(eng, loss)
        eng.finetune_backward(ctx, g_emo, dict(emo=G["emo_color.grid"], emo_w=[G[n] for n in self._names[0::2]],
                                               emo_b=[G[n] for n in self._names[1::2]]))
        if eng.range_hit():               # (with the backward queued: the wait is free; nothing has left the step -- _RangeGuard)
            return None
        G["emo_color.grid"] = G["emo_color.grid"].permute(0, 4, 1, 2, 3)       # logical [1,6,X,Y,Z]
        return loss, G

    def assign_grads(self, grads: Dict[str, torch.Tensor]):
        for n, p in self.model.named_parameters():
            if n in grads:
                p.grad = grads[n]

    def close(self):
        pass
