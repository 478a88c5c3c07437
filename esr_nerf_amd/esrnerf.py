"""Drop-in for the reference's LTS / PDRA renderer ``app.fine.model.ESRNeRF``
(reference: app/fine/model/esrnerf.py) -- training forward only this round.

Same constructor, ``train()`` / ``forward(**kwargs)`` protocol, the 16 result keys of
``forward_training`` (esrnerf.py:832-851), sub-module names (``brdf``, ``brdfnet``,
``emitnet``, ``envmap`` join the fine-stage ones; the reference's optimizer addresses them by
name, cfg/app/lts.yaml:61-71) and ``state_dict`` keys.  ``forward_training`` is one autograd
node over the kernels of libesr_hip.so (esr_nerf_amd/lts_engine.py).

``forward_finetune`` (re-lighting fine-tune target, esrnerf.py:241-484) runs on the same kernels.
``eval_emit`` / ``eval_esp`` (the PDRA trainer's regrouping queries) are forward-only passes over the same kernels.
``forward_evaluate`` renders images incl. the per-sample light-transport decomposition.  ``render_envmap`` / ``extract_geometry`` are torch utilities outside the path.
"""
from __future__ import annotations

from typing import List

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .fine_engine import KIND_RADIANCE, KIND_TONEMAP
from .lts_engine import KIND_BRDF, KIND_EMIT, LtsEngine
from .modules import DenseGrid, _mlp_stack, _linears
from .voxurff import VoxurfF

OUT_KEYS = ("etc/alphainv_cum", "srgb/rgb", "lin/rgb", "lin/pbr/off", "lin/pbr/off_hat", "lin/pbr/emo",
            "lin/pbr/emo_hat", "emit_marched", "etc/normal", "etc/normal_eps", "etc/emit", "etc/emit_eps",
            "etc/brdf", "etc/brdf_eps")


class BRDFNet(nn.Module):
    """76 -> width x (depth-1) -> 5 (sigmoid; basecolor 3, roughness 1, metallic 1); key prefix brdfnet.brdfnet."""

    def __init__(self, inputdim, width, depth, mode=None):
        super().__init__()
        self.brdfnet = _mlp_stack(inputdim, width, depth, 5)
        nn.init.constant_(self.brdfnet[-1].bias, 0)

    def layers(self):
        return _linears(self.brdfnet)

    def forward(self, x):
        return torch.sigmoid(self.brdfnet(x)).split([3, 1, 1], -1)


class EmissionNet(nn.Module):
    """76 -> width x (depth-1) -> 3 (softplus); key prefix emitnet.brdfnet (sic, as in the reference)."""

    def __init__(self, inputdim, width, depth):
        super().__init__()
        self.brdfnet = _mlp_stack(inputdim, width, depth, 3)
        nn.init.constant_(self.brdfnet[-1].bias, 0)

    def layers(self):
        return _linears(self.brdfnet)

    def forward(self, x):
        return F.softplus(self.brdfnet(x))


class SphericalGaussian(nn.Module):
    """48-lobe environment map; initialisation restated from app/utils/pbr/module.py:86-131
    (softplus activation: mus = softplus^-1 of energy-normalised amplitudes)."""

    def __init__(self, num_sg: int = 48, activation: str = "softplus"):
        super().__init__()
        if activation != "softplus":
            raise NotImplementedError("libesr_hip implements the softplus environment map (cfg/app/lts.yaml:29)")
        mus = torch.randn(num_sg, 3)
        lambdas = 10.0 + torch.abs(torch.randn(num_sg, 1) * 20.0)
        lobes = torch.randn(num_sg, 3)
        lam = torch.abs(lambdas)
        energy = F.softplus(mus) * 2.0 * torch.pi / lam * (1.0 - torch.exp(-2.0 * lam))
        normalized = F.softplus(mus) / torch.sum(energy, dim=0, keepdim=True) * 2.0 * torch.pi * 0.8
        self.mus = nn.Parameter(torch.log(torch.exp(normalized) - 1.0))
        self.lambdas = nn.Parameter(lambdas)
        self.lobes = nn.Parameter(lobes)

    def forward(self, dirs):
        lobes = F.normalize(self.lobes, dim=-1)
        e = torch.exp(self.lambdas.abs() * ((dirs.unsqueeze(-2) * lobes).sum(-1, keepdim=True) - 1.0))
        return F.softplus((self.mus * e).sum(-2))


class _LtsRender(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, batch, draws, sdf, off_color, emo_color, brdf, mus, lambdas, lobes, *mlp_params):
        eng: LtsEngine = model.engine
        nets = (("off", KIND_RADIANCE, 8), ("emo", KIND_RADIANCE, 8), ("tone", KIND_TONEMAP, 4),
                ("brdf", KIND_BRDF, 8), ("emit", KIND_EMIT, 8))
        def pack():
            o = 0
            with eng.packing():
                for name, kind, n in nets:
                    ps = mlp_params[o:o + n]
                    eng.pack(name, kind, list(ps[0::2]), list(ps[1::2]))
                    o += n
        pack()
        scene = model.scene_struct()
        scene2 = model.scene_struct(near=model.lts_near)
        grids = dict(sdf=model.sdf.device_view(), off=model.off_color.device_view(),
                     emo=model.emo_color.device_view(), brdf=model.brdf.device_view(),
                     mask=model.mask_cache.density.view(*model.mask_cache.density.shape[2:]))
        env = dict(mus=mus.detach(), lambdas=lambdas.detach(), lobes=lobes.detach())
        cfg = dict(num_2ndrays=model.num_2ndrays, num_ltspts=model.num_ltspts, normal_eps=batch["normal_eps"],
                   emit_eps=batch["emit_eps"], pdra=model.pdra_mode)
        lctx, out = eng.lts_forward(scene, scene2, batch, grids, env, cfg, draws)
        if eng.range_hit():       # a split-fp16 kernel's range flag: again on the f32 MFMA kernels, same draws (fine_engine.py)
            with eng.f32_only():
                pack()
                lctx, out = eng.lts_forward(scene, scene2, batch, grids, env, cfg, eng.last_draws)
        ctx.lctx, ctx.model = lctx, model
        ctx.set_materialize_grads(False)         # unused result tensors arrive as None and cost nothing
        ctx.shapes = [tuple(p.shape) for p in mlp_params]
        model.last_counts = dict(eng.prim.counts)
        # (fresh tensor objects: voxurff._FineRender.forward)
        return tuple(out[k].detach() for k in OUT_KEYS)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gout):
        model = ctx.model
        dev = model.sdf.grid.device
        z = lambda shape: torch.zeros(shape, dtype=torch.float32, device=dev)
        X, Y, Z = model._world_size_l           # host copy: int(device scalar) is a sync each
        g_sdf = z((1, 1, X, Y, Z))
        g_off, g_emo, g_brdf = z((1, X, Y, Z, 6)), z((1, X, Y, Z, 6)), z((1, X, Y, Z, 6))
        mg = [z(s) for s in ctx.shapes]
        J = model.envmap.mus.shape[0]
        grads = dict(sdf=g_sdf, off=g_off, emo=g_emo, brdf=g_brdf, mus=z((J, 3)), lambdas=z((J, 1)), lobes=z((J, 3)),
                     off_w=mg[0:8:2], off_b=mg[1:8:2], emo_w=mg[8:16:2], emo_b=mg[9:16:2],
                     tone_w=mg[16:20:2], tone_b=mg[17:20:2], brdf_w=mg[20:28:2], brdf_b=mg[21:28:2],
                     emit_w=mg[28:36:2], emit_b=mg[29:36:2])
        g = dict(zip(OUT_KEYS, gout))
        model.engine.lts_backward(ctx.lctx, g, grads)
        perm5 = lambda t: t.permute(0, 4, 1, 2, 3)
        return (None, None, None, g_sdf, perm5(g_off), perm5(g_emo), perm5(g_brdf), grads["mus"], grads["lambdas"],
                grads["lobes"], *mg)


class _FinetuneRender(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, batch, draws, emo_color, *emo_params):
        eng: LtsEngine = model.engine
        def pack():
            with eng.packing():
                eng.pack("emo", KIND_RADIANCE, list(emo_params[0::2]), list(emo_params[1::2]))
                for name, kind, net in (("brdf", KIND_BRDF, model.brdfnet), ("emit", KIND_EMIT, model.emitnet)):
                    lins = net.layers()
                    eng.pack(name, kind, [l.weight.detach() for l in lins], [l.bias.detach() for l in lins])
        pack()
        grids = dict(sdf=model.sdf.device_view(), emo=model.emo_color.device_view(), brdf=model.brdf.device_view(),
                     emit=model.emit_color.device_view(),
                     mask=model.mask_cache.density.view(*model.mask_cache.density.shape[2:]))
        cfg = dict(num_2ndrays=model.num_2ndrays, num_ltspts=model.num_ltspts)
        fctx, out = eng.finetune_forward(model.scene_struct(), model.scene_struct(near=model.lts_near), batch, grids,
                                         cfg, draws)
        if eng.range_hit():       # (the range fallback, as in _LtsRender)
            with eng.f32_only():
                pack()
                fctx, out = eng.finetune_forward(model.scene_struct(), model.scene_struct(near=model.lts_near), batch, grids,
                                                 cfg, eng.last_draws)
        ctx.fctx, ctx.model = fctx, model
        ctx.shapes = [tuple(p.shape) for p in emo_params]
        ctx.set_materialize_grads(False)
        model.last_counts = dict(eng.prim.counts)
        emo, emo_hat = out["lin/pbr/emo"].detach(), out["lin/pbr/emo_hat"].detach()    # (fresh objects: voxurff._FineRender.forward)
        ctx.mark_non_differentiable(emo_hat)
        return emo, emo_hat

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_emo, _g_hat):
        model = ctx.model
        dev = model.sdf.grid.device
        X, Y, Z = model._world_size_l           # host copy: int(device scalar) is a sync each
        g_grid = torch.zeros((1, X, Y, Z, 6), dtype=torch.float32, device=dev)
        mg = [torch.zeros(s, dtype=torch.float32, device=dev) for s in ctx.shapes]
        if g_emo is not None:
            model.engine.finetune_backward(ctx.fctx, g_emo.contiguous(), dict(emo=g_grid, emo_w=mg[0::2], emo_b=mg[1::2]))
        return (None, None, None, g_grid.permute(0, 4, 1, 2, 3), *mg)


class ESRNeRF(VoxurfF):
    def __init__(self, cfg, near, far, xyz_min, xyz_max, mask_xyz_min, mask_xyz_max, mask_alpha_init,
                 mask_density, s_val, num_voxles):
        # construction order reproduces the reference's parameter creation (and RNG draw) order:
        # the fine-stage members first, then brdf grid, emitnet, brdfnet, envmap (esrnerf.py:106-195)
        super().__init__(cfg, near, far, xyz_min, xyz_max, mask_xyz_min, mask_xyz_max, mask_alpha_init,
                         mask_density, s_val, num_voxles)
        m = cfg.app.model
        self.brdfnet_width, self.brdfnet_depth = m.brdfnet_width, m.brdfnet_depth
        self.env_sg, self.env_activation = m.env_sg, m.env_activation
        self.ray_sampling = m.ray_sampling
        self.num_2ndrays, self.num_ltspts, self.lts_near = m.num_2ndrays, m.num_ltspts, m.lts_near
        if self.brdfnet_width != 128 or self.brdfnet_depth != 4:
            raise NotImplementedError("libesr_hip kernels are built for brdfnet 128 x 4 (cfg/app/lts.yaml:25-26)")
        # the reference's spellings (esrnerf.py:188-192): 'random' | 'rand' and 'fib' | 'fibo' | 'fibonacci'
        alias = {"random": "random", "rand": "random", "fib": "fib", "fibo": "fib", "fibonacci": "fib"}
        if str(self.ray_sampling).lower() not in alias:
            raise ValueError(f"ray_sampling must be one of {sorted(alias)}, got {self.ray_sampling!r}")
        self._ray_sampling_mode = alias[str(self.ray_sampling).lower()]
        grid_args = dict(world_size=self.world_size, xyz_min=self.xyz_min, xyz_max=self.xyz_max)
        self.brdf = DenseGrid(channels=self.color_dim, **grid_args)
        dim0 = (3 + 3 * self.posbase_pe * 2) + self.color_dim + len(self.grad_feat) * 9 + 1
        self.emitnet = EmissionNet(dim0, self.brdfnet_width, self.brdfnet_depth)
        self.brdfnet = BRDFNet(dim0, self.brdfnet_width, self.brdfnet_depth)
        self.envmap = SphericalGaussian(self.env_sg, self.env_activation)
        self.pdra_mode = False
        self.to(self.device)

    @property
    def engine(self) -> LtsEngine:
        if self._engine is None:
            if not str(self.device).startswith("cuda"):
                raise RuntimeError("ESRNeRF.forward_training runs on libesr_hip.so and needs a GPU device "
                                   "(there is no CPU fallback)")
            self._engine = LtsEngine(self.device, getattr(self, "mlp_dtype", "f32"))
            self._engine.ray_sampling = self._ray_sampling_mode
            self._engine.neus_grad = self.neus_alpha == "grad"
        return self._engine

    def scene_struct(self, near=None):
        sc = super().scene_struct()
        if near is not None:
            sc.near_ = float(near)
        return sc

    def _mlp_params(self) -> List[torch.Tensor]:
        ps = []
        for net in (self.off_rgbnet, self.emo_rgbnet, self.tonemapper, self.brdfnet, self.emitnet):
            for lin in net.layers():
                ps += [lin.weight, lin.bias]
        return ps

    def forward_training(self, draws=None, **kwargs):
        """``draws`` (optional, not part of the reference signature): dict idx / dirs / noise_normal /
        noise_emit to replace the internal random draws (parity tests)."""
        self.s_val = kwargs["s_val"]
        batch = dict(rays_o=kwargs["rays_o"].contiguous(), rays_d=kwargs["rays_d"].contiguous(),
                     viewdirs=kwargs["viewdirs"].contiguous(), em_modes=kwargs["em_modes"].contiguous(),
                     uncert_masks=kwargs["uncert_masks"], normal_eps=kwargs["normal_eps"],
                     emit_eps=kwargs["emit_eps"])
        outs = _LtsRender.apply(self, batch, draws, self.sdf.grid, self.off_color.grid, self.emo_color.grid,
                                self.brdf.grid, self.envmap.mus, self.envmap.lambdas, self.envmap.lobes,
                                *self._mlp_params())
        r = dict(zip(OUT_KEYS, outs))
        um = batch["uncert_masks"]
        em = r.pop("emit_marched")
        return {
            "etc/alphainv_cum": r["etc/alphainv_cum"], "etc/white_bg": r["etc/alphainv_cum"][..., None],
            "srgb/rgb": r["srgb/rgb"], "lin/rgb": r["lin/rgb"],
            "lin/pbr/off": r["lin/pbr/off"], "lin/pbr/off_hat": r["lin/pbr/off_hat"],
            "lin/pbr/emo": r["lin/pbr/emo"], "lin/pbr/emo_hat": r["lin/pbr/emo_hat"],
            "etc/emit_uncert": em[um], "etc/emit_cert": em[~um],
            "etc/normal": r["etc/normal"], "etc/normal_eps": r["etc/normal_eps"],
            "etc/emit": r["etc/emit"], "etc/emit_eps": r["etc/emit_eps"],
            "etc/brdf": r["etc/brdf"], "etc/brdf_eps": r["etc/brdf_eps"],
        }

    def forward_finetune(self, draws=None, **kwargs):
        """Re-lighting fine-tune target (esrnerf.py:241-484): {"lin/pbr/emo" (differentiable w.r.t. emo_color and
        emo_rgbnet only), "lin/pbr/emo_hat" (constant)}.  Uses ``self.s_val`` like the reference."""
        batch = {k: kwargs[k].contiguous() for k in ("rays_o", "rays_d", "viewdirs", "em_modes", "em_intensities",
                                                      "em_colors")}
        ps = [t for lin in self.emo_rgbnet.layers() for t in (lin.weight, lin.bias)]
        emo, emo_hat = _FinetuneRender.apply(self, batch, draws, self.emo_color.grid, *ps)
        return {"lin/pbr/emo": emo, "lin/pbr/emo_hat": emo_hat}

    def train(self, mode=True, finetune=False):
        """esrnerf.py:218-239: fine-tune mode renders through ``forward_finetune`` and freezes a copy of the emo
        colour grid (``emit_color``) for the emission head; plain training drops the copy again; evaluation aliases
        ``emit_color`` to ``emo_color`` when no copy exists."""
        if mode and finetune:
            ret = nn.Module.train(self, mode)
            self.forward = self.forward_finetune
            self.emit_color = DenseGrid(channels=self.color_dim, world_size=self.world_size, xyz_min=self.xyz_min,
                                        xyz_max=self.xyz_max).to(self.device)
            self.emit_color.load_state_dict(self.emo_color.state_dict())
            for p in self.emit_color.parameters():
                p.requires_grad_(False)
            return ret
        if mode and hasattr(self, "emit_color"):
            del self.emit_color
        ret = super().train(mode)
        if not mode and not hasattr(self, "emit_color"):
            self.emit_color = self.emo_color
        return ret

    def _eval_query(self, what, **kwargs):
        eng = self.engine
        if what == "emit":
            lins = self.emitnet.layers()
            eng.pack("emit", KIND_EMIT, [l.weight.detach() for l in lins], [l.bias.detach() for l in lins])
        grid = getattr(self, "emit_color", self.emo_color)
        return eng.eval_query(self.scene_struct(), kwargs["rays_o"].contiguous(), kwargs["rays_d"].contiguous(),
                              kwargs["viewdirs"].contiguous(),
                              self.mask_cache.density.view(*self.mask_cache.density.shape[2:]), self.sdf.device_view(),
                              grid.device_view(), what)

    @torch.no_grad()
    def eval_emit(self, **kwargs):
        """Composited emission per ray (esrnerf.py:1299-1358; PDRA's ray regrouping, pdra.py:882-932)."""
        return self._eval_query("emit", **kwargs)

    @torch.no_grad()
    def eval_esp(self, **kwargs):
        """Weight-composited sample position per ray (esrnerf.py:1360-1407)."""
        return self._eval_query("esp", **kwargs)

    @torch.no_grad()
    def forward_evaluate(self, draws=None, **kwargs):
        """Image rendering (esrnerf.py:853-1297): kwargs rays_o, rays_d, viewdirs [N,3], em_modes (one scalar), pos_rt
        [3,3], render_pbr (bool), chunk_sz (samples per light-transport chunk); uses ``self.s_val``.  Returns the
        reference's 16 result keys, 21 with ``render_pbr``.  ``draws`` (optional): list of [chunk, num_2ndrays, 3]
        standard-normal tensors replacing the internal scattering draws (parity tests)."""
        eng = self.engine
        for name, kind, net in (("off", KIND_RADIANCE, self.off_rgbnet), ("emo", KIND_RADIANCE, self.emo_rgbnet),
                                ("tone", KIND_TONEMAP, self.tonemapper), ("brdf", KIND_BRDF, self.brdfnet),
                                ("emit", KIND_EMIT, self.emitnet)):
            lins = net.layers()
            eng.pack(name, kind, [l.weight.detach() for l in lins], [l.bias.detach() for l in lins])
        emit_grid = getattr(self, "emit_color", self.emo_color)
        grids = dict(sdf=self.sdf.device_view(), off=self.off_color.device_view(), emo=self.emo_color.device_view(),
                     brdf=self.brdf.device_view(),
                     emit=self.emo_color.device_view() if emit_grid is self.emo_color else emit_grid.device_view(),
                     mask=self.mask_cache.density.view(*self.mask_cache.density.shape[2:]))
        if emit_grid is self.emo_color:
            grids["emit"] = grids["emo"]
        env = dict(mus=self.envmap.mus.detach(), lambdas=self.envmap.lambdas.detach(), lobes=self.envmap.lobes.detach())
        em = kwargs["em_modes"]
        em = int(em.reshape(-1)[0]) if torch.is_tensor(em) else int(em)
        return eng.evaluate(self.scene_struct(), self.scene_struct(near=self.lts_near), kwargs["rays_o"].contiguous(),
                            kwargs["rays_d"].contiguous(), kwargs["viewdirs"].contiguous(), grids, env, kwargs["pos_rt"],
                            self.far, em, bool(kwargs.get("render_pbr", False)), int(kwargs.get("chunk_sz", 4096)),
                            self.num_2ndrays, draws)

    @torch.no_grad()
    def render_envmap(self, H, W):
        """Lat-long image [H,W,3] of the spherical-Gaussian environment map (esrnerf.py:1675-1691); a logging utility."""
        phi, theta = torch.meshgrid([torch.linspace(0.0, np.pi, H, device=self.device),
                                     torch.linspace(1.0 * np.pi, -1.0 * np.pi, W, device=self.device)], indexing="ij")
        dirs = torch.stack([torch.cos(theta) * torch.sin(phi), torch.sin(theta) * torch.sin(phi), torch.cos(phi)], -1)
        return self.envmap(dirs.view(-1, 3)).view(H, W, 3)

    @torch.no_grad()
    def scale_volume_grid(self, num_voxels):
        super().scale_volume_grid(num_voxels)
        self.brdf.scale_volume_grid(self.world_size)
