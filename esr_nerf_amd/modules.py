"""Field / network primitives with the reference's interface (names, ctor
arguments, state_dict keys) whose hot methods run on libesr_hip.so.

Mirrors (reference paths):
  app/utils/base/module.py:9-75    DenseGrid
  app/utils/base/module.py:78-114  MaskCache
  app/utils/base/module.py:117-143 Alphas2Weights
  app/utils/base/module.py:146-211 Gaussian3DConv, GradientConv (fixed-kernel convs)
  app/utils/pbr/module.py:6-39     RadianceNet, TonemapNet

Differences that are deliberate and invisible through the interface:
  * multi-channel DenseGrid parameters are STORED channels-last
    (torch.channels_last_3d): logical shape stays [1,C,X,Y,Z] -- state_dict, the
    optimizer and checkpoints see the reference layout -- while the HIP gathers
    read one 24-B voxel record instead of 6 cache lines;
  * the MLP classes hold the parameters under the reference's names; on the
    training path they are evaluated by the fused MFMA kernels
    (esr_nerf_amd/fine_engine.py), never through their torch ``forward`` (kept only
    so the objects stay ordinary nn.Modules for utilities outside the hot path).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import render_utils


class ForwardSwitch:
    """The renderers' ``train()`` protocol (voxurff.py:116-121, esrnerf.py:218-239 in the reference: ``self.forward =
    self.forward_training`` / ``forward_evaluate`` / ``forward_finetune``) without the reference CYCLE that assignment makes:
    a bound method stored on its own instance keeps the model -- and with it the engine's workspaces, 5-7 GB of device memory
    at the bench sizes -- alive until Python's cycle collector happens to run, which device memory pressure does not trigger
    (a 140-experiment statistics run ended in an out-of-memory error with 285 GB held by dead models).  ``forward`` is a
    property that looks the chosen method up on every call; assigning one of the instance's own methods records its NAME,
    anything else is stored as given."""
    _forward_name = "forward_evaluate"

    @property
    def forward(self):
        fn = self.__dict__.get("_forward_fn")
        return fn if fn is not None else getattr(self, self._forward_name)

    @forward.setter
    def forward(self, fn):
        if getattr(fn, "__self__", None) is self:
            self.__dict__.pop("_forward_fn", None)
            self.__dict__["_forward_name"] = fn.__name__
        else:
            self.__dict__["_forward_fn"] = fn


class DenseGrid(nn.Module):
    def __init__(self, channels: int, world_size: torch.Tensor, xyz_min: torch.Tensor,
                 xyz_max: torch.Tensor):
        super().__init__()
        self.channels = channels
        self.world_size = world_size
        self.xyz_min = xyz_min
        self.xyz_max = xyz_max
        self.grid = nn.Parameter(self._storage(torch.zeros([1, channels, *[int(v) for v in world_size]])))

    def _storage(self, t: torch.Tensor) -> torch.Tensor:
        if self.channels > 1:
            return t.contiguous(memory_format=torch.channels_last_3d)
        return t.contiguous()

    def device_view(self) -> torch.Tensor:
        """Tensor whose memory is [X,Y,Z,C] contiguous (a view when the storage is
        already channels-last, which is the normal case)."""
        g = self.grid
        if self.channels == 1:
            return g.detach().contiguous().view(*g.shape[2:])
        return g.detach().permute(0, 2, 3, 4, 1).contiguous()[0]

    def forward(self, xyz):
        # torch path kept for callers outside the fused renderer (evaluation utilities)
        shape = xyz.shape[:-1]
        pts = xyz.reshape(1, 1, 1, -1, 3)
        norm = ((pts - self.xyz_min) / (self.xyz_max - self.xyz_min)).flip((-1,)) * 2 - 1
        out = F.grid_sample(self.grid, norm, mode="bilinear", align_corners=True)
        out = out.reshape(self.channels, -1).T.reshape(*shape, self.channels)
        return out.squeeze(-1) if self.channels == 1 else out

    def scale_volume_grid(self, new_world_size):
        self.world_size = new_world_size
        size = tuple(int(v) for v in new_world_size)
        if self.channels == 0:
            self.grid = nn.Parameter(torch.zeros([1, self.channels, *size]))
        else:
            up = F.interpolate(self.grid.data.contiguous(), size=size, mode="trilinear", align_corners=True)
            self.grid = nn.Parameter(self._storage(up))

    def total_variation_add_grad(self, wx, wy, wz, dense_mode, mask=None):
        if mask is not None:
            raise NotImplementedError("masked TV is dead code in the reference (callers pass mask=None)")
        if self.channels != 1:
            g = self.grid.grad.contiguous()
            render_utils.total_variation_add_grad(self.grid.detach().contiguous(), g, wx, wy, wz, dense_mode)
            self.grid.grad.copy_(g)
        else:
            render_utils.total_variation_add_grad(self.grid.detach(), self.grid.grad, wx, wy, wz, dense_mode)

    def get_dense_grid(self):
        return self.grid

    @torch.no_grad()
    def __isub__(self, val):
        self.grid.data -= val
        return self

    def extra_repr(self):
        return f"channels={self.channels}, world_size={[int(v) for v in self.world_size]}"


class MaskCache(nn.Module):
    def __init__(self, xyz_min, xyz_max, density, alpha_init, cache_thres, ks):
        super().__init__()
        self.xyz_min = xyz_min
        self.xyz_max = xyz_max
        self.mask_cache_thres = cache_thres
        self.ks = ks
        self.density = F.max_pool3d(density, kernel_size=ks, padding=ks // 2, stride=1).contiguous()
        self.act_shift = math.log(1 / (1 - alpha_init) - 1)

    @torch.no_grad()
    def forward(self, xyz):
        shape = xyz.shape[:-1]
        pts = xyz.reshape(1, 1, 1, -1, 3)
        norm = ((pts - self.xyz_min) / (self.xyz_max - self.xyz_min)).flip((-1,)) * 2 - 1
        d = F.grid_sample(self.density, norm, align_corners=True)
        alpha = 1 - torch.exp(-F.softplus(d + self.act_shift))
        return alpha.reshape(*shape) >= self.mask_cache_thres


class Alphas2Weights(torch.autograd.Function):
    """alpha -> (weights, alphainv_last) on the HIP compositing kernels."""

    @staticmethod
    def forward(ctx, alpha, ray_id, N):
        weights, T, last, i_s, i_e = render_utils.alpha2weight(alpha.contiguous(), ray_id.contiguous(), N)
        if alpha.requires_grad:
            ctx.save_for_backward(alpha, weights, T, last, i_s, i_e)
            ctx.n_rays = N
        return weights, last

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_weights, grad_last):
        alpha, weights, T, last, i_s, i_e = ctx.saved_tensors
        g = render_utils.alpha2weight_backward(alpha, weights, T, last, i_s, i_e, ctx.n_rays,
                                               grad_weights.contiguous(), grad_last.contiguous())
        return g, None, None


def _fixed_conv3d(weight: np.ndarray, channel: int = 1) -> nn.Conv3d:
    k = weight.shape[-1]
    m = nn.Conv3d(channel, channel, k, stride=1, padding=k // 2, padding_mode="replicate", groups=channel)
    w = torch.from_numpy(weight).float()
    m.weight.data = torch.stack([w[None] for _ in range(channel)], 0)
    m.bias.data = torch.zeros(channel)
    for p in m.parameters():
        p.requires_grad = False
    return m


class Gaussian3DConv(nn.Module):
    def __init__(self, ksize: int = 3, sigma: float = 1.0, channel: int = 1):
        super().__init__()
        r = np.arange(-(ksize // 2), ksize // 2 + 1, 1)
        xx, yy, zz = np.meshgrid(r, r, r)
        kern = np.exp(-(xx ** 2 + yy ** 2 + zz ** 2) / (2 * sigma ** 2))
        kern = torch.FloatTensor(kern).numpy()          # reference rounds to fp32 before normalising
        self.m = _fixed_conv3d(kern / kern.sum(), channel)

    def forward(self, x):
        return self.m(x)


class GradientConv(nn.Module):
    """Frozen 3x3x3 smoothing conv of the TV term (state_dict keys tv_smooth_conv.m.*)."""

    def __init__(self, sigma: int = 0):
        super().__init__()
        one_d = np.array([1.0, 2.0, 1.0])
        base = one_d[:, None, None] * one_d[None, :, None] * one_d[None, None, :]
        idx = np.arange(3) - 1
        dist = idx[:, None, None] ** 2 + idx[None, :, None] ** 2 + idx[None, None, :] ** 2 - 1
        kern = base * np.exp(-dist * sigma)
        self.m = nn.Conv3d(1, 1, (3, 3, 3), stride=1, padding=1, padding_mode="replicate")
        self.m.weight.data = torch.from_numpy(kern / kern.sum()).float()[None, None]
        self.m.bias.data = torch.zeros(1)
        for p in self.m.parameters():
            p.requires_grad = False

    def forward(self, x):
        return self.m(x)


def _mlp_stack(din: int, width: int, depth: int, dout: int) -> nn.Sequential:
    """Linear+ReLU, (depth-2) x Sequential(Linear, ReLU), Linear -- the nesting fixes the
    checkpoint key names (``0``, ``2.0``, ``3.0``, ..., ``depth``)."""
    # construction order = first, hidden..., last: it fixes the RNG draw order of the default init,
    # so a seeded build reproduces the reference's initial weights bit for bit
    first = nn.Linear(din, width)
    mid = [nn.Sequential(nn.Linear(width, width), nn.ReLU(inplace=True)) for _ in range(depth - 2)]
    return nn.Sequential(first, nn.ReLU(inplace=True), *mid, nn.Linear(width, dout, bias=True))


def _linears(seq: nn.Sequential):
    out = []
    for m in seq:
        if isinstance(m, nn.Linear):
            out.append(m)
        elif isinstance(m, nn.Sequential):
            out.append(m[0])
    return out


class RadianceNet(nn.Module):
    def __init__(self, inputdim: int, width: int, depth: int):
        super().__init__()
        self.linear = _mlp_stack(inputdim, width, depth, 3)

    def layers(self):
        return _linears(self.linear)

    def forward(self, x) -> torch.Tensor:
        return F.softplus(self.linear(x))


class TonemapNet(nn.Module):
    def __init__(self, dim0: int, width: int, depth: int):
        super().__init__()
        self.srgb = _mlp_stack(dim0, width, depth, 3)

    def layers(self):
        return _linears(self.srgb)

    def forward(self, x):
        return torch.sigmoid(self.srgb(x))


def extract_sdf_field(model, resolution=512, batch_size=64, smooth=True, sigma=0.5) -> torch.Tensor:
    """``-sdf`` on a resolution^3 lattice of the bounding box (extract_fields + the query of extract_geometry,
    app/utils/base/functions.py:108-139, voxurff.py:745-770).  A mesh-export utility outside the rendering path:
    plain torch ``grid_sample`` in ``batch_size``^3 blocks."""
    lo, hi = model.xyz_min.float(), model.xyz_max.float()
    grid = model.sdf.grid
    if smooth:
        grid = Gaussian3DConv(sigma=sigma).to(grid.device)(grid)
    if resolution is None:
        resolution = int(model.world_size[0])
    axes = [torch.linspace(float(lo[i]), float(hi[i]), resolution, device=grid.device) for i in range(3)]
    u = torch.zeros([resolution] * 3, device=grid.device)
    with torch.no_grad():
        for xi, xs in enumerate(axes[0].split(batch_size)):
            for yi, ys in enumerate(axes[1].split(batch_size)):
                for zi, zs in enumerate(axes[2].split(batch_size)):
                    pts = torch.stack(torch.meshgrid(xs, ys, zs, indexing="ij"), -1).reshape(1, 1, 1, -1, 3)
                    norm = ((pts - lo) / (hi - lo)).flip((-1,)) * 2 - 1
                    val = F.grid_sample(-grid, norm, mode="bilinear", align_corners=True).reshape(len(xs), len(ys), len(zs))
                    u[xi * batch_size: xi * batch_size + len(xs), yi * batch_size: yi * batch_size + len(ys),
                      zi * batch_size: zi * batch_size + len(zs)] = val
    return u


def extract_geometry(model, resolution=512, threshold=0.0, batch_size=64, smooth=True, sigma=0.5):
    """(vertices, triangles) of the zero level set, as the reference's ``extract_geometry`` (needs PyMCubes)."""
    try:
        import mcubes
    except ImportError as e:                                         # not in this image; the reference requires it too
        raise ImportError("extract_geometry needs PyMCubes (`mcubes`), as the reference does (requirements.txt)") from e
    u = extract_sdf_field(model, resolution, batch_size, smooth, sigma).cpu().numpy()
    vertices, triangles = mcubes.marching_cubes(u, threshold)
    lo, hi = model.xyz_min.float().cpu().numpy(), model.xyz_max.float().cpu().numpy()
    res = u.shape[0]
    return vertices / (res - 1.0) * (hi - lo)[None, :] + lo[None, :], triangles
