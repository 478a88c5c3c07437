"""CPU restatement of the coarse-stage renderer ``VoxurfC.forward_training`` and the loss lines of the
coarse trainer.  TEST INFRASTRUCTURE ONLY -- imported by tests/, smoke() and
bench.py's cpu_baseline; the product path (esr_nerf_amd/) never touches it.

Follows (paths under the reference tree):
  app/coarse/model/voxurfc.py:186-271   forward_training
  app/coarse/model/voxurfc.py:597-616   neus_sdf_gradient (dense central differences of the RAW sdf grid)
  app/utils/base/module.py:145-177      Gaussian3DConv (5^3 kernel, replicate padding, normalised)
  app/utils/base/functions.py:72-105    neus_alpha_from_sdf_scatter_interp
  app/coarse/coarse.py:338-352          trainer loss (white background, clamp, MSE, last-ray entropy)
Differences to the fine stage that matter for parity: the SDF is read from the SMOOTHED grid; no
alpha > thres mask; alpha2weight runs a second time over the weight > thres survivors (voxurfc.py:211,219)
and weights / alphainv_last come from that second pass; 12 colour channels; the normal feature is the
trilinearly sampled dense gradient divided by (|g| + 1e-5); both radiance heads are sigmoids that are
summed (no detach, no tone mapper); white_bg = 1 - sum of weights.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor

from . import fine_path as fp
from . import native

RGB_KEYS = ("0", "2.0", "3")


def gaussian_kernel(ksize: int, sigma: float) -> Tensor:
    """[1,1,k,k,k] normalised Gaussian as module.py:152-171 builds it (float32 of the float64 exp, then / sum)."""
    r = np.arange(-(ksize // 2), ksize // 2 + 1, 1)
    xx, yy, zz = np.meshgrid(r, r, r)
    k = torch.FloatTensor(np.exp(-(xx ** 2 + yy ** 2 + zz ** 2) / (2 * sigma ** 2)))
    return (k[None, None] / k.sum())


def smooth_grid(grid: Tensor, kernel: Tensor) -> Tensor:
    k = kernel.shape[-1]
    return F.conv3d(F.pad(grid, [k // 2] * 6, mode="replicate"), kernel)


def dense_gradient(grid: Tensor, voxel_size) -> Tensor:
    g = torch.zeros([1, 3, *grid.shape[-3:]])
    g[:, 0, 1:-1] = (grid[:, 0, 2:] - grid[:, 0, :-2]) / 2 / voxel_size
    g[:, 1, :, 1:-1] = (grid[:, 0, :, 2:] - grid[:, 0, :, :-2]) / 2 / voxel_size
    g[:, 2, :, :, 1:-1] = (grid[:, 0, :, :, 2:] - grid[:, 0, :, :, :-2]) / 2 / voxel_size
    return g


def make_consts(cfg_model, xyz_min, xyz_max, mask_xyz_min, mask_xyz_max, mask_alpha_init, mask_density, near):
    voxel_size, world_size = fp.grid_resolution(xyz_min, xyz_max, int(cfg_model.num_voxels))
    ks = int(cfg_model.mask_ks)
    pooled = F.max_pool3d(mask_density, kernel_size=ks, padding=ks // 2, stride=1)
    return fp.FineConsts(
        xyz_min=xyz_min, xyz_max=xyz_max, mask_xyz_min=mask_xyz_min, mask_xyz_max=mask_xyz_max,
        voxel_size=voxel_size, world_size=world_size, near=float(near), stepsize=float(cfg_model.stepsize),
        mask_density=pooled, act_shift=math.log(1 / (1 - mask_alpha_init) - 1),
        maskcache_thres=float(cfg_model.maskcache_thres), fastcolor_thres=float(cfg_model.fastcolor_thres),
        grad_feat=torch.zeros(0), posbase_pe=int(cfg_model.posbase_pe), viewbase_pe=int(cfg_model.viewbase_pe),
        colorbase_pe=0, neus_alpha=str(getattr(cfg_model, "neus_alpha", "interp")))


def _alpha(c, viewdirs, ray_id, dist, sdf, grad, s_val):
    """voxurfc.py:171-174, 207-210: the configured NeuS alpha; "grad" extrapolates with the SAMPLED dense gradient."""
    if c.neus_alpha == "grad":
        return fp.neus_alpha_grad(viewdirs, ray_id, dist, sdf, grad, s_val)
    return fp.neus_alpha_interp(sdf, ray_id, s_val)


def forward_training(P: Dict[str, Tensor], c: fp.FineConsts, batch: Dict[str, Tensor], s_val: float,
                     ksize: int = 5, sigma: float = 0.8, keep=None) -> Dict[str, Tensor]:
    rays_o, rays_d, viewdirs, em_modes = batch["rays_o"], batch["rays_d"], batch["viewdirs"], batch["em_modes"]
    N = rays_o.shape[0]
    stepdist = c.stepsize * c.voxel_size
    pts, out_box, ray_id = native.sample_pts_on_rays(rays_o.contiguous(), rays_d.contiguous(), c.xyz_min, c.xyz_max,
                                                      c.near, 1e9, float(stepdist))[:3]
    inb = ~out_box
    pts, ray_id = pts[inb], ray_id[inb]
    n0 = pts.shape[0]
    m = fp.mask_cache(c, pts)
    pts, ray_id = pts[m], ray_id[m]
    n1 = pts.shape[0]
    sm = smooth_grid(P["sdf.grid"], gaussian_kernel(ksize, sigma))
    norm = fp.to_norm(pts, c.xyz_min, c.xyz_max)
    sdf = fp.sample_grid(sm, norm)[:, 0]
    grad = fp.sample_grid(dense_gradient(P["sdf.grid"], c.voxel_size), norm)
    alpha = _alpha(c, viewdirs, ray_id, torch.as_tensor(float(stepdist)), sdf, grad, s_val)
    weights, _ = fp._Composite.apply(alpha, ray_id, N)
    m = weights > c.fastcolor_thres
    pts, ray_id, alpha, grad, norm = pts[m], ray_id[m], alpha[m], grad[m], norm[m]
    weights, alphainv_last = fp._Composite.apply(alpha, ray_id, N)
    n3 = pts.shape[0]
    on = em_modes[ray_id] == 1
    unit = (pts - c.xyz_min) / (c.xyz_max - c.xyz_min)
    pf = torch.tensor([2.0 ** i for i in range(c.posbase_pe)])
    vf = torch.tensor([2.0 ** i for i in range(c.viewbase_pe)])
    xe = (unit.unsqueeze(-1) * pf).flatten(-2)
    ve = (viewdirs.unsqueeze(-1) * vf).flatten(-2)
    normal = grad / (grad.norm(dim=-1, keepdim=True) + 1e-5)
    feat = torch.cat([unit, xe.sin(), xe.cos(), ve[ray_id], ve.sin()[ray_id], ve.cos()[ray_id], normal], -1)
    rgb = torch.zeros_like(pts)
    x_on = torch.cat([fp.sample_grid(P["emo_color.grid"], norm[on]), feat[on]], -1)
    rgb[on] = torch.sigmoid(fp.mlp(P, [f"emo_rgbnet.{k}" for k in RGB_KEYS], x_on))
    x_all = torch.cat([fp.sample_grid(P["off_color.grid"], norm), feat], -1)
    rgb = rgb + torch.sigmoid(fp.mlp(P, [f"off_rgbnet.{k}" for k in RGB_KEYS], x_all))
    w = weights.unsqueeze(-1)
    rgb_m = torch.zeros(N, 3).index_add(0, ray_id, w * rgb)
    cum = torch.zeros(N, 1).index_add(0, ray_id, w)
    if keep is not None:
        keep.update(counts=(n0, n1, n1, n3), ray_id=ray_id)
    return {"etc/alphainv_cum": alphainv_last, "etc/white_bg": 1 - cum, "srgb/rgb": rgb_m}


def coarse_loss(results: Dict[str, Tensor], rgbs: Tensor, white_bg: bool = True, weight_entropy_last: float = 0.001):
    """app/coarse/coarse.py:341-352 (TV terms excluded)."""
    srgb = (results["srgb/rgb"] + results["etc/white_bg"] * (1.0 if white_bg else 0.0)).clamp(min=0.0, max=1.0)
    loss = F.mse_loss(srgb, rgbs)
    pout = results["etc/alphainv_cum"][..., -1].clamp(1e-6, 1 - 1e-6)
    ent = -(pout * torch.log(pout) + (1 - pout) * torch.log(1 - pout)).mean()
    return loss + weight_entropy_last * ent, dict(mse=float(loss.detach()))


@torch.no_grad()
def forward_evaluate(P: Dict[str, Tensor], c: fp.FineConsts, batch: Dict[str, Tensor], s_val: float, far: float,
                     em_mode: int, pos_rt: Tensor, ksize: int = 5, sigma: float = 0.8) -> Dict[str, Tensor]:
    """VoxurfC.forward_evaluate (voxurfc.py:273-422): image rendering of the coarse stage."""
    rays_o, rays_d, viewdirs = batch["rays_o"], batch["rays_d"], batch["viewdirs"]
    N = rays_o.shape[0]
    stepdist = c.stepsize * c.voxel_size
    pts, out_box, ray_id, step_id = native.sample_pts_on_rays(rays_o.contiguous(), rays_d.contiguous(), c.xyz_min, c.xyz_max,
                                                              c.near, 1e9, float(stepdist))[:4]
    inb = ~out_box
    pts, ray_id, step_id = pts[inb], ray_id[inb], step_id[inb]
    m = fp.mask_cache(c, pts)
    pts, ray_id, step_id = pts[m], ray_id[m], step_id[m]
    norm = fp.to_norm(pts, c.xyz_min, c.xyz_max)
    sdf = fp.sample_grid(smooth_grid(P["sdf.grid"], gaussian_kernel(ksize, sigma)), norm)[:, 0]
    grad = fp.sample_grid(dense_gradient(P["sdf.grid"], c.voxel_size), norm)
    alpha = _alpha(c, viewdirs, ray_id, torch.as_tensor(float(stepdist)), sdf, grad, s_val)
    weights, _ = fp._Composite.apply(alpha, ray_id, N)
    m = weights > c.fastcolor_thres
    pts, ray_id, step_id, alpha, grad, norm = pts[m], ray_id[m], step_id[m], alpha[m], grad[m], norm[m]
    weights, _ = fp._Composite.apply(alpha, ray_id, N)
    unit = (pts - c.xyz_min) / (c.xyz_max - c.xyz_min)
    pf = torch.tensor([2.0 ** i for i in range(c.posbase_pe)])
    vf = torch.tensor([2.0 ** i for i in range(c.viewbase_pe)])
    xe = (unit.unsqueeze(-1) * pf).flatten(-2)
    ve = (viewdirs.unsqueeze(-1) * vf).flatten(-2)
    normal = grad / (grad.norm(dim=-1, keepdim=True) + 1e-5)
    feat = torch.cat([unit, xe.sin(), xe.cos(), ve[ray_id], ve.sin()[ray_id], ve.cos()[ray_id], normal], -1)
    off = torch.sigmoid(fp.mlp(P, [f"off_rgbnet.{k}" for k in RGB_KEYS], torch.cat([fp.sample_grid(P["off_color.grid"], norm), feat], -1)))
    emo = torch.sigmoid(fp.mlp(P, [f"emo_rgbnet.{k}" for k in RGB_KEYS], torch.cat([fp.sample_grid(P["emo_color.grid"], norm), feat], -1)))
    w = weights.unsqueeze(-1)
    comp = lambda x: torch.zeros(N, x.shape[-1]).index_add(0, ray_id, w * x)
    out = {"srgb/off_rgb": comp(off), "srgb/emo_rgb": comp(emo), "srgb/on_rgb": comp(off + emo)}
    bg = 1 - comp(torch.ones_like(w))
    nrm = ((normal @ pos_rt) * torch.tensor([1.0, -1.0, -1.0]) + 1.0) / 2.0
    depth = torch.zeros(N).index_add(0, ray_id, weights * step_id * stepdist)
    out.update({"etc/depth": depth, "etc/disp": 1 / (depth + bg[..., -1] * far), "etc/normal": comp(nrm), "etc/white_bg": bg})
    out["srgb/rgb"] = out["srgb/off_rgb"] if em_mode == 0 else out["srgb/on_rgb"]
    return out
