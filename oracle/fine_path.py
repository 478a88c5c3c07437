"""CPU restatement of the fine-stage training path -- TEST INFRASTRUCTURE ONLY.

A functional (state-dict driven) torch-CPU statement of what the reference's
``VoxurfF.forward_training`` plus the ``Fine.learn`` loss compute, written from the
algorithm, not from the code.  It is the checker for the HIP path and the
``cpu_baseline`` ("port") of bench.py; the product package never imports it.

Reference lines restated (relative to /root/reference):
  sampler + in-box compaction ..... app/fine/model/voxurff.py:623-654
  mask cache ...................... app/utils/base/module.py:95-114
  SDF value ....................... app/fine/model/voxurff.py:656-676
  NeuS alpha (interp) ............. app/utils/base/functions.py:72-105
  thresholds / compaction order ... app/fine/model/voxurff.py:200-213
  24-tap SDF stencil .............. app/fine/model/voxurff.py:678-721
  85-d feature assembly ........... app/fine/model/voxurff.py:225-254
  RadianceNet / TonemapNet ........ app/utils/pbr/module.py:6-39, voxurff.py:783-788
  compositing ..................... app/fine/model/voxurff.py:258-278
  loss ............................ app/fine/fine.py:355-382, utils2/image.py:14-26

Pinned against the imported reference by oracle/gen_golden.py and
tests/test_oracle_fine.py (outputs, loss and every parameter gradient).

Dense grids use the same third-party op as the reference (F.grid_sample,
bilinear, align_corners=True); ``trilinear_explicit`` below spells that op out
(index math, corner order, zero padding) and is cross-checked against it in the
tests because the HIP kernels implement exactly that spelling.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from oracle import native

Tensor = torch.Tensor


# --------------------------------------------------------------------------- #
# scene constants derived like voxurff.py:539-545
# --------------------------------------------------------------------------- #
@dataclass
class FineConsts:
    xyz_min: Tensor
    xyz_max: Tensor
    mask_xyz_min: Tensor
    mask_xyz_max: Tensor
    voxel_size: Tensor            # 0-dim fp32
    world_size: Tensor            # int64 [3]
    near: float
    stepsize: float
    mask_density: Tensor          # max-pooled, [1,1,a,b,c]
    act_shift: float
    maskcache_thres: float
    fastcolor_thres: float
    grad_feat: Tensor             # displacements of the feature stencil
    posbase_pe: int
    viewbase_pe: int
    colorbase_pe: int
    neus_alpha: str = "interp"    # or "grad" (app/utils/base/functions.py:45-69)


def grid_resolution(xyz_min: Tensor, xyz_max: Tensor, num_voxels: int):
    ext = xyz_max - xyz_min
    voxel_size = (ext.prod() / num_voxels).pow(1 / 3)
    world_size = (ext / voxel_size).long()
    return voxel_size, world_size


def make_consts(cfg_model, xyz_min, xyz_max, mask_xyz_min, mask_xyz_max, mask_alpha_init,
                mask_density, near, num_voxels) -> FineConsts:
    voxel_size, world_size = grid_resolution(xyz_min, xyz_max, num_voxels)
    ks = int(cfg_model.mask_ks)
    pooled = F.max_pool3d(mask_density, kernel_size=ks, padding=ks // 2, stride=1)
    return FineConsts(
        xyz_min=xyz_min, xyz_max=xyz_max, mask_xyz_min=mask_xyz_min, mask_xyz_max=mask_xyz_max,
        voxel_size=voxel_size, world_size=world_size, near=float(near),
        stepsize=float(cfg_model.stepsize), mask_density=pooled,
        act_shift=math.log(1 / (1 - mask_alpha_init) - 1),
        maskcache_thres=float(cfg_model.maskcache_thres),
        fastcolor_thres=float(cfg_model.fastcolor_thres),
        grad_feat=torch.tensor(list(cfg_model.grad_feat), dtype=torch.float32),
        posbase_pe=int(cfg_model.posbase_pe), viewbase_pe=int(cfg_model.viewbase_pe),
        colorbase_pe=int(cfg_model.colorbase_pe),
        neus_alpha=str(getattr(cfg_model, "neus_alpha", "interp")),
    )


# --------------------------------------------------------------------------- #
# dense-grid lookups
# --------------------------------------------------------------------------- #
def to_norm(xyz: Tensor, lo: Tensor, hi: Tensor) -> Tensor:
    """world xyz -> grid_sample coordinates; the flip maps world x to the slowest
    grid axis and world z to the fastest."""
    return ((xyz - lo) / (hi - lo)).flip((-1,)) * 2 - 1


def sample_grid(grid: Tensor, norm: Tensor) -> Tensor:
    """grid [1,C,X,Y,Z], norm [M,3] -> [M,C]."""
    out = F.grid_sample(grid, norm.reshape(1, 1, 1, -1, 3), mode="bilinear", align_corners=True)
    return out.reshape(grid.shape[1], -1).T


def trilinear_explicit(grid: Tensor, norm: Tensor) -> Tensor:
    """What F.grid_sample(bilinear, align_corners=True, zeros padding) does on a 5-D
    input, spelled out.  norm[..., 0] addresses the LAST grid axis."""
    C = grid.shape[1]
    sizes = [grid.shape[4], grid.shape[3], grid.shape[2]]           # per norm component
    g = grid[0]
    ix = [((norm[:, a] + 1) / 2) * (sizes[a] - 1) for a in range(3)]
    i0 = [torch.floor(v) for v in ix]
    out = torch.zeros(norm.shape[0], C, dtype=grid.dtype)
    # corner order of the ATen CPU kernel: the fastest axis toggles first
    for dz in (0, 1):          # slowest grid axis (norm component 2)
        for dy in (0, 1):
            for dx in (0, 1):  # fastest grid axis (norm component 0)
                d = (dx, dy, dz)
                w = None
                inb = torch.ones(norm.shape[0], dtype=torch.bool)
                idx = []
                for a in range(3):
                    c = i0[a] + d[a]
                    wa = (ix[a] - i0[a]) if d[a] else ((i0[a] + 1) - ix[a])
                    w = wa if w is None else w * wa
                    inb &= (c >= 0) & (c <= sizes[a] - 1)
                    idx.append(c.clamp(0, sizes[a] - 1).long())
                v = g[:, idx[2], idx[1], idx[0]].T                  # [M,C]
                out = out + torch.where(inb[:, None], v * w[:, None], torch.zeros_like(v))
    return out


def mask_cache(c: FineConsts, pts: Tensor) -> Tensor:
    d = sample_grid(c.mask_density, to_norm(pts, c.mask_xyz_min, c.mask_xyz_max))[:, 0]
    alpha = 1 - torch.exp(-F.softplus(d + c.act_shift))
    return alpha >= c.maskcache_thres


def sdf_stencil(c: FineConsts, sdf_grid: Tensor, pts: Tensor, displace: Tensor, diff_eps: float = 0.0):
    """Clamped +-displace taps along the three grid axes.

    Returns feat [M,6K] (tap order: -ax0,+ax0,-ax1,+ax1,-ax2,+ax2 with ax0 = world z,
    column = dir*K + k), grad [M,3K] ((f+ - f-)/clamped index distance/voxel) and
    normal [M,3K] (normalised over the three axes for every k)."""
    M, K = pts.shape[0], displace.numel()
    size_zyx = torch.tensor([sdf_grid.shape[4], sdf_grid.shape[3], sdf_grid.shape[2]])
    norm = to_norm(pts, c.xyz_min, c.xyz_max)
    ind = ((norm + 1) / 2) * (size_zyx - 1)
    axis_dir = torch.tensor([[-1, 0, 0], [1, 0, 0], [0, -1, 0], [0, 1, 0], [0, 0, -1], [0, 0, 1]])
    offset = (axis_dir[:, None, :] * displace[None, :, None]).reshape(-1, 3)    # [6K,3]
    taps = ind[:, None, :] + offset[None]                                        # [M,6K,3]
    taps = torch.minimum(torch.maximum(taps, torch.zeros(3)), (size_zyx - 1).float())
    tap_norm = (taps / (size_zyx - 1)) * 2 - 1
    feat = sample_grid(sdf_grid, tap_norm.reshape(-1, 3))[:, 0].reshape(M, 6, K)
    taps = taps.reshape(M, 6, K, 3)
    diff = (taps[:, 1::2] - taps[:, 0::2]).max(dim=-1).values                    # [M,3,K]
    # the LTS renderer guards the division: diff + 1e-12 (esrnerf.py:1560); the fine renderer does not
    grad = (feat[:, 1::2] - feat[:, 0::2]) / ((diff + diff_eps) if diff_eps else diff) / c.voxel_size
    normal = F.normalize(grad, dim=1)
    return feat.reshape(M, 6 * K), grad.reshape(M, 3 * K), normal.reshape(M, 3 * K)


def neus_alpha_interp(sdf: Tensor, ray_id: Tensor, s_val: float) -> Tensor:
    """Mid-point SDFs between consecutive surviving samples of the same ray."""
    same = ray_id[:-1] == ray_id[1:]
    mid = (sdf[:-1] + sdf[1:]) * 0.5
    nxt = torch.cat([torch.where(same, mid, sdf[:-1]), sdf[-1:]])
    prv = torch.cat([sdf[:1], torch.where(same, mid, sdf[1:])])
    pc = torch.sigmoid(prv * s_val)
    nc = torch.sigmoid(nxt * s_val)
    return ((F.relu(pc - nc) + 1e-5) / (pc + 1e-5)).clip(0.0, 1.0)


def neus_alpha_grad(viewdirs: Tensor, ray_id: Tensor, dist: Tensor, sdf: Tensor, grad: Tensor, s_val: float) -> Tensor:
    """cfg ``neus_alpha: grad`` (functions.py:45-69): section SDFs extrapolated along the ray from the sample's own
    value and gradient, sdf -+ 0.5 * dist * (viewdir . grad)."""
    half = (viewdirs[ray_id] * grad).sum(-1) * dist * 0.5
    pc = torch.sigmoid((sdf - half) * s_val)
    nc = torch.sigmoid((sdf + half) * s_val)
    return ((F.relu(pc - nc) + 1e-5) / (pc + 1e-5)).clip(0.0, 1.0)


def sdf_value_and_grad(c: FineConsts, P, pts: Tensor):
    """``sample_sdf_grad`` (voxurff.py:670-676): value + radius-1 clamped central differences, in world xyz order."""
    sdf = sample_grid(P["sdf.grid"], to_norm(pts, c.xyz_min, c.xyz_max))[:, 0]
    _, g1, _ = sdf_stencil(c, P["sdf.grid"], pts, torch.tensor([1.0]))
    return sdf, torch.cat([g1[:, [2]], g1[:, [1]], g1[:, [0]]], -1)


def alpha_of(c: FineConsts, P, pts: Tensor, ray_id: Tensor, viewdirs: Tensor, s_val: float):
    """(sdf, alpha) of the mask-cache survivors under the configured NeuS alpha mode (voxurff.py:193-199)."""
    if c.neus_alpha == "grad":
        sdf, grad = sdf_value_and_grad(c, P, pts)
        return sdf, neus_alpha_grad(viewdirs, ray_id, c.stepsize * c.voxel_size, sdf, grad, s_val)
    sdf = sample_grid(P["sdf.grid"], to_norm(pts, c.xyz_min, c.xyz_max))[:, 0]
    return sdf, neus_alpha_interp(sdf, ray_id, s_val)


def admit_survivors(m: Tensor, alpha: Tensor, key: Tensor, survivors: Tensor, keep: Optional[dict], name: str) -> Tensor:
    """Forced decisions, first threshold (``alpha > fastcolor_thres``): a sample the other implementation kept to the END is
    a candidate here too, whatever this evaluation's alpha says (``keep[name]``: the alphas of the samples admitted that
    way -- legitimate only ON the threshold, which the tests assert).  The other implementation's candidates that did NOT
    survive its second threshold are not known; this evaluation's own stay (each changes a transmittance by < 1e-4)."""
    admitted = torch.isin(key, survivors) & ~m
    if keep is not None:
        keep[name] = alpha.detach()[admitted]
    return m | admitted


class _Composite(torch.autograd.Function):
    """alpha -> (weights, alphainv_last) with the reference's early stop, through
    the C oracle (module.py:117-143 semantics)."""

    @staticmethod
    def forward(ctx, alpha, ray_id, n_rays):
        w, T, last, i_s, i_e = native.alpha2weight(alpha, ray_id, n_rays)
        ctx.save_for_backward(alpha.detach(), w, T, last, i_s, i_e)
        ctx.n_rays = n_rays
        return w, last

    @staticmethod
    def backward(ctx, gw, gl):
        alpha, w, T, last, i_s, i_e = ctx.saved_tensors
        g = native.alpha2weight_backward(alpha, w, T, last, i_s, i_e, ctx.n_rays,
                                         gw.contiguous(), gl.contiguous())
        return g, None, None


KNIFE_LOG: Optional[list] = None      # tests may set a list: receives (weight key, unit indices) of hidden units with |z| < 2e-6


FLIP_LOG: Optional[list] = None       # tests may set a list: receives one record per forced ReLU layer (see mlp)


def mlp(P: Dict[str, Tensor], keys, x: Tensor, knife: Optional[Tensor] = None, force=None) -> Tensor:
    """Linear+ReLU chain; the last layer is linear.  ``knife`` ([rows] float, optional) is lowered in place to the
    smallest |hidden pre-activation| of every row: a ReLU unit that close to its kink can take the other branch under
    a different fp32 summation order, which changes that row's GRADIENT discontinuously (tests use it to tell such
    rows apart; it does not enter the arithmetic).

    ``force`` (optional): one bool tensor [rows, width] per hidden layer -- the ReLU BRANCHES another implementation of
    the same net took (the HIP kernels' saved sign bits).  The layer's output is then ``pre-activation * mask`` instead
    of ``relu(pre-activation)``: the same piecewise-linear function on the same piece, so the two implementations can
    be compared at the tolerance of their arithmetic with no sample set aside.  Every unit whose forced branch differs
    from this evaluation's own is ARBITRATED in float64: FLIP_LOG receives (layer key, number of flipped units,
    largest |float64 pre-activation| among them) -- a flip is legitimate only where the exact value is within fp32
    summation noise of the kink (the tests assert the bound)."""
    for i, k in enumerate(keys):
        xin = x
        x = F.linear(x, P[k + ".weight"], P[k + ".bias"])
        if i + 1 < len(keys):
            if knife is not None:
                with torch.no_grad():
                    torch.minimum(knife, x.detach().abs().amin(-1), out=knife)
                    if KNIFE_LOG is not None:
                        KNIFE_LOG.append((k, (x.detach().abs() < 2e-6).nonzero()[:, 1].unique()))
            if force is not None:
                mask = force[i]
                with torch.no_grad():
                    flip = mask != (x.detach() > 0)
                    if FLIP_LOG is not None:
                        worst = 0.0
                        if bool(flip.any()):
                            rows = flip.any(-1).nonzero()[:, 0]
                            x64 = F.linear(xin.detach()[rows].double(), P[k + ".weight"].detach().double(),
                                           P[k + ".bias"].detach().double())
                            worst = float(x64[flip[rows]].abs().max())
                        FLIP_LOG.append((k, int(flip.sum()), worst))
                x = x * mask.to(x.dtype)
            else:
                x = F.relu(x)
    return x


RADIANCE_KEYS = ("linear.0", "linear.2.0", "linear.3.0", "linear.4")
TONEMAP_KEYS = ("srgb.0", "srgb.2")


def radiance(P, prefix, x, knife: Optional[Tensor] = None, force=None):
    return F.softplus(mlp(P, [f"{prefix}.{k}" for k in RADIANCE_KEYS], x, knife, force))


def tonemap(P, c: FineConsts, lin: Tensor, knife: Optional[Tensor] = None, force=None) -> Tensor:
    freq = torch.tensor([2.0 ** i for i in range(c.colorbase_pe)])
    emb = (lin.unsqueeze(-1) * freq).flatten(-2)
    x = torch.cat([lin, emb.sin(), emb.cos()], -1)
    return torch.sigmoid(mlp(P, [f"tonemapper.{k}" for k in TONEMAP_KEYS], x, knife, force))


def forward_training(P: Dict[str, Tensor], c: FineConsts, batch: Dict[str, Tensor], s_val: float,
                     keep: Optional[dict] = None, force: Optional[dict] = None) -> Dict[str, Tensor]:
    """P: tensors under the reference's state_dict names (sdf.grid, off_color.grid,
    emo_color.grid, off_rgbnet.*, emo_rgbnet.*, tonemapper.*).

    ``force`` (tests only): the DISCRETE decisions of another implementation of this path, taken over so that the two
    can be compared with nothing set aside --
      "survivors": int64 keys ``ray * 2**20 + step`` of ITS final survivor set: replaces this evaluation's
                   ``weights > fastcolor_thres`` test (``keep["threshold_flips"]``: the weights of the samples whose
                   membership changed -- legitimate only ON the threshold, which the tests assert);
      "relu":      callable (ray_id, step_id, on) -> dict(emo=[3 masks], off=[3 masks], tone=[1 mask]), bool tensors in
                   THIS evaluation's sample order (emo: the on-samples, off: the off-samples, tone: all): see ``mlp``."""
    rays_o, rays_d = batch["rays_o"].contiguous(), batch["rays_d"].contiguous()
    viewdirs, em_modes = batch["viewdirs"], batch["em_modes"]
    N = rays_o.shape[0]
    stepdist = c.stepsize * c.voxel_size
    pts, out_box, ray_id, step_id = native.sample_pts_on_rays(
        rays_o, rays_d, c.xyz_min, c.xyz_max, c.near, 1e9, float(stepdist))[:4]
    inb = ~out_box
    pts, ray_id, step_id = pts[inb], ray_id[inb], step_id[inb]
    n0 = pts.shape[0]

    m = mask_cache(c, pts)
    pts, ray_id, step_id = pts[m], ray_id[m], step_id[m]
    n1 = pts.shape[0]

    sdf, alpha = alpha_of(c, P, pts, ray_id, viewdirs, s_val)

    m = alpha > c.fastcolor_thres
    if force is not None and force.get("survivors") is not None:
        m = admit_survivors(m, alpha, ray_id * (1 << 20) + step_id, force["survivors"], keep, "alpha_flips")
    alpha, pts, ray_id, step_id, sdf = alpha[m], pts[m], ray_id[m], step_id[m], sdf[m]
    n2 = pts.shape[0]

    weights, alphainv_last = _Composite.apply(alpha, ray_id, N)
    m = weights > c.fastcolor_thres
    if force is not None and force.get("survivors") is not None:
        m_forced = torch.isin(ray_id * (1 << 20) + step_id, force["survivors"])
        if keep is not None:
            keep["threshold_flips"] = weights.detach()[m_forced != m]
        m = m_forced
    weights, pts, ray_id, step_id, sdf = weights[m], pts[m], ray_id[m], step_id[m], sdf[m]
    n3 = pts.shape[0]

    feat, _, normal = sdf_stencil(c, P["sdf.grid"], pts, c.grad_feat)
    unit = (pts - c.xyz_min) / (c.xyz_max - c.xyz_min)
    pfreq = torch.tensor([2.0 ** i for i in range(c.posbase_pe)])
    vfreq = torch.tensor([2.0 ** i for i in range(c.viewbase_pe)])
    pe = (unit.unsqueeze(-1) * pfreq).flatten(-2)
    ve = (viewdirs.unsqueeze(-1) * vfreq).flatten(-2)
    common = torch.cat([unit, pe.sin(), pe.cos(), ve[ray_id], ve.sin()[ray_id], ve.cos()[ray_id],
                        sdf[:, None], feat, normal], -1)

    on = em_modes[ray_id] == 1
    off = ~on
    norm_pts = to_norm(pts, c.xyz_min, c.xyz_max)
    lin = torch.zeros_like(pts)
    knife = torch.full((n3,), float("inf")) if keep is not None else None
    k_on, k_off = (knife[on], knife[off]) if knife is not None else (None, None)
    fr = force["relu"](ray_id, step_id, on) if (force is not None and force.get("relu") is not None) else {}
    x_on_emo = torch.cat([sample_grid(P["emo_color.grid"], norm_pts[on]), common[on]], -1)
    x_on_off = torch.cat([sample_grid(P["off_color.grid"], norm_pts[on]), common[on]], -1)
    # (the detached off-net pass on the on-samples carries no gradient: its branches are not forced)
    lin[on] = radiance(P, "emo_rgbnet", x_on_emo, k_on, fr.get("emo")) + radiance(P, "off_rgbnet", x_on_off).detach()
    x_off = torch.cat([sample_grid(P["off_color.grid"], norm_pts[off]), common[off]], -1)
    lin[off] = radiance(P, "off_rgbnet", x_off, k_off, fr.get("off"))

    rgb = tonemap(P, c, lin, knife, fr.get("tone"))
    if knife is not None:
        knife[on] = torch.minimum(knife[on], k_on)
        knife[off] = torch.minimum(knife[off], k_off)
    w = weights.unsqueeze(-1)
    rgb_marched = torch.zeros(N, 3).index_add(0, ray_id, w * rgb)
    lin_marched = torch.zeros(N, 3).index_add(0, ray_id, w * lin)
    if keep is not None:
        keep.update(counts=(n0, n1, n2, n3), ray_id=ray_id, step_id=step_id, weights=weights,
                    sdf=sdf, pts=pts, feat=feat, normal=normal, lin=lin, rgb=rgb, common=common, knife=knife)
    return {
        "etc/alphainv_cum": alphainv_last,
        "etc/white_bg": alphainv_last[..., None],
        "srgb/rgb": rgb_marched,
        "lin/rgb": lin_marched,
    }


@torch.no_grad()
def forward_evaluate(P: Dict[str, Tensor], c: FineConsts, batch: Dict[str, Tensor], s_val: float, far: float,
                     em_mode: int, pos_rt: Tensor) -> Dict[str, Tensor]:
    """VoxurfF.forward_evaluate (voxurff.py:280-461): image rendering -- the three radiance variants (off, emo,
    on = off + emo) tone-mapped separately, depth / disparity, camera-space normals from the radius-1 finite
    differences.  ``em_mode`` is one scalar per call (fine.py:549)."""
    rays_o, rays_d, viewdirs = batch["rays_o"].contiguous(), batch["rays_d"].contiguous(), batch["viewdirs"]
    N = rays_o.shape[0]
    stepdist = c.stepsize * c.voxel_size
    pts, out_box, ray_id, step_id = native.sample_pts_on_rays(rays_o, rays_d, c.xyz_min, c.xyz_max, c.near, 1e9,
                                                              float(stepdist))[:4]
    inb = ~out_box
    pts, ray_id, step_id = pts[inb], ray_id[inb], step_id[inb]
    m = mask_cache(c, pts)
    pts, ray_id, step_id = pts[m], ray_id[m], step_id[m]
    sdf, alpha = alpha_of(c, P, pts, ray_id, viewdirs, s_val)
    m = alpha > c.fastcolor_thres
    alpha, pts, ray_id, step_id, sdf = alpha[m], pts[m], ray_id[m], step_id[m], sdf[m]
    zeros = torch.zeros(N, 3)
    weights, alphainv_last = _Composite.apply(alpha, ray_id, N)
    m = weights > c.fastcolor_thres
    weights, pts, ray_id, step_id, sdf = weights[m], pts[m], ray_id[m], step_id[m], sdf[m]
    _, g1, _ = sdf_stencil(c, P["sdf.grid"], pts, torch.tensor([1.0]))
    grad = torch.cat([g1[:, [2]], g1[:, [1]], g1[:, [0]]], -1)
    feat, _, normal12 = sdf_stencil(c, P["sdf.grid"], pts, c.grad_feat)
    unit = (pts - c.xyz_min) / (c.xyz_max - c.xyz_min)
    pfreq = torch.tensor([2.0 ** i for i in range(c.posbase_pe)])
    vfreq = torch.tensor([2.0 ** i for i in range(c.viewbase_pe)])
    pe = (unit.unsqueeze(-1) * pfreq).flatten(-2)
    ve = (viewdirs.unsqueeze(-1) * vfreq).flatten(-2)
    common = torch.cat([unit, pe.sin(), pe.cos(), ve[ray_id], ve.sin()[ray_id], ve.cos()[ray_id], sdf[:, None], feat,
                        normal12], -1)
    npts = to_norm(pts, c.xyz_min, c.xyz_max)
    lin_off = radiance(P, "off_rgbnet", torch.cat([sample_grid(P["off_color.grid"], npts), common], -1))
    lin_emo = radiance(P, "emo_rgbnet", torch.cat([sample_grid(P["emo_color.grid"], npts), common], -1))
    lin_on = lin_off + lin_emo
    w = weights.unsqueeze(-1)
    comp = lambda x: zeros.clone().index_add(0, ray_id, w * x)
    out = {}
    for name, lin in (("off", lin_off), ("on", lin_on), ("emo", lin_emo)):
        out[f"srgb/{name}_rgb"] = comp(tonemap(P, c, lin))
        out[f"lin/{name}_rgb"] = comp(lin)
    nrm = F.normalize(grad, dim=-1) @ pos_rt
    nrm = (nrm * torch.tensor([1.0, -1.0, -1.0]) + 1.0) / 2.0
    depth = torch.zeros(N).index_add(0, ray_id, weights * step_id * stepdist)
    out.update({"etc/depth": depth, "etc/disp": 1 / (depth + alphainv_last * far), "etc/normal": comp(nrm),
                "etc/white_bg": alphainv_last.unsqueeze(-1)})
    pick = "off" if em_mode == 0 else "on"
    out["srgb/rgb"], out["lin/rgb"] = out[f"srgb/{pick}_rgb"], out[f"lin/{pick}_rgb"]
    return out


# --------------------------------------------------------------------------- #
# trainer-step loss
# --------------------------------------------------------------------------- #
def srgb_oetf(x: Tensor) -> Tensor:
    """Standard sRGB transfer curve (utils2/image.py:14-26)."""
    return torch.where(x <= 0.0031308, 12.92 * x, 1.055 * torch.pow(x.clamp(min=0.0031308), 1 / 2.4) - 0.055)


def fine_loss(results: Dict[str, Tensor], rgbs: Tensor, white_bg: bool = True,
              weight_linear: float = 0.1, weight_entropy_last: float = 0.001):
    """MSE(srgb) + w_lin MSE(gamma(lin)) + w_ent entropy(alphainv_last)."""
    bg = results["etc/white_bg"] * (1.0 if white_bg else 0.0)
    srgb = (results["srgb/rgb"] + bg).clamp(0.0, 1.0)
    lin = (results["lin/rgb"] + bg).clamp(min=0.0)
    l_srgb = F.mse_loss(srgb, rgbs)
    l_lin = F.mse_loss(srgb_oetf(torch.where(rgbs >= 1, lin.clamp(max=1.0), lin)), rgbs)
    # reference quirk kept (fine.py:378): alphainv_cum is [N], so ``[..., -1]`` picks
    # the LAST RAY only -- the entropy term regularises a single ray per batch.
    p = results["etc/alphainv_cum"][..., -1].clamp(1e-6, 1 - 1e-6)
    ent = -(p * torch.log(p) + (1 - p) * torch.log(1 - p)).mean()
    loss = l_srgb + weight_linear * l_lin + weight_entropy_last * ent
    return loss, dict(srgb_mse=l_srgb.detach(), lin_mse=l_lin.detach(), entropy=ent.detach())


def params_from_state_dict(sd: Dict[str, Tensor], requires_grad: bool = True) -> Dict[str, Tensor]:
    P = {}
    for k, v in sd.items():
        if k.startswith("tv_smooth_conv"):
            continue
        t = v.detach().clone().float()
        t.requires_grad_(requires_grad)
        P[k] = t
    return P
