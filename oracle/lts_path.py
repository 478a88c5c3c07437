"""CPU restatement of the LTS-stage training path -- TEST INFRASTRUCTURE ONLY.

Functional torch-CPU statement of ``ESRNeRF.forward_training`` with
``pdra_mode`` off/on and of the ``LTS.learn`` loss,
written from the algorithm.  Checker for the LTS kernels; never imported by the
product package.

Reference lines restated (relative to /root/reference):
  primary pass ..................... app/fine/model/esrnerf.py:681-788
  exact trilinear SDF gradient ..... app/fine/model/esrnerf.py:1572-1605, app/utils/base/functions.py:142-309
  stencil with the +1e-12 guard .... app/fine/model/esrnerf.py:1527-1570
  light-transport segment .......... app/fine/model/esrnerf.py:487-679
  hemisphere directions ............ app/utils/pbr/functions.py:10-18
  Disney reflection ................ app/utils/pbr/functions.py:108-173
  BRDF / emission / SG env nets .... app/utils/pbr/module.py:42-143
  perturbed re-evaluations ......... app/fine/model/esrnerf.py:807-830
  loss ............................. app/fine/lts.py:337-379

Randomness.  The reference draws, in this order: ``np.random.choice`` (which
surviving samples host a light-transport segment), ``torch.randn`` for the
hemisphere directions, two ``torch.randn_like`` for the perturbations.  Here the
draws are explicit inputs (``Draws``) so that the imported reference, this
restatement and the HIP path can be fed identical numbers.
"""
from __future__ import annotations

import math

from dataclasses import dataclass
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from oracle import fine_path as fp
from oracle import native

Tensor = torch.Tensor
BRDF_KEYS = ("brdfnet.0", "brdfnet.2.0", "brdfnet.3.0", "brdfnet.4")


@dataclass
class Draws:
    idx: Tensor          # [P] int64, indices into the surviving primary samples
    dirs: Tensor         # [P, R+1, 3] standard normal (before normalisation / hemisphere flip)
    noise_normal: Tensor  # [M3, 3] standard normal
    noise_emit: Tensor    # [M3, 3] standard normal


def trilinear_xyz(c: fp.FineConsts, grid: Tensor, pts: Tensor) -> Tensor:
    """1-channel trilinear interpolant as a differentiable function of the WORLD point
    (weights before clamping, border-replicated indices: functions.py:153-260)."""
    g = grid[0, 0]
    size = [g.shape[0], g.shape[1], g.shape[2]]
    u = (pts - c.xyz_min) / (c.xyz_max - c.xyz_min)
    n = u * 2 - 1
    ix = [((n[:, a] + 1) / 2) * (size[a] - 1) for a in range(3)]
    with torch.no_grad():
        i0 = [torch.floor(v) for v in ix]
    out = 0
    for dx in (0, 1):
        for dy in (0, 1):
            for dz in (0, 1):
                d = (dx, dy, dz)
                w = 1
                idx = []
                for a in range(3):
                    w = w * ((ix[a] - i0[a]) if d[a] else ((i0[a] + 1) - ix[a]))
                    idx.append((i0[a] + d[a]).clamp(0, size[a] - 1).long())
                out = out + g[idx[0], idx[1], idx[2]] * w
    return out


def sdf_expgrad(c: fp.FineConsts, grid: Tensor, pts: Tensor):
    """SDF value and its exact spatial gradient (differentiable w.r.t. the grid)."""
    with torch.enable_grad():
        p = pts.detach().clone().requires_grad_(True)
        sdf = trilinear_xyz(c, grid, p)
        (g,) = torch.autograd.grad(sdf.sum(), p, create_graph=True)
    return sdf, g


def fib_dirs(count: int) -> Tensor:
    """Fibonacci-spiral hemisphere table of ``ray_sampling: fib`` (pbr/functions.py:176-194, random = False, up = True):
    point k of a 2*count-point spiral for k = count .. 2*count-1; azimuth golden_angle * ((k + 1) mod 2 count), height
    (k + 0.5) / count - 1.  Used as the un-normalised ``dirs`` draw: every surface point gets the same table."""
    k = torch.arange(count, 2 * count).to(torch.float32)
    golden = math.pi * (3.0 - math.sqrt(5.0))
    az = golden * torch.remainder(k + 1.0, 2 * count)
    h = (k + 0.5) * (1.0 / count) - 1.0
    r = torch.sqrt(1.0 - h * h)
    return torch.stack([torch.cos(az) * r, torch.sin(az) * r, h], -1)


def hemisphere_dirs(normal: Tensor, raw: Tensor) -> Tensor:
    d = F.normalize(raw, dim=-1)
    flip = (d * normal.unsqueeze(-2)).sum(-1) < 0
    return torch.where(flip[..., None], -d, d)


def disney_reflection(albedo, roughness, metallic, normal, win, wout):
    eps = 1e-7
    dot = lambda a, b: (a * b).sum(-1, keepdim=True)
    h = F.normalize(win + wout, dim=-1)
    noh, ooh = dot(normal, h).clamp(min=0), dot(wout, h).clamp(min=0)
    ion, oon = dot(win, normal).clamp(min=0), dot(wout, normal).clamp(min=0)
    fd = (1 - metallic) * albedo / torch.pi
    r2 = (roughness * roughness).clamp(min=eps)
    D = 1 / (r2 * torch.pi) * torch.exp(2 / r2 * (noh - 1))
    f0 = 0.04 * (1 - metallic) + albedo * metallic
    Fr = f0 + (1.0 - f0) * ((1.0 - ooh) ** 5)

    def v(cos):
        k = ((1 + roughness) ** 2) / 8
        return 0.5 / (cos * (1 - k) + k).clamp(min=eps)

    fs = D * Fr * (v(ion) * v(oon))
    return (fd + fs) * ion * torch.pi * 2


def sg_envmap(P: Dict[str, Tensor], dirs: Tensor) -> Tensor:
    lobes = F.normalize(P["envmap.lobes"], dim=-1)
    lam = P["envmap.lambdas"].abs()
    e = torch.exp(lam * ((dirs.unsqueeze(-2) * lobes).sum(-1, keepdim=True) - 1.0))
    return F.softplus((P["envmap.mus"] * e).sum(-2))


def brdf_net(P, x, knife=None, force=None):
    o = torch.sigmoid(fp.mlp(P, [f"brdfnet.{k}" for k in BRDF_KEYS], x, knife, force))
    return o[:, 0:3], o[:, 3:4], o[:, 4:5]


def emit_net(P, x, knife=None, force=None):
    return F.softplus(fp.mlp(P, [f"emitnet.{k}" for k in BRDF_KEYS], x, knife, force))


def _knife(keep, pts):
    """With ``keep``: a per-row record of the smallest |hidden pre-activation| over every net evaluated on ``pts``
    (fp.mlp), registered under keep["knife_sets"] as (positions, record) -- test bookkeeping, not arithmetic."""
    if keep is None:
        return None
    k = torch.full((pts.shape[0],), float("inf"))
    keep.setdefault("knife_sets", []).append((pts.detach(), k))
    return k


def _pe(c: fp.FineConsts, pts: Tensor) -> Tensor:
    unit = (pts - c.xyz_min) / (c.xyz_max - c.xyz_min)
    freq = torch.tensor([2.0 ** i for i in range(c.posbase_pe)])
    e = (unit.unsqueeze(-1) * freq).flatten(-2)
    return torch.cat([unit, e.sin(), e.cos()], -1)


def _view_pe(c: fp.FineConsts, v: Tensor) -> Tensor:
    freq = torch.tensor([2.0 ** i for i in range(c.viewbase_pe)])
    e = (v.unsqueeze(-1) * freq).flatten(-2)
    return torch.cat([e, e.sin(), e.cos()], -1)


def _march(P, c: fp.FineConsts, rays_o, rays_d, near, s_val, counts: Optional[list] = None):
    """sampler -> mask cache -> SDF -> alpha -> thresholds -> weights (voxurff/esrnerf common part).
    ``counts`` (a list) receives the in-box and mask-cache survivor counts M0, M1."""
    N = rays_o.shape[0]
    stepdist = c.stepsize * c.voxel_size
    pts, out_box, ray_id, step_id = native.sample_pts_on_rays(rays_o.contiguous(), rays_d.contiguous(), c.xyz_min,
                                                               c.xyz_max, near, 1e9, float(stepdist))[:4]
    inb = ~out_box
    pts, ray_id, step_id = pts[inb], ray_id[inb], step_id[inb]
    m = fp.mask_cache(c, pts)
    if counts is not None:
        counts += [int(pts.shape[0]), int(m.sum()), step_id[m]]
    return N, pts[m], ray_id[m]


def _alpha(c, P, pts, ray_id, viewdirs, sdf, s_val):
    """The configured NeuS alpha of one march (esrnerf.py:197-200).  "grad": section SDFs extrapolated with the radius-1
    finite-difference gradient of ``sample_sdf_grad`` (esrnerf.py:1519-1525; its + 1e-12 guard included) along the
    rays' view directions -- for the secondary rays those are their own directions (esrnerf.py:575-591)."""
    if c.neus_alpha == "grad":
        _, g1, _ = fp.sdf_stencil(c, P["sdf.grid"], pts, torch.tensor([1.0]), diff_eps=1e-12)
        grad = torch.cat([g1[:, [2]], g1[:, [1]], g1[:, [0]]], -1)
        return fp.neus_alpha_grad(viewdirs, ray_id, c.stepsize * c.voxel_size, sdf, grad, s_val)
    return fp.neus_alpha_interp(sdf, ray_id, s_val)


def _stencil(c, grid, pts):
    return fp.sdf_stencil(c, grid, pts, c.grad_feat, diff_eps=1e-12)


def forward_training(P: Dict[str, Tensor], c: fp.FineConsts, batch: Dict[str, Tensor], s_val: float,
                     draws: Draws, normal_eps: float, emit_eps: float, num_2ndrays: int, lts_near: float,
                     pdra_mode: bool = False, keep: Optional[dict] = None, force: Optional[dict] = None) -> Dict[str, Tensor]:
    """``force`` (tests only; oracle/fine_path.py: forward_training): the discrete decisions of another implementation, taken
    over pass by pass -- "prim_survivors" / "sec_survivors": int64 keys ``ray * 2**20 + step`` of its final survivor sets;
    "prim": callable (ray_id, step_id, on) -> dict(emo, off, tone, brdf, emit) of ReLU-branch masks in this evaluation's
    sample order (emo: the on-samples); "pts": dict(off, emo) for the 2 P point rows; "sec": callable (ray_id, step_id) ->
    dict(off, emo); "eps": dict(emit, brdf) for the perturbed heads' rows."""
    force = force or {}
    rays_o, rays_d, viewdirs = batch["rays_o"], batch["rays_d"], batch["viewdirs"]
    em_modes, uncert = batch["em_modes"], batch["uncert_masks"]
    prim_counts: list = []
    N, pts, ray_id = _march(P, c, rays_o, rays_d, c.near, s_val, prim_counts)
    step_id = prim_counts.pop()                               # step ids of the mask-cache survivors
    sdf, expg = sdf_expgrad(c, P["sdf.grid"], pts)
    alpha = _alpha(c, P, pts, ray_id, viewdirs, sdf, s_val)
    m = alpha > c.fastcolor_thres
    if force.get("prim_survivors") is not None:
        m = fp.admit_survivors(m, alpha, ray_id * (1 << 20) + step_id, force["prim_survivors"], keep, "alpha_flips")
    alpha, pts, ray_id, sdf, expg, step_id = alpha[m], pts[m], ray_id[m], sdf[m], expg[m], step_id[m]
    prim_counts.append(int(pts.shape[0]))
    weights, alphainv_last = fp._Composite.apply(alpha, ray_id, N)
    m = weights > c.fastcolor_thres
    if force.get("prim_survivors") is not None:
        m_forced = torch.isin(ray_id * (1 << 20) + step_id, force["prim_survivors"])
        if keep is not None:
            keep["threshold_flips"] = weights.detach()[m_forced != m]
        m = m_forced
    weights, pts, ray_id, sdf, expg, step_id = weights[m], pts[m], ray_id[m], sdf[m], expg[m], step_id[m]

    on = em_modes[ray_id] == 1
    feat, _, nrm = _stencil(c, P["sdf.grid"], pts)
    xyz_pe = _pe(c, pts)
    vpe = _view_pe(c, viewdirs)[ray_id]
    common = torch.cat([xyz_pe, vpe, sdf[:, None], feat, nrm], -1)
    gpts = fp.to_norm(pts, c.xyz_min, c.xyz_max)
    lin = torch.zeros_like(pts)
    kn = _knife(keep, pts)
    kn_on = kn[on] if kn is not None else None
    fr = force["prim"](ray_id, step_id, on) if force.get("prim") is not None else {}
    lin[on] = fp.radiance(P, "emo_rgbnet", torch.cat([fp.sample_grid(P["emo_color.grid"], gpts[on]), common[on]], -1), kn_on,
                          fr.get("emo"))
    if kn is not None:
        kn[on] = kn_on
    lin = lin + fp.radiance(P, "off_rgbnet", torch.cat([fp.sample_grid(P["off_color.grid"], gpts), common], -1), kn, fr.get("off"))
    rgb = fp.tonemap(P, c, lin, kn, fr.get("tone"))
    bfeat = torch.cat([xyz_pe, sdf[:, None], feat, nrm], -1)
    base, rough, metal = brdf_net(P, torch.cat([fp.sample_grid(P["brdf.grid"], gpts), bfeat], -1), kn, fr.get("brdf"))
    emit = emit_net(P, torch.cat([fp.sample_grid(P["emo_color.grid"], gpts), bfeat], -1), kn, fr.get("emit"))
    w = weights.unsqueeze(-1)
    zeros = lambda: torch.zeros(N, 3)
    rgb_m = zeros().index_add(0, ray_id, w * rgb)
    lin_m = zeros().index_add(0, ray_id, w * lin)
    emit_m = zeros().index_add(0, ray_id, w * emit)
    normal = F.normalize(expg.detach(), dim=-1)

    idx = draws.idx
    lts = light_transport_segment(P, c, pts[idx], viewdirs[ray_id][idx], normal[idx], sdf[idx], base[idx],
                                  rough[idx], metal[idx], emit[idx], uncert[ray_id][idx], draws.dirs,
                                  s_val, num_2ndrays, lts_near, pdra_mode, keep=keep, force=force)
    _, expg_eps = sdf_expgrad(c, P["sdf.grid"], pts + draws.noise_normal * normal_eps)
    pts2 = pts + draws.noise_emit * emit_eps
    gp2 = fp.to_norm(pts2, c.xyz_min, c.xyz_max)
    sdf2 = fp.sample_grid(P["sdf.grid"], gp2)[:, 0]
    feat2, _, nrm2 = _stencil(c, P["sdf.grid"], pts2)
    bfeat2 = torch.cat([_pe(c, pts2), sdf2[:, None], feat2, nrm2], -1)
    kn2 = _knife(keep, pts2)
    fe = force.get("eps") or {}
    emit2 = emit_net(P, torch.cat([fp.sample_grid(P["emo_color.grid"], gp2), bfeat2], -1), kn2, fe.get("emit"))
    base2, rough2, metal2 = brdf_net(P, torch.cat([fp.sample_grid(P["brdf.grid"], gp2), bfeat2], -1), kn2, fe.get("brdf"))
    if keep is not None:
        keep.update(m3=pts.shape[0], ray_id=ray_id, step_id=step_id, pts=pts, counts=tuple(prim_counts + [int(pts.shape[0])]))
    return {
        "etc/alphainv_cum": alphainv_last, "etc/white_bg": alphainv_last[..., None],
        "srgb/rgb": rgb_m, "lin/rgb": lin_m,
        "lin/pbr/off": lts["off"], "lin/pbr/off_hat": lts["off_hat"],
        "lin/pbr/emo": lts["emo"], "lin/pbr/emo_hat": lts["emo_hat"],
        "etc/emit_uncert": emit_m[uncert], "etc/emit_cert": emit_m[~uncert],
        "etc/normal": expg, "etc/normal_eps": expg_eps, "etc/emit": emit, "etc/emit_eps": emit2,
        "etc/brdf": torch.cat([base, rough, metal], -1), "etc/brdf_eps": torch.cat([base2, rough2, metal2], -1),
    }


def light_transport_segment(P, c, pts, viewdirs, normal, sdf, base, rough, metal, emission, umask, raw_dirs,
                            s_val, R, lts_near, pdra_mode, keep: Optional[dict] = None, force: Optional[dict] = None):
    """Outgoing radiance of P surface points predicted by the radiance nets ("off", "emo", for the
    camera direction and one random direction) against the rendering equation evaluated with R
    secondary rays per point ("off_hat", "emo_hat")."""
    Pn = pts.shape[0]
    dirs_all = hemisphere_dirs(normal, raw_dirs)
    v_rand = -dirs_all[:, -1]
    dirs = dirs_all[:, :-1]
    feat, _, nrm = _stencil(c, P["sdf.grid"], pts)
    xyz_pe = _pe(c, pts)
    vpe = _view_pe(c, torch.cat([viewdirs, v_rand], 0))
    rep = lambda t: t.repeat([2] + [1] * (t.dim() - 1))
    common = torch.cat([rep(xyz_pe), vpe, rep(sdf[:, None]), rep(feat), rep(nrm)], -1)
    gp = fp.to_norm(pts, c.xyz_min, c.xyz_max)
    knp = _knife(keep, rep(pts))
    force = force or {}
    fpt = force.get("pts") or {}
    off = fp.radiance(P, "off_rgbnet", torch.cat([rep(fp.sample_grid(P["off_color.grid"], gp)), common], -1), knp, fpt.get("off"))
    emo = fp.radiance(P, "emo_rgbnet", torch.cat([rep(fp.sample_grid(P["emo_color.grid"], gp)), common], -1), knp, fpt.get("emo"))

    ex = lambda t: t.view(Pn, 1, -1).expand(Pn, R, t.shape[-1]).flatten(0, 1)
    o2, v2, vr2, n2 = ex(pts), ex(viewdirs), ex(v_rand), ex(normal)
    d2 = dirs.flatten(0, 1)
    Rf = disney_reflection(rep(ex(base)), rep(ex(rough)), rep(ex(metal)), rep(n2), rep(d2),
                           torch.cat([-v2, -vr2], 0))
    # incoming radiance along the secondary rays
    sec_counts: list = []
    N2, p2, rid = _march(P, c, o2, d2, lts_near, s_val, sec_counts)
    sid2 = sec_counts.pop()                                   # step ids of the mask-cache survivors (bookkeeping)
    s2 = fp.sample_grid(P["sdf.grid"], fp.to_norm(p2, c.xyz_min, c.xyz_max))[:, 0]
    a2 = _alpha(c, P, p2, rid, d2, s2, s_val) if s2.numel() > 1 else s2.new_zeros(s2.shape)
    m = a2 > c.fastcolor_thres
    force = force or {}
    if force.get("sec_survivors") is not None:
        m = fp.admit_survivors(m, a2, rid * (1 << 20) + sid2, force["sec_survivors"], keep, "sec_alpha_flips")
    a2, p2, rid, s2, sid2 = a2[m], p2[m], rid[m], s2[m], sid2[m]
    sec_counts.append(int(p2.shape[0]))
    w2, last2 = fp._Composite.apply(a2, rid, N2)
    m = w2 > c.fastcolor_thres
    if force.get("sec_survivors") is not None:
        m_forced = torch.isin(rid * (1 << 20) + sid2, force["sec_survivors"])
        if keep is not None:
            keep["sec_threshold_flips"] = w2.detach()[m_forced != m]
        m = m_forced
    w2, p2, rid, s2, sid2 = w2[m], p2[m], rid[m], s2[m], sid2[m]
    if keep is not None:
        keep["sec_counts"] = tuple(sec_counts + [int(p2.shape[0])])
        keep["sec"] = dict(ray_id=rid, step_id=sid2, weights=w2.detach(), rays_o=o2.detach(), rays_d=d2.detach())
    f2, _, nr2 = _stencil(c, P["sdf.grid"], p2)
    feat2 = torch.cat([_pe(c, p2), _view_pe(c, d2)[rid], s2[:, None], f2, nr2], -1)
    g2 = fp.to_norm(p2, c.xyz_min, c.xyz_max)
    kns = _knife(keep, p2)
    fs = force["sec"](rid, sid2) if force.get("sec") is not None else {}
    loff = fp.radiance(P, "off_rgbnet", torch.cat([fp.sample_grid(P["off_color.grid"], g2), feat2], -1), kns, fs.get("off"))
    lemo = fp.radiance(P, "emo_rgbnet", torch.cat([fp.sample_grid(P["emo_color.grid"], g2), feat2], -1), kns, fs.get("emo"))
    off_m = torch.zeros(N2, 3).index_add(0, rid, w2[:, None] * loff)
    emo_m = torch.zeros(N2, 3).index_add(0, rid, w2[:, None] * lemo)
    env = sg_envmap(P, d2) * last2.unsqueeze(-1)
    off_hat = (rep(off_m + env) * Rf).view(-1, R, 3).mean(-2)
    reflect = (rep(emo_m) * Rf).view(-1, R, 3).mean(-2)
    if pdra_mode:
        um = rep(umask)
        emo_hat = torch.where(um[:, None], rep(emission) + reflect.detach(), reflect)
    else:
        emo_hat = rep(emission) + reflect
    return dict(off=off, emo=emo, off_hat=off_hat, emo_hat=emo_hat)


def lts_loss(results: Dict[str, Tensor], rgbs: Tensor, white_bg: bool = True, weight_linear: float = 10.0,
             weight_lts: float = 0.01, weight_entropy_last: float = 0.001, weight_normal_smooth: float = 0.001):
    """app/fine/lts.py:337-379 (TV terms excluded, as in fine_loss)."""
    base, aux = fp.fine_loss(results, rgbs, white_bg, weight_linear, weight_entropy_last)
    l_off = F.mse_loss(results["lin/pbr/off"], results["lin/pbr/off_hat"])
    l_emo = F.mse_loss(results["lin/pbr/emo"], results["lin/pbr/emo_hat"])
    l_n = F.l1_loss(results["etc/normal"], results["etc/normal_eps"])
    return base + weight_lts * (l_off + l_emo) + weight_normal_smooth * l_n, aux


def pdra_loss(results: Dict[str, Tensor], rgbs: Tensor, white_bg: bool = True, weight_linear: float = 10.0,
              weight_lts: float = 0.01, weight_entropy_last: float = 0.001, weight_normal_smooth: float = 0.001,
              weight_emit_smooth: float = 0.1, weight_lts_l: float = 50.0, weight_lts_r: float = 1.0,
              weight_emit_supp: float = 0.1):
    """app/fine/pdra.py:383-457 (TV terms excluded, as in fine_loss)."""
    base, aux = fp.fine_loss(results, rgbs, white_bg, weight_linear, weight_entropy_last)
    l_off = F.l1_loss(results["lin/pbr/off"], results["lin/pbr/off_hat"])
    l_emo_l = F.l1_loss(results["lin/pbr/emo"].detach(), results["lin/pbr/emo_hat"])
    l_emo_r = F.l1_loss(results["lin/pbr/emo"], results["lin/pbr/emo_hat"].detach())
    loss = base + weight_lts * l_off + weight_lts * (weight_lts_l * l_emo_l + weight_lts_r * l_emo_r)
    emc = results["etc/emit_cert"]
    if len(emc):
        loss = loss + weight_emit_supp * torch.pow(emc, 2).mean()
    loss = loss + weight_normal_smooth * F.l1_loss(results["etc/normal"], results["etc/normal_eps"])
    loss = loss + weight_emit_smooth * F.l1_loss(results["etc/emit"], results["etc/emit_eps"])
    return loss, aux


# ------------------------------------------------------------------ re-lighting fine-tune (A16)
def rgb_to_hsv(rgb: Tensor, eps: float = 1e-8) -> Tensor:
    """app/utils/pbr/functions.py:214-236: hue in [0,1), saturation = delta / (max + eps), value = max."""
    mx, arg = rgb.max(-1)
    mn = rgb.min(-1).values
    delta = mx - mn
    s = delta / (mx + eps)
    d = torch.where(delta == 0, torch.ones_like(delta), delta)
    rc, gc, bc = (mx.unsqueeze(-1) - rgb).unbind(-1)
    cand = torch.stack([bc - gc, (rc - bc) + 2.0 * d, (gc - rc) + 4.0 * d], -1) / d.unsqueeze(-1)
    h = cand.gather(-1, arg.unsqueeze(-1)).squeeze(-1)
    return torch.stack([(h / 6.0) % 1.0, s, mx], -1)


def hsv_to_rgb(hsv: Tensor) -> Tensor:
    """app/utils/pbr/functions.py:239-255 (sector table v,q,p,p,t,v / t,v,v,q,p,p / p,p,t,v,v,q)."""
    h, s, v = hsv.unbind(-1)
    sector = torch.floor(h * 6) % 6
    f = ((h * 6) % 6) - sector
    p, q, t = v * (1 - s), v * (1 - f * s), v * (1 - (1 - f) * s)
    k = sector.long().unsqueeze(-1)
    r = torch.stack([v, q, p, p, t, v], -1).gather(-1, k)
    g = torch.stack([t, v, v, q, p, p], -1).gather(-1, k)
    b = torch.stack([p, p, t, v, v, q], -1).gather(-1, k)
    return torch.cat([r, g, b], -1)


def edit_emission(emit: Tensor, em_modes: Tensor, em_intensities: Tensor, em_colors: Tensor) -> Tensor:
    """esrnerf.py:432-441: mode 0 off, 2/4 intensity scale, 3/4 hue+saturation replaced."""
    emit = emit.clone()
    emit[em_modes == 0] = 0
    im = (em_modes == 2) | (em_modes == 4)
    emit[im] = emit[im] * em_intensities[im][..., None]
    cm = (em_modes == 3) | (em_modes == 4)
    hsv = rgb_to_hsv(emit[cm])
    hsv[..., :-1] = em_colors[cm]
    emit[cm] = hsv_to_rgb(hsv)
    return emit


def forward_finetune(P: Dict[str, Tensor], c: fp.FineConsts, batch: Dict[str, Tensor], s_val: float, idx: Tensor,
                     raw_dirs: Tensor, num_2ndrays: int, lts_near: float) -> Dict[str, Tensor]:
    """ESRNeRF.forward_finetune (esrnerf.py:241-484).  P["emit_color.grid"] is the frozen copy of the emo
    colour grid that feeds the emission head; only emo_color.grid / emo_rgbnet.* receive gradients (the
    reference evaluates everything else under no_grad)."""
    rays_o, rays_d, viewdirs = batch["rays_o"], batch["rays_d"], batch["viewdirs"]
    R = num_2ndrays
    with torch.no_grad():
        N, pts, ray_id = _march(P, c, rays_o, rays_d, c.near, s_val)
        sdf = fp.sample_grid(P["sdf.grid"], fp.to_norm(pts, c.xyz_min, c.xyz_max))[:, 0]
        alpha = _alpha(c, P, pts, ray_id, viewdirs, sdf, s_val)
        m = alpha > c.fastcolor_thres
        alpha, pts, ray_id = alpha[m], pts[m], ray_id[m]
        weights, _ = fp._Composite.apply(alpha, ray_id, N)
        m = weights > c.fastcolor_thres
        pts, ray_id = pts[m], ray_id[m]
        p, vd = pts[idx], viewdirs[ray_id][idx]
        modes, inten, cols = batch["em_modes"][ray_id][idx], batch["em_intensities"][ray_id][idx], batch["em_colors"][ray_id][idx]
        Pn = p.shape[0]
        sdf_p, expg = sdf_expgrad(c, P["sdf.grid"], p)
        sdf_p, normal = sdf_p.detach(), F.normalize(expg.detach(), dim=-1)
        dirs_all = hemisphere_dirs(normal, raw_dirs)
        v_rand, dirs = -dirs_all[:, -1], dirs_all[:, :-1]
        feat, _, nrm = _stencil(c, P["sdf.grid"], p)
        xyz_pe = _pe(c, p)
        rep = lambda t: t.repeat([2] + [1] * (t.dim() - 1))
        common = torch.cat([rep(xyz_pe), _view_pe(c, torch.cat([vd, v_rand], 0)), rep(sdf_p[:, None]), rep(feat), rep(nrm)], -1)
        gp = fp.to_norm(p, c.xyz_min, c.xyz_max)
        bfeat = torch.cat([xyz_pe, sdf_p[:, None], feat, nrm], -1)
        base, rough, metal = brdf_net(P, torch.cat([fp.sample_grid(P["brdf.grid"], gp), bfeat], -1))
        emit = emit_net(P, torch.cat([fp.sample_grid(P["emit_color.grid"], gp), bfeat], -1))
    emo = fp.radiance(P, "emo_rgbnet", torch.cat([rep(fp.sample_grid(P["emo_color.grid"], gp)), common.detach()], -1))
    with torch.no_grad():
        ex = lambda t: t.view(Pn, 1, -1).expand(Pn, R, t.shape[-1]).flatten(0, 1)
        o2, d2 = ex(p), dirs.flatten(0, 1)
        Rf = disney_reflection(rep(ex(base)), rep(ex(rough)), rep(ex(metal)), rep(ex(normal)), rep(d2),
                               torch.cat([-ex(vd), -ex(v_rand)], 0))
        N2, p2, rid = _march(P, c, o2, d2, lts_near, s_val)
        s2 = fp.sample_grid(P["sdf.grid"], fp.to_norm(p2, c.xyz_min, c.xyz_max))[:, 0]
        a2 = _alpha(c, P, p2, rid, d2, s2, s_val) if s2.numel() > 1 else s2.new_zeros(s2.shape)
        m = a2 > c.fastcolor_thres
        a2, p2, rid, s2 = a2[m], p2[m], rid[m], s2[m]
        w2, _ = fp._Composite.apply(a2, rid, N2)
        m = w2 > c.fastcolor_thres
        w2, p2, rid, s2 = w2[m], p2[m], rid[m], s2[m]
        f2, _, nr2 = _stencil(c, P["sdf.grid"], p2)
        feat2 = torch.cat([_pe(c, p2), _view_pe(c, d2)[rid], s2[:, None], f2, nr2], -1)
        lemo = fp.radiance(P, "emo_rgbnet", torch.cat([fp.sample_grid(P["emo_color.grid"], fp.to_norm(p2, c.xyz_min, c.xyz_max)), feat2], -1))
        emo_m = torch.zeros(N2, 3).index_add(0, rid, w2[:, None] * lemo)
        emit = edit_emission(emit, modes, inten, cols)
        reflect = (rep(emo_m) * Rf).view(-1, R, 3).mean(-2)
        emo_hat = rep(emit) + reflect
    return {"lin/pbr/emo": emo, "lin/pbr/emo_hat": emo_hat}


# ------------------------------------------------------------------ PDRA regrouping queries
@torch.no_grad()
def _primary_survivors(P, c, batch, s_val):
    rays_o, rays_d = batch["rays_o"], batch["rays_d"]
    N, pts, ray_id = _march(P, c, rays_o, rays_d, c.near, s_val)
    sdf = fp.sample_grid(P["sdf.grid"], fp.to_norm(pts, c.xyz_min, c.xyz_max))[:, 0]
    alpha = _alpha(c, P, pts, ray_id, batch["viewdirs"], sdf, s_val)
    m = alpha > c.fastcolor_thres
    alpha, pts, ray_id, sdf = alpha[m], pts[m], ray_id[m], sdf[m]
    weights, _ = fp._Composite.apply(alpha, ray_id, N)
    m = weights > c.fastcolor_thres
    return N, weights[m], pts[m], ray_id[m], sdf[m]


@torch.no_grad()
def eval_emit(P, c, batch, s_val, emit_grid_key: str = "emo_color.grid") -> Tensor:
    """ESRNeRF.eval_emit (esrnerf.py:1299-1358): composited emission per ray, [N,3].  In eval mode the emission
    head reads ``emit_color``, which is ``emo_color`` unless a fine-tune froze a copy (esrnerf.py:236-238)."""
    N, w, pts, ray_id, sdf = _primary_survivors(P, c, batch, s_val)
    feat, _, nrm = _stencil(c, P["sdf.grid"], pts)
    bfeat = torch.cat([_pe(c, pts), sdf[:, None], feat, nrm], -1)
    emit = emit_net(P, torch.cat([fp.sample_grid(P[emit_grid_key], fp.to_norm(pts, c.xyz_min, c.xyz_max)), bfeat], -1))
    return torch.zeros(N, 3).index_add(0, ray_id, w[:, None] * emit)


@torch.no_grad()
def eval_esp(P, c, batch, s_val) -> Tensor:
    """ESRNeRF.eval_esp (esrnerf.py:1360-1407): weight-composited sample position per ray, [N,3]."""
    N, w, pts, ray_id, _ = _primary_survivors(P, c, batch, s_val)
    return torch.zeros(N, 3).index_add(0, ray_id, w[:, None] * pts)


# ------------------------------------------------------------------ image rendering (esrnerf.py:853-1297)
@torch.no_grad()
def _evaluate_lts(P, c, pts, viewdirs, normal, base, rough, metal, emit, raw_dirs, s_val, R, lts_near):
    """The per-sample light-transport decomposition of forward_evaluate (esrnerf.py:854-1001) for one chunk."""
    n = pts.shape[0]
    dirs = hemisphere_dirs(normal, raw_dirs)
    ex = lambda t: t.view(n, 1, -1).expand(n, R, t.shape[-1]).flatten(0, 1)
    d2 = dirs.flatten(0, 1)
    Rf = disney_reflection(ex(base), ex(rough), ex(metal), ex(normal), d2, -ex(viewdirs))
    N2, p2, rid = _march(P, c, ex(pts), d2, lts_near, s_val)
    s2 = fp.sample_grid(P["sdf.grid"], fp.to_norm(p2, c.xyz_min, c.xyz_max))[:, 0]
    a2 = _alpha(c, P, p2, rid, d2, s2, s_val) if s2.numel() > 1 else s2.new_zeros(s2.shape)
    m = a2 > c.fastcolor_thres
    a2, p2, rid, s2 = a2[m], p2[m], rid[m], s2[m]
    w2, last2 = fp._Composite.apply(a2, rid, N2)
    m = w2 > c.fastcolor_thres
    w2, p2, rid, s2 = w2[m], p2[m], rid[m], s2[m]
    f2, _, nr2 = _stencil(c, P["sdf.grid"], p2)
    feat2 = torch.cat([_pe(c, p2), _view_pe(c, d2)[rid], s2[:, None], f2, nr2], -1)
    g2 = fp.to_norm(p2, c.xyz_min, c.xyz_max)
    loff = fp.radiance(P, "off_rgbnet", torch.cat([fp.sample_grid(P["off_color.grid"], g2), feat2], -1))
    lemo = fp.radiance(P, "emo_rgbnet", torch.cat([fp.sample_grid(P["emo_color.grid"], g2), feat2], -1))
    off_m = torch.zeros(N2, 3).index_add(0, rid, w2[:, None] * loff)
    emo_m = torch.zeros(N2, 3).index_add(0, rid, w2[:, None] * lemo)
    env = sg_envmap(P, d2) * last2.unsqueeze(-1)
    mean = lambda t: t.view(-1, R, 3).mean(-2)
    out = {"lin/env_dir": mean(env * Rf), "lin/env_indir": mean(off_m * Rf), "lin/emit_(in)dir": mean(emo_m * Rf)}
    out["lin/env_effects"] = out["lin/env_dir"] + out["lin/env_indir"]
    out["lin/emit_effects"] = emit + out["lin/emit_(in)dir"]
    return out


@torch.no_grad()
def forward_evaluate(P, c, batch, s_val, far, em_mode, pos_rt, render_pbr, chunk_sz, raw_dirs_chunks, num_2ndrays,
                     lts_near, emit_grid_key: str = "emo_color.grid") -> Dict[str, Tensor]:
    """ESRNeRF.forward_evaluate (esrnerf.py:853-1297).  ``raw_dirs_chunks``: the standard-normal draws of
    diffuse_scattering, one [chunk, R, 3] tensor per chunk of ``chunk_sz`` surviving samples (render_pbr only)."""
    rays_o, rays_d, viewdirs = batch["rays_o"].contiguous(), batch["rays_d"].contiguous(), batch["viewdirs"]
    N = rays_o.shape[0]
    stepdist = c.stepsize * c.voxel_size
    pts, out_box, ray_id, step_id = native.sample_pts_on_rays(rays_o, rays_d, c.xyz_min, c.xyz_max, c.near, 1e9,
                                                              float(stepdist))[:4]
    inb = ~out_box
    pts, ray_id, step_id = pts[inb], ray_id[inb], step_id[inb]
    m = fp.mask_cache(c, pts)
    pts, ray_id, step_id = pts[m], ray_id[m], step_id[m]
    sdf, expg = sdf_expgrad(c, P["sdf.grid"], pts)
    sdf, expg = sdf.detach(), expg.detach()
    alpha = _alpha(c, P, pts, ray_id, viewdirs, sdf, s_val)
    m = alpha > c.fastcolor_thres
    alpha, pts, ray_id, step_id, sdf, expg = alpha[m], pts[m], ray_id[m], step_id[m], sdf[m], expg[m]
    weights, alphainv_last = fp._Composite.apply(alpha, ray_id, N)
    m = weights > c.fastcolor_thres
    weights, pts, ray_id, step_id, sdf, expg = weights[m], pts[m], ray_id[m], step_id[m], sdf[m], expg[m]
    _, g1, _ = fp.sdf_stencil(c, P["sdf.grid"], pts, torch.tensor([1.0]), diff_eps=1e-12)
    grad = torch.cat([g1[:, [2]], g1[:, [1]], g1[:, [0]]], -1)
    feat, _, nrm12 = _stencil(c, P["sdf.grid"], pts)
    xyz_pe = _pe(c, pts)
    common = torch.cat([xyz_pe, _view_pe(c, viewdirs)[ray_id], sdf[:, None], feat, nrm12], -1)
    gp = fp.to_norm(pts, c.xyz_min, c.xyz_max)
    lin_off = fp.radiance(P, "off_rgbnet", torch.cat([fp.sample_grid(P["off_color.grid"], gp), common], -1))
    lin_emo = fp.radiance(P, "emo_rgbnet", torch.cat([fp.sample_grid(P["emo_color.grid"], gp), common], -1))
    lin_on = lin_off + lin_emo
    bfeat = torch.cat([xyz_pe, sdf[:, None], feat, nrm12], -1)
    base, rough, metal = brdf_net(P, torch.cat([fp.sample_grid(P["brdf.grid"], gp), bfeat], -1))
    emit = emit_net(P, torch.cat([fp.sample_grid(P[emit_grid_key], gp), bfeat], -1))
    w = weights.unsqueeze(-1)
    comp3 = lambda x: torch.zeros(N, 3).index_add(0, ray_id, w * x)
    comp1 = lambda x: torch.zeros(N).index_add(0, ray_id, weights * x)
    out = {}
    for name, lin in (("off", lin_off), ("on", lin_on), ("emo", lin_emo)):
        out[f"srgb/{name}_rgb"] = comp3(fp.tonemap(P, c, lin))
        out[f"lin/{name}_rgb"] = comp3(lin)
    out.update({"lin/basecolor": comp3(base), "lin/roughness": comp1(rough.squeeze(-1)),
                "lin/metallic": comp1(metal.squeeze(-1)), "lin/emit": comp3(emit)})
    nrm = ((F.normalize(grad, dim=-1) @ pos_rt) * torch.tensor([1.0, -1.0, -1.0]) + 1.0) / 2.0
    depth = torch.zeros(N).index_add(0, ray_id, weights * step_id * stepdist)
    out.update({"etc/depth": depth, "etc/disp": 1 / (depth + alphainv_last * far), "etc/normal": comp3(nrm),
                "etc/white_bg": alphainv_last.unsqueeze(-1)})
    if render_pbr:
        vd = viewdirs[ray_id]
        normal = F.normalize(expg, dim=-1)
        parts: Dict[str, list] = {}
        for ci, idx in enumerate(torch.arange(pts.shape[0]).split(chunk_sz)):
            r = _evaluate_lts(P, c, pts[idx], vd[idx], normal[idx], base[idx], rough[idx], metal[idx], emit[idx],
                              raw_dirs_chunks[ci], s_val, num_2ndrays, lts_near)
            for k, v in r.items():
                parts.setdefault(k, []).append(v)
        for k, v in parts.items():
            out[k] = comp3(torch.cat(v, 0))
    pick = "off" if em_mode == 0 else "on"
    out["srgb/rgb"], out["lin/rgb"] = out[f"srgb/{pick}_rgb"], out[f"lin/{pick}_rgb"]
    return out
