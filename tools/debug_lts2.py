"""Per-output linear-functional gradient comparison of the HIP LTS path vs the CPU oracle on a synthetic scene."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from conftest import rel_err
from esr_nerf_amd.config import lts_cfg
from esr_nerf_amd.esrnerf import ESRNeRF
from esr_nerf_amd.synthetic import init_slab_model, slab_scene
from oracle import fine_path as fp
from oracle import lts_path as lp

mode, scene_name, n_rays, s_val = sys.argv[1], sys.argv[2], int(sys.argv[3]), float(sys.argv[4])
R, Pn = 16, 20
sc = slab_scene(scene_name, s_val=s_val, oblique=True, n_rays=n_rays, seed=11)
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 0
torch.manual_seed(seed); np.random.seed(seed)
cfg = lts_cfg("cuda:0", num_2ndrays=R, num_ltspts=Pn)
m = ESRNeRF(cfg, sc.near, sc.far, sc.xyz_min, sc.xyz_max, sc.xyz_min, sc.xyz_max, sc.mask_alpha_init,
            sc.mask_density, sc.s_val, sc.num_voxels)
m.train()
init_slab_model(m, sc, seed=4)
with torch.no_grad():
    m.brdf.grid.normal_(0.0, 0.3)
m.pdra_mode = mode == "pdra"
ccfg = lts_cfg("cpu", num_2ndrays=R, num_ltspts=Pn)
c = fp.make_consts(ccfg.app.model, sc.xyz_min, sc.xyz_max, sc.xyz_min, sc.xyz_max, sc.mask_alpha_init,
                   sc.mask_density, sc.near, sc.num_voxels)
sd = {k: v.detach().cpu().contiguous() for k, v in m.state_dict().items()}
keep = {}
fp.forward_training(fp.params_from_state_dict(sd), c, sc.batch, s_val, keep=keep)
m3 = keep["counts"][3]
g = torch.Generator().manual_seed(7)
draws = dict(idx=torch.randperm(m3, generator=g)[:Pn], dirs=torch.randn(Pn, R + 1, 3, generator=g),
             noise_normal=torch.randn(m3, 3, generator=g), noise_emit=torch.randn(m3, 3, generator=g))
um = torch.rand(n_rays, generator=g) < 0.4
batch = dict(sc.batch, uncert_masks=um)
tr = cfg.app.trainer
b = {k: v.cuda() for k, v in batch.items()}
print("m3", m3, "counts", keep["counts"])
keys = None
ws = {}
for it in range(40):
    P = fp.params_from_state_dict(sd)
    ro = lp.forward_training(P, c, batch, s_val, lp.Draws(**draws), tr.normal_eps, tr.emit_eps, R,
                             ccfg.app.model.lts_near, pdra_mode=(mode == "pdra"))
    if keys is None:
        keys = sorted(ro)
        for k in keys:
            ws[k] = torch.randn(ro[k].shape, generator=g) / max(1, ro[k].numel()) ** 0.5
    if it >= len(keys):
        break
    k = keys[it]
    m.zero_grad(set_to_none=True)
    rg = m(rays_o=b["rays_o"], rays_d=b["rays_d"], viewdirs=b["viewdirs"], em_modes=b["em_modes"],
           uncert_masks=b["uncert_masks"], s_val=s_val, normal_eps=tr.normal_eps, emit_eps=tr.emit_eps,
           draws={kk: v.cuda() for kk, v in draws.items()})
    if not ro[k].requires_grad:
        print(f"[{k}] no grad in oracle; fwd err {rel_err(rg[k], ro[k]):.2e}")
        continue
    (ro[k] * ws[k]).sum().backward()
    (rg[k] * ws[k].cuda()).sum().backward()
    errs = {}
    for n, p in m.named_parameters():
        go = P[n].grad if n in P else None
        if go is None:
            if p.grad is not None and float(p.grad.abs().max()) > 0:
                errs[n] = ("oracle none", float(p.grad.abs().max()))
            continue
        if p.grad is None:
            if float(go.abs().max()) > 0:
                errs[n] = ("gpu none", float(go.abs().max()))
            continue
        e = rel_err(p.grad, go)
        if e > 2e-5:
            errs[n] = (round(e, 6), float(go.abs().max()))
    print(f"[{k}] fwd err {rel_err(rg[k], ro[k]):.2e}", errs if errs else "grads all < 2e-5")
