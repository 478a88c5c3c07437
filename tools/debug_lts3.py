"""Which part of d(sum(w*brdf_eps))/d(sdf.grid) is off: the SDF-value tap or the stencil taps of the eps pass?"""
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

mode, scene_name, n_rays, s_val = "pdra", "small", 160, 90.0
R, Pn = 16, 20
sc = slab_scene(scene_name, s_val=s_val, oblique=True, n_rays=n_rays, seed=11)
torch.manual_seed(2); np.random.seed(2)
cfg = lts_cfg("cuda:0", num_2ndrays=R, num_ltspts=Pn)
m = ESRNeRF(cfg, sc.near, sc.far, sc.xyz_min, sc.xyz_max, sc.xyz_min, sc.xyz_max, sc.mask_alpha_init,
            sc.mask_density, sc.s_val, sc.num_voxels)
m.train()
init_slab_model(m, sc, seed=4)
with torch.no_grad():
    m.brdf.grid.normal_(0.0, 0.3)
m.pdra_mode = True
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
w = torch.randn(m3, 5, generator=g) / (m3 * 5) ** 0.5

orig_stencil, orig_sample = lp._stencil, fp.sample_grid
def run_oracle(detach_stencil, detach_value):
    n = {"st": 0, "sg": 0}
    def st(c_, grid, pts):
        if pts.shape[0] == m3:
            n["st"] += 1
            if n["st"] == 2 and detach_stencil:
                return orig_stencil(c_, grid.detach(), pts)
        return orig_stencil(c_, grid, pts)
    def sg(grid, gp):
        if grid.shape[1] == 1 and gp.shape[0] == m3:
            n["sg"] += 1
            if detach_value:
                return orig_sample(grid.detach(), gp)
        return orig_sample(grid, gp)
    lp._stencil, fp.sample_grid = st, sg
    try:
        P = fp.params_from_state_dict(sd)
        ro = lp.forward_training(P, c, batch, s_val, lp.Draws(**draws), tr.normal_eps, tr.emit_eps, R,
                                 ccfg.app.model.lts_near, pdra_mode=True)
        (ro["etc/brdf_eps"] * w).sum().backward()
    finally:
        lp._stencil, fp.sample_grid = orig_stencil, orig_sample
    print("  oracle calls", n)
    return P["sdf.grid"].grad.clone()

g_full = run_oracle(False, False)
g_val = run_oracle(True, False)       # value tap only
g_st = run_oracle(False, True)        # stencil only
print("oracle: |full|", float(g_full.abs().max()), "|value|", float(g_val.abs().max()), "|stencil|", float(g_st.abs().max()),
      "additivity", rel_err(g_val + g_st, g_full))
m.zero_grad(set_to_none=True)
rg = m(rays_o=b["rays_o"], rays_d=b["rays_d"], viewdirs=b["viewdirs"], em_modes=b["em_modes"],
       uncert_masks=b["uncert_masks"], s_val=s_val, normal_eps=tr.normal_eps, emit_eps=tr.emit_eps,
       draws={kk: v.cuda() for kk, v in draws.items()})
(rg["etc/brdf_eps"] * w.cuda()).sum().backward()
gg = m.sdf.grid.grad.cpu()
print("gpu vs full", rel_err(gg, g_full))
print("gpu - value vs stencil", rel_err(gg - g_val, g_st), " gpu - stencil vs value", rel_err(gg - g_st, g_val))
d = (gg - g_full)[0, 0]
nz = d.abs() > 1e-3 * g_full.abs().max()
print("cells off:", int(nz.sum()), "of nonzero", int((g_full != 0).sum()))
idx = nz.nonzero()[:12]
for i in idx:
    x, y, z = [int(v) for v in i]
    print((x, y, z), "gpu", float(gg[0, 0, x, y, z]), "full", float(g_full[0, 0, x, y, z]), "val", float(g_val[0, 0, x, y, z]), "st", float(g_st[0, 0, x, y, z]))
